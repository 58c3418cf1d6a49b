"""faststyle_amd/cvresize.py: cv2.resize of OpenCV 3.1 (INTER_CUBIC up, INTER_AREA down -- reference utils.py:25-40) restated on the host.  OpenCV cannot be
installed here, so the restatement is pinned by HAND-DERIVED known answers (the arithmetic is in the comments) and by a float64 evaluation of the published
kernels within the fixed-point rounding; parity with the binary itself stays unpinned beyond these."""
import numpy as np

from faststyle_amd import cvresize as cv, utils


def test_output_size_is_cvround_of_size_times_scale():
    # cvRound = lrint: ties to even.  5 * 0.5 = 2.5 -> 2, 7 * 0.5 = 3.5 -> 4; 3 * 1.5 = 4.5 -> 4, 5 * 1.5 = 7.5 -> 8
    img = np.zeros((5, 7, 3), np.uint8)
    assert cv.resize(img, 0.5).shape == (2, 4, 3)
    assert cv.resize(np.zeros((3, 5, 3), np.uint8), 1.5).shape == (4, 8, 3)
    assert cv.resize(img, 1.0) is img or np.array_equal(cv.resize(img, 1.0), img)


def test_cubic_weights_are_keys_kernel_with_a_minus_three_quarters_in_11_bit_fixed_point():
    # A = -0.75.  x = 0.5: w(1.5) = A(3.375) - 5A(2.25) + 8A(1.5) - 4A = -0.09375, w(0.5) = 1.25(0.125) - 2.25(0.25) + 1 = 0.59375 -> x 2048 = -192, 1216
    assert np.array_equal(cv.cv_round(cv._cubic_coeffs(np.float32(0.5)) * 2048), [-192, 1216, 1216, -192])
    # x = 0.25: w(1.25) = -0.10547, w(0.25) = 0.87891, w(0.75) = 0.26172, w(1.75) = -0.03516 -> -216, 1800, 536, -72 (sum 2048)
    assert np.array_equal(cv.cv_round(cv._cubic_coeffs(np.float32(0.25)) * 2048), [-216, 1800, 536, -72])
    # (PIL's BICUBIC uses A = -0.5: w(1.5) = -0.0625 -- the two libraries differ by ~6 grey levels beside a full-range edge)


def test_cubic_x2_of_a_step_edge_known_answer():
    # one row [0, 0, 255, 255] -> 8 columns; source coordinate (dx + 0.5) / 2 - 0.5 = -0.25, 0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3.25; taps clamped to the edge.
    #   dx 3: floor 1, frac .25, taps (0, 1, 2, 3) x (-216, 1800, 536, -72): 255 (536 - 72) = 118320 -> / 2048 = 57.8 -> 58
    #   dx 4: frac .75, taps (0, 1, 2, 3) x (-72, 536, 1800, -216): 255 (1800 - 216) = 403920 -> 197.2 -> 197
    #   dx 2: 255 x (-216) -> -26.9 -> saturates to 0 (the undershoot of A = -0.75);  dx 5: 255 x 2264 / 2048 = 281.9 -> saturates to 255
    # one source row: the four vertical taps clamp to it and their weights sum to 2048 -> the vertical pass is the identity with round-half-up
    img = np.repeat(np.array([[0, 0, 255, 255]], np.uint8)[:, :, None], 3, axis=2)
    out = cv.resize(img, 2.0)
    assert out.shape == (2, 8, 3)
    for c in range(3):
        assert np.array_equal(out[0, :, c], [0, 0, 0, 58, 197, 255, 255, 255]) and np.array_equal(out[1, :, c], out[0, :, c])


def test_cubic_agrees_with_a_float64_evaluation_of_the_kernel_within_the_fixed_point_rounding():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (17, 23, 3)).astype(np.uint8)

    def keys(x, a=-0.75):
        x = abs(x)
        return (a + 2) * x ** 3 - (a + 3) * x ** 2 + 1 if x <= 1 else (a * x ** 3 - 5 * a * x ** 2 + 8 * a * x - 4 * a if x < 2 else 0.0)

    def mat(n, nd, s):
        M = np.zeros((nd, n))
        for d in range(nd):
            c = (d + 0.5) * s - 0.5
            b = int(np.floor(c))
            for k in range(-1, 3):
                M[d, min(max(b + k, 0), n - 1)] += keys(c - (b + k))
        return M
    for f in (2.0, 1.5, 1.3):
        out = cv.resize(img, f)
        ref = np.einsum("yh,hwc,xw->yxc", mat(17, out.shape[0], 1 / f), img.astype(np.float64), mat(23, out.shape[1], 1 / f))
        assert np.abs(out - np.clip(ref, 0, 255)).max() < 1.0        # (0.5 of the final rounding + the 11-bit weights)


def test_area_half_is_the_2x2_mean_rounded_half_up():
    # (1 + 2 + 3 + 4 + 2) >> 2 = 3: the factor-2 path adds 2 and shifts (2.5 -> 3), it does NOT round to even
    img = np.repeat(np.array([[1, 2], [3, 4]], np.uint8)[:, :, None], 3, axis=2)
    assert np.array_equal(cv.resize(img, 0.5)[0, 0], [3, 3, 3])
    # odd extents: 5 x 7 -> 2 x 4 (sizes above); column 3 covers source column 6 only, rows 0..3 are used, row 4 is dropped
    img = np.arange(35, dtype=np.uint8).reshape(5, 7)[:, :, None].repeat(3, axis=2)
    out = cv.resize(img, 0.5)
    assert out[0, 0, 0] == (0 + 1 + 7 + 8 + 2) >> 2 and out[1, 2, 0] == (18 + 19 + 25 + 26 + 2) >> 2
    assert out[0, 3, 0] == 10 and out[1, 3, 0] == 24            # the mean of the two pixels that exist: (6 + 13) / 2 = 9.5 -> 10 (even), (20 + 27) / 2 = 23.5 -> 24


def test_area_quarter_rounds_the_float_mean_to_even():
    # factor 4: cvRound(float32(sum) * 0.0625): a 4 x 4 block summing to 40 -> 2.5 -> 2, to 56 -> 3.5 -> 4
    blk = np.zeros((4, 8), np.uint8)
    blk[0, :4] = [10, 10, 10, 10]
    blk[0, 4:] = [14, 14, 14, 14]
    img = blk[:, :, None].repeat(3, axis=2)
    assert np.array_equal(cv.resize(img, 0.25)[0, :, 0], [2, 4])


def test_area_fractional_factor_known_answer():
    # one row [30, 60, 90], scale 2/3 (factor 1.5): 3 -> 2 columns.  Cell 0 covers pixel 0 fully and half of pixel 1: (30 + 0.5 * 60) / 1.5 = 40;
    # cell 1 covers half of pixel 1 and pixel 2: (0.5 * 60 + 90) / 1.5 = 80.  One source row -> one output row with weight 1.
    img = np.array([[30, 60, 90]], np.uint8)[:, :, None].repeat(3, axis=2)
    out = cv.resize(img, 2.0 / 3.0)
    assert out.shape == (1, 2, 3) and np.array_equal(out[0, :, 0], [40, 80])


def test_area_agrees_with_the_coverage_weighted_mean():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)

    def mat(n, nd, s):
        M = np.zeros((nd, n))
        for d in range(nd):
            a, b = d * s, min(d * s + s, n)
            for i in range(n):
                M[d, i] = max(0.0, min(b, i + 1) - max(a, i))
            M[d] /= M[d].sum()
        return M
    for f in (0.7, 0.3, 0.45, 1.0 / 3.0):
        out = cv.resize(img, f)
        ref = np.einsum("yh,hwc,xw->yxc", mat(37, out.shape[0], 1 / f), img.astype(np.float64), mat(53, out.shape[1], 1 / f))
        assert np.abs(out - ref).max() <= 0.5 + 1e-3


def test_utils_imresize_is_the_opencv_restatement():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (12, 10, 3)).astype(np.uint8)
    assert utils.imresize(img, 1.0) is img
    assert np.array_equal(utils.imresize(img, 0.5), cv.resize_area_u8(img, 0.5, 0.5))
    assert np.array_equal(utils.imresize(img, 1.5), cv.resize_cubic_u8(img, 1.5, 1.5))
