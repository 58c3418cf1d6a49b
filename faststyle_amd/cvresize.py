"""cv2.resize of OpenCV 3.1.0 (the reference's image library, README.md:22-25) restated on the host for the two cases reference utils.py:25-40
uses on uint8 images: INTER_CUBIC for scale > 1 and INTER_AREA for scale < 1, both called with dsize=None, fx=fy=scale.

Restated from the published algorithm of OpenCV 3.1.0's imgproc/src/imgwarp.cpp (cv::resize, resizeGeneric_ / HResizeCubic / VResizeCubic with the
8-bit fixed-point path, resizeAreaFast_ / resizeArea_ with computeResizeAreaTab) -- OpenCV itself is not installable here (no network), so the
restatement is pinned by hand-derived known answers only (tests/test_cvresize.py); state: PARITY UNPINNED BEYOND KNOWN ANSWERS.  What it reproduces:

  * output size: dsize = (cvRound(W * fx), cvRound(H * fy)), cvRound = round-half-to-even; the sampling scale is 1 / fx as given (not W / dsize.w);
  * INTER_CUBIC, uchar: source coordinate fx = float32((dx + 0.5) / scale - 0.5), Keys' kernel with A = -0.75 evaluated in float32, the four weights
    rounded to 11-bit fixed point (cvRound(w * 2048), no renormalisation), out-of-range taps clamped to the edge pixel; horizontal pass in int32, vertical pass
    in int32 with the rows' own 11-bit weights, result (v + 2^21) >> 22 saturated to 0..255 -- the C++ path; the SSE2 row kernel of a vectorised build does
    the vertical sum in float32 and may round an exact tie the other way;
  * INTER_AREA with an integer factor 1 / scale (0.5, 0.25, ...): the box mean -- (a + b + c + d + 2) >> 2 for factor 2, cvRound(float32(sum) * float32(1 /
    area)) for others; edge cells of a size the factor does not divide average the pixels that exist;
  * INTER_AREA with a fractional factor: the separable coverage-weighted sum with float32 weights and float32 accumulation in OpenCV's tap order
    (computeResizeAreaTab: leading partial pixel, whole pixels at 1 / cellWidth, trailing partial pixel), saturate_cast<uchar> of the float sum.

PIL's BICUBIC (A = -0.5) / BOX, which earlier rounds used here, differ from both by up to ~10 grey levels at edges; the target Gram matrices of
train.py:136-151 under --style_target_resize are computed from this image."""
import numpy as np

_A = np.float32(-0.75)
_COEF_BITS = 11
_COEF_SCALE = 1 << _COEF_BITS


def cv_round(x):
    """cvRound: nearest integer, ties to even (lrint under the default rounding mode)."""
    return np.rint(x).astype(np.int64)


def _cubic_coeffs(x):
    """interpolateCubic (imgwarp.cpp), float32: the four Keys weights of the taps at -1, 0, +1, +2 for the fractional offset x in [0, 1)."""
    x = np.asarray(x, np.float32)
    one = np.float32(1)
    c0 = ((_A * (x + one) - np.float32(5) * _A) * (x + one) + np.float32(8) * _A) * (x + one) - np.float32(4) * _A
    c1 = ((_A + np.float32(2)) * x - (_A + np.float32(3))) * x * x + one
    xm = one - x
    c2 = ((_A + np.float32(2)) * xm - (_A + np.float32(3))) * xm * xm + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.float32)


def _cubic_axis(n_src, n_dst, scale):
    """per output index: the four clamped source indices and their 11-bit fixed-point weights (int64)"""
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    frac = (f - s.astype(np.float32)).astype(np.float32)
    w = cv_round(_cubic_coeffs(frac) * np.float32(_COEF_SCALE))          # saturate_cast<short>: |w| <= 2048 * 1.0x, no saturation occurs
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
    return idx, w


def resize_cubic_u8(img, fx, fy):
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    Wd, Hd = int(cv_round(W * fx)), int(cv_round(H * fy))
    xi, xw = _cubic_axis(W, Wd, 1.0 / fx)
    yi, yw = _cubic_axis(H, Hd, 1.0 / fy)
    src = img.astype(np.int64)
    hor = np.zeros((H, Wd, img.shape[2]), np.int64)
    for k in range(4):
        hor += src[:, xi[:, k], :] * xw[None, :, k, None]
    out = np.zeros((Hd, Wd, img.shape[2]), np.int64)
    for k in range(4):
        out += hor[yi[:, k], :, :] * yw[:, k, None, None]
    out = (out + (1 << (2 * _COEF_BITS - 1))) >> (2 * _COEF_BITS)       # FixedPtCast<int, uchar, 22>
    return np.clip(out, 0, 255).astype(np.uint8)


def _area_tab(n_src, n_dst, scale):
    """computeResizeAreaTab: [(dst index, src index, float32 weight)] in OpenCV's order"""
    tab = []
    for d in range(n_dst):
        fs1 = d * scale
        fs2 = fs1 + scale
        cell = min(scale, n_src - fs1)
        s1, s2 = int(np.ceil(fs1)), int(np.floor(fs2))
        s2 = min(s2, n_src - 1)
        s1 = min(s1, s2)
        if s1 - fs1 > 1e-3:
            tab.append((d, s1 - 1, np.float32((s1 - fs1) / cell)))
        for s in range(s1, s2):
            tab.append((d, s, np.float32(1.0 / cell)))
        if fs2 - s2 > 1e-3:
            tab.append((d, s2, np.float32(min(min(fs2 - s2, 1.0), cell) / cell)))
    return tab


def resize_area_u8(img, fx, fy):
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, C = img.shape
    Wd, Hd = int(cv_round(W * fx)), int(cv_round(H * fy))
    sx, sy = 1.0 / fx, 1.0 / fy
    isx, isy = int(cv_round(sx)), int(cv_round(sy))
    eps = np.finfo(np.float64).eps
    if abs(sx - isx) < eps and abs(sy - isy) < eps:                    # is_area_fast: integer factors
        out = np.zeros((Hd, Wd, C), np.uint8)
        src = img.astype(np.int64)
        area = isx * isy
        wfast, hfast = min(W // isx, Wd), min(H // isy, Hd)              # cells fully inside the image
        if wfast and hfast:
            blk = src[:hfast * isy, :wfast * isx].reshape(hfast, isy, wfast, isx, C).sum(axis=(1, 3))
            if isx == 2 and isy == 2:
                out[:hfast, :wfast] = ((blk + 2) >> 2).astype(np.uint8)
            else:
                out[:hfast, :wfast] = np.clip(cv_round(blk.astype(np.float32) * np.float32(1.0 / area)), 0, 255).astype(np.uint8)
        for dy in range(Hd):                                            # edge cells: the mean of the pixels that exist (0 when none)
            for dx in range(Wd):
                if dy < hfast and dx < wfast:
                    continue
                y0, x0 = dy * isy, dx * isx
                cell = src[y0:min(y0 + isy, H), x0:min(x0 + isx, W)]
                if cell.size == 0:
                    out[dy, dx] = 0
                else:
                    n = cell.shape[0] * cell.shape[1]
                    out[dy, dx] = np.clip(cv_round(cell.sum(axis=(0, 1)).astype(np.float32) / np.float32(n)), 0, 255).astype(np.uint8)
        return out
    # general case: ResizeArea_Invoker -- per source row the horizontal weighted sums (float32, in the tap order of the table), then the rows combined with the
    # vertical table in its order (the first tap of an output row assigns, the others add).  Vectorised over the outputs, sequential over a cell's taps.
    def by_rank(tab, n_dst):
        ranks, seen = [], {}
        for (d, si, a) in tab:
            r = seen.get(d, 0)
            seen[d] = r + 1
            if r == len(ranks):
                ranks.append(([], [], []))
            ranks[r][0].append(d)
            ranks[r][1].append(si)
            ranks[r][2].append(a)
        return [(np.asarray(d, np.int64), np.asarray(si, np.int64), np.asarray(a, np.float32)) for d, si, a in ranks]
    src = img.astype(np.float32)
    hor = np.zeros((H, Wd, C), np.float32)
    for (d, si, a) in by_rank(_area_tab(W, Wd, sx), Wd):
        hor[:, d, :] += src[:, si, :] * a[None, :, None]
    acc = np.zeros((Hd, Wd, C), np.float32)
    for r, (d, si, b) in enumerate(by_rank(_area_tab(H, Hd, sy), Hd)):
        term = hor[si] * b[:, None, None]
        if r == 0:
            acc[d] = term
        else:
            acc[d] = acc[d] + term
    return np.clip(cv_round(acc), 0, 255).astype(np.uint8)


def resize(img, scale):
    """utils.imresize's dispatch (reference utils.py:34-39)."""
    if scale > 1.0:
        return resize_cubic_u8(img, scale, scale)
    if scale < 1.0:
        return resize_area_u8(img, scale, scale)
    return np.asarray(img)
