"""Numpy restatements of the stock TF1 ops the hot path instantiates (test oracle).

Every function is dtype-generic (float32 in -> float32 arithmetic, float64 in ->
float64) and NHWC / HWIO like the reference.  Forward functions cite the reference call
site; ``*_bwd`` functions are the hand-derived adjoints used to check the HIP backward
kernels (cross-checked against torch autograd in tests/test_oracle_backward.py).
"""
import numpy as np


# ----------------------------------------------------------------------------- padding
def same_pads(size, k, stride):
    """TF 'SAME' padding for one spatial dim -> (out, pad_before, pad_after).

    out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); before = total//2.
    (tf.nn.conv2d / max_pool semantics used at im_transf_net.py:115,150 and vgg16.py:47.)
    """
    out = -(-size // stride)
    tot = max((out - 1) * stride + k - size, 0)
    return out, tot // 2, tot - tot // 2


def reflect_pad(x, p):
    """tf.pad(..., mode='REFLECT') on H and W (im_transf_net.py:78-88)."""
    return np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)), mode="reflect")


def reflect_pad_bwd(dy, p):
    """Adjoint of reflect_pad: fold the mirrored borders back (scatter-add)."""
    H, W = dy.shape[1] - 2 * p, dy.shape[2] - 2 * p
    # padded row p-i mirrors source row i (i=1..p); padded row p+H-1+i mirrors H-1-i.
    rows = dy[:, p:-p].copy()                  # [N,H,W+2p,C]
    rows[:, 1:p + 1] += dy[:, :p][:, ::-1]
    rows[:, H - 1 - p:H - 1] += dy[:, -p:][:, ::-1]
    dx = rows[:, :, p:-p].copy()
    dx[:, :, 1:p + 1] += rows[:, :, :p][:, :, ::-1]
    dx[:, :, W - 1 - p:W - 1] += rows[:, :, -p:][:, :, ::-1]
    return dx


# ----------------------------------------------------------------------------- conv2d
def _conv_geometry(H, W, kh, kw, stride, padding):
    if padding == "SAME":
        Ho, pt, pb = same_pads(H, kh, stride)
        Wo, pl, pr = same_pads(W, kw, stride)
    elif padding == "VALID":
        Ho, Wo = (H - kh) // stride + 1, (W - kw) // stride + 1
        pt = pb = pl = pr = 0
    else:
        raise ValueError(padding)
    return Ho, Wo, pt, pb, pl, pr


def conv2d(x, w, stride=1, padding="SAME"):
    """tf.nn.conv2d: cross-correlation, NHWC x HWIO, no bias (im_transf_net.py:115)."""
    N, H, W, Ci = x.shape
    kh, kw, ci2, Co = w.shape
    assert Ci == ci2
    Ho, Wo, pt, pb, pl, pr = _conv_geometry(H, W, kh, kw, stride, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((N, Ho, Wo, Co), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            xs = xp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
            y += np.tensordot(xs, w[i, j], axes=1)
    return y


def conv2d_bwd_input(dy, w, in_hw, stride=1, padding="SAME"):
    """dL/dx of conv2d (== tf.nn.conv2d_backprop_input)."""
    H, W = in_hw
    kh, kw, Ci, Co = w.shape
    N, Ho, Wo, _ = dy.shape
    Ho2, Wo2, pt, pb, pl, pr = _conv_geometry(H, W, kh, kw, stride, padding)
    assert (Ho, Wo) == (Ho2, Wo2)
    dxp = np.zeros((N, H + pt + pb, W + pl + pr, Ci), dtype=dy.dtype)
    for i in range(kh):
        for j in range(kw):
            dxp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :] += \
                np.tensordot(dy, w[i, j].T, axes=1)
    return dxp[:, pt:pt + H, pl:pl + W, :]


def conv2d_bwd_filter(x, dy, k, stride=1, padding="SAME"):
    """dL/dw of conv2d (== tf.nn.conv2d_backprop_filter)."""
    N, H, W, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    _, _, pt, pb, pl, pr = _conv_geometry(H, W, k, k, stride, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dw = np.zeros((k, k, Ci, Co), dtype=x.dtype)
    for i in range(k):
        for j in range(k):
            xs = xp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
            dw[i, j] = np.tensordot(xs, dy, axes=([0, 1, 2], [0, 1, 2]))
    return dw


def conv2d_transpose(x, w, stride):
    """tf.nn.conv2d_transpose, SAME, output = in*stride, filter [k,k,Cout,Cin]
    (im_transf_net.py:184-188).  By definition the input-gradient of conv2d."""
    N, H, W, _ = x.shape
    return conv2d_bwd_input(x, w, (H * stride, W * stride), stride, "SAME")


# ----------------------------------------------------------------------------- resize
def resize_nearest(x, factor):
    """tf.image.resize_images(method=1) to an integer multiple, align_corners=False:
    up[i] = x[i // factor] (im_transf_net.py:140-142)."""
    return np.repeat(np.repeat(x, factor, axis=1), factor, axis=2)


def resize_nearest_bwd(dy, factor):
    N, H, W, C = dy.shape
    return dy.reshape(N, H // factor, factor, W // factor, factor, C).sum(axis=(2, 4))


# ----------------------------------------------------------------------------- instance norm
def inst_norm(x, scale, shift, eps=1e-3):
    """im_transf_net.py:218-247: tf.nn.moments over axes [1,2] (population variance,
    two-pass), (x-mu)/sqrt(var+eps)*scale+shift.  Returns (y, cache)."""
    # float64 accumulators: numpy's strided float32 reduction is a naive running sum
    # (error ~ n*eps, visible at 4e5 pixels); TF/Eigen reduce tree-wise.  Values are
    # rounded back to x.dtype so the float32 oracle still models float32 statistics.
    mu = x.mean(axis=(1, 2), keepdims=True, dtype=np.float64).astype(x.dtype)
    var = np.square(x - mu).mean(axis=(1, 2), keepdims=True, dtype=np.float64).astype(x.dtype)
    rstd = (1.0 / np.sqrt(var + x.dtype.type(eps))).astype(x.dtype)
    xhat = (x - mu) * rstd
    return scale * xhat + shift, (xhat, rstd, scale)


def inst_norm_bwd(dy, cache):
    xhat, rstd, scale = cache
    t = dy.dtype
    dshift = dy.sum(axis=(0, 1, 2), dtype=np.float64).astype(t)
    dscale = (dy * xhat).sum(axis=(0, 1, 2), dtype=np.float64).astype(t)
    g = dy * scale
    m1 = g.mean(axis=(1, 2), keepdims=True, dtype=np.float64).astype(t)
    m2 = (g * xhat).mean(axis=(1, 2), keepdims=True, dtype=np.float64).astype(t)
    dx = rstd * (g - m1 - xhat * m2)
    return dx, dscale, dshift


# ----------------------------------------------------------------------------- activations
def relu(x):
    return np.maximum(x, 0)


def scaled_tanh(x):
    """(255*tanh(x)+255)/2 (im_transf_net.py:202-215)."""
    t = x.dtype.type
    return (t(255.0) * np.tanh(x) + t(255.0)) / t(2.0)


def scaled_tanh_bwd(dy, x):
    th = np.tanh(x)
    return dy * x.dtype.type(127.5) * (1 - th * th)


# ----------------------------------------------------------------------------- VGG pieces
def bias_relu(x, b):
    return np.maximum(x + b, 0)


def max_pool_2x2(x):
    """tf.nn.max_pool ksize 2, stride 2, SAME (vgg16.py:63-67): out=ceil(in/2), the
    padded cells (only 'after', odd sizes) never win.  Returns (y, argmax index 0..3)."""
    N, H, W, C = x.shape
    Ho, Wo = -(-H // 2), -(-W // 2)
    xp = np.full((N, Ho * 2, Wo * 2, C), -np.inf, dtype=x.dtype)
    xp[:, :H, :W, :] = x
    win = xp.reshape(N, Ho, 2, Wo, 2, C).transpose(0, 1, 3, 2, 4, 5).reshape(N, Ho, Wo, 4, C)
    idx = win.argmax(axis=3)          # first maximum in row-major window order
    y = np.take_along_axis(win, idx[:, :, :, None, :], axis=3)[:, :, :, 0, :]
    return y, idx


def max_pool_2x2_bwd(dy, idx, in_hw):
    H, W = in_hw
    N, Ho, Wo, C = dy.shape
    win = np.zeros((N, Ho, Wo, 4, C), dtype=dy.dtype)
    np.put_along_axis(win, idx[:, :, :, None, :], dy[:, :, :, None, :], axis=3)
    dxp = win.reshape(N, Ho, Wo, 2, 2, C).transpose(0, 1, 3, 2, 4, 5).reshape(N, Ho * 2, Wo * 2, C)
    return dxp[:, :H, :W, :]
