"""Image-transform net: numpy restatement of reference im_transf_net.py (test oracle).

``create_net`` follows im_transf_net.py:14-75 line by line in the AS-WRITTEN form
(materialised x4 nearest upsample followed by a 3x3 stride-2 conv, unfused instance norm).
Parameters are a dict keyed like the checkpoint minus the ``img_t_net/`` scope prefix:
``initconv_0/W``, ``resblock_3/INscale2`` ....
"""
from collections import OrderedDict

import numpy as np

from . import nnops as F

# (name, kind, k, cin, cout) in network order -- im_transf_net.py:37-70
LAYERS = [("initconv_0", "conv", 9, 3, 16), ("initconv_1", "conv", 3, 16, 32),
          ("initconv_2", "conv", 3, 32, 64)] + \
         [("resblock_%d" % i, "res", 3, 64, 64) for i in range(5)] + \
         [("upsample_0", "up", 3, 64, 32), ("upsample_1", "up", 3, 32, 16),
          ("upsample_2", "conv", 9, 16, 3)]


def param_shapes(upsample_method="resize"):
    """name -> shape for the 48 variables, in sorted (= checkpoint) order."""
    out = {}
    for name, kind, k, ci, co in LAYERS:
        if kind == "res":
            for s in ("1", "2"):
                out[name + "/INscale" + s] = (co,)
                out[name + "/INshift" + s] = (co,)
                out[name + "/W" + s] = (k, k, ci, co)
        else:
            out[name + "/INscale"] = (co,)
            out[name + "/INshift"] = (co,)
            # deconv2d stores its filter as [k,k,Cout,Cin] (im_transf_net.py:174)
            tr = upsample_method == "deconv" and name.startswith("upsample")
            out[name + "/W"] = (k, k, co, ci) if tr else (k, k, ci, co)
    return OrderedDict(sorted(out.items()))


def init_params(seed=0, upsample_method="resize", dtype=np.float32):
    """Initialisers of the reference: conv2d N(0,0.1) (im_transf_net.py:114), upconv2d /
    deconv2d N(0,1) (:149,:180), INscale=1, INshift=0 (:233-236).  (Distribution only --
    TF's RNG stream is not reproducible outside TF.)"""
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    for name, shape in param_shapes(upsample_method).items():
        leaf = name.split("/")[1]
        if leaf.startswith("INscale"):
            p[name] = np.ones(shape, dtype)
        elif leaf.startswith("INshift"):
            p[name] = np.zeros(shape, dtype)
        else:
            layer = name.split("/")[0]
            std = 1.0 if layer in ("upsample_0", "upsample_1") else 0.1
            if upsample_method == "deconv" and layer == "upsample_2":
                std = 1.0
            p[name] = (rng.standard_normal(shape) * std).astype(dtype)
    return p


def out_shape(H, W):
    """Output H,W of create_net for an HxW input:
    4*(ceil(ceil((H+80)/2)/2) - 20)."""
    f = lambda s: 4 * (-(-(-(-(s + 80) // 2)) // 2) - 20)
    return f(H), f(W)


def upconv2d(x, w):
    """im_transf_net.py:122-155: resize x(stride**2)=x4 NEAREST, then conv3x3 stride 2 SAME."""
    return F.conv2d(F.resize_nearest(x, 4), w, stride=2, padding="SAME")


def create_net(x, params, upsample_method="resize", keep=False):
    """Forward of im_transf_net.create_net (im_transf_net.py:14-75).

    x: [N,H,W,3] float (RGB 0..255, not mean-subtracted).  Returns y, or (y, cache) with
    everything the backward needs when ``keep``.
    """
    assert upsample_method in ("deconv", "resize")       # im_transf_net.py:28
    P = params
    c = {"x_shape": x.shape, "method": upsample_method}
    acts = {}                                            # named intermediates (for tests)
    h = F.reflect_pad(x, 40)                             # :34
    strides = {"initconv_0": 1, "initconv_1": 2, "initconv_2": 2}
    for name in ("initconv_0", "initconv_1", "initconv_2"):   # :37-42
        c[name + "/in"] = h
        z = F.conv2d(h, P[name + "/W"], strides[name], "SAME")
        n, c[name + "/in_cache"] = F.inst_norm(z, P[name + "/INscale"], P[name + "/INshift"])
        c[name + "/n"] = n
        h = F.relu(n)
        acts[name] = h
    for i in range(5):                                   # :45-54, res_layer :250-276
        name = "resblock_%d" % i
        c[name + "/in"] = h
        z1 = F.conv2d(h, P[name + "/W1"], 1, "VALID")
        n1, c[name + "/in_cache1"] = F.inst_norm(z1, P[name + "/INscale1"], P[name + "/INshift1"])
        c[name + "/n1"] = n1
        a1 = F.relu(n1)
        c[name + "/a1"] = a1
        z2 = F.conv2d(a1, P[name + "/W2"], 1, "VALID")
        n2, c[name + "/in_cache2"] = F.inst_norm(z2, P[name + "/INscale2"], P[name + "/INshift2"])
        h = n2 + h[:, 2:-2, 2:-2, :]                      # :268-274 (no ReLU after the add)
        acts[name] = h
    for name in ("upsample_0", "upsample_1"):            # :57-68
        c[name + "/in"] = h
        if upsample_method == "resize":
            z = upconv2d(h, P[name + "/W"])
        else:
            z = F.conv2d_transpose(h, P[name + "/W"], 2)
        n, c[name + "/in_cache"] = F.inst_norm(z, P[name + "/INscale"], P[name + "/INshift"])
        c[name + "/n"] = n
        h = F.relu(n)
        acts[name] = h
    name = "upsample_2"                                  # :62-63 / :69-70
    c[name + "/in"] = h
    if upsample_method == "resize":
        z = F.conv2d(h, P[name + "/W"], 1, "SAME")
    else:
        z = F.conv2d_transpose(h, P[name + "/W"], 1)
    n, c[name + "/in_cache"] = F.inst_norm(z, P[name + "/INscale"], P[name + "/INshift"])
    c[name + "/n"] = n
    y = F.scaled_tanh(n)                                 # :202-215
    acts[name] = y
    c["acts"] = acts
    return (y, c) if keep else y


def create_net_bwd(dy, params, cache, masks=None):
    """Gradients of create_net wrt its 48 parameters, given dL/dy.  Returns a dict with
    the same keys as ``params`` (the input image gets no gradient in train.py).

    ``masks`` (tests only): boolean ReLU masks of the ten rectified units taken from another evaluation of the same forward
    (keys ``initconv_0..2``, ``resblock_k`` for the ReLU between a block's two convs, ``upsample_0..1``) instead of this
    evaluation's own sign pattern -- see perceptual.vgg16_bwd."""
    P, c = params, cache
    masks = masks or {}
    deconv = c["method"] == "deconv"
    g = {}
    name = "upsample_2"
    dn = F.scaled_tanh_bwd(dy, c[name + "/n"])
    dz, g[name + "/INscale"], g[name + "/INshift"] = F.inst_norm_bwd(dn, c[name + "/in_cache"])
    if deconv:
        # y = conv2d_transpose(x, W) is the input-gradient of conv2d(., W): its adjoints are
        # dx = conv2d(dy, W) and dW = conv2d_bwd_filter(input=dy, grad=x)
        g[name + "/W"] = F.conv2d_bwd_filter(dz, c[name + "/in"], 9, 1, "SAME")
        dh = F.conv2d(dz, P[name + "/W"], 1, "SAME")
    else:
        g[name + "/W"] = F.conv2d_bwd_filter(c[name + "/in"], dz, 9, 1, "SAME")
        dh = F.conv2d_bwd_input(dz, P[name + "/W"], c[name + "/in"].shape[1:3], 1, "SAME")
    for name in ("upsample_1", "upsample_0"):
        dn = dh * masks.get(name, c[name + "/n"] > 0)
        dz, g[name + "/INscale"], g[name + "/INshift"] = F.inst_norm_bwd(dn, c[name + "/in_cache"])
        if deconv:
            g[name + "/W"] = F.conv2d_bwd_filter(dz, c[name + "/in"], 3, 2, "SAME")
            dh = F.conv2d(dz, P[name + "/W"], 2, "SAME")
            continue
        up = F.resize_nearest(c[name + "/in"], 4)
        g[name + "/W"] = F.conv2d_bwd_filter(up, dz, 3, 2, "SAME")
        dup = F.conv2d_bwd_input(dz, P[name + "/W"], up.shape[1:3], 2, "SAME")
        dh = F.resize_nearest_bwd(dup, 4)
    for i in reversed(range(5)):
        name = "resblock_%d" % i
        dz2, g[name + "/INscale2"], g[name + "/INshift2"] = F.inst_norm_bwd(dh, c[name + "/in_cache2"])
        g[name + "/W2"] = F.conv2d_bwd_filter(c[name + "/a1"], dz2, 3, 1, "VALID")
        da1 = F.conv2d_bwd_input(dz2, P[name + "/W2"], c[name + "/a1"].shape[1:3], 1, "VALID")
        dn1 = da1 * masks.get(name, c[name + "/n1"] > 0)
        dz1, g[name + "/INscale1"], g[name + "/INshift1"] = F.inst_norm_bwd(dn1, c[name + "/in_cache1"])
        g[name + "/W1"] = F.conv2d_bwd_filter(c[name + "/in"], dz1, 3, 1, "VALID")
        dskip = np.pad(dh, ((0, 0), (2, 2), (2, 2), (0, 0)))
        dh = F.conv2d_bwd_input(dz1, P[name + "/W1"], c[name + "/in"].shape[1:3], 1, "VALID") + dskip
    strides = {"initconv_0": 1, "initconv_1": 2, "initconv_2": 2}
    for name in ("initconv_2", "initconv_1", "initconv_0"):
        k = 9 if name == "initconv_0" else 3
        dn = dh * masks.get(name, c[name + "/n"] > 0)
        dz, g[name + "/INscale"], g[name + "/INshift"] = F.inst_norm_bwd(dn, c[name + "/in_cache"])
        g[name + "/W"] = F.conv2d_bwd_filter(c[name + "/in"], dz, k, strides[name], "SAME")
        if name != "initconv_0":
            dh = F.conv2d_bwd_input(dz, P[name + "/W"], c[name + "/in"].shape[1:3], strides[name], "SAME")
    return OrderedDict((k, g[k]) for k in params)


def strip_scope(tensors, scope="img_t_net/"):
    """Checkpoint names -> oracle names (drops the variable_scope prefix used at
    stylize_image.py:63 / train.py:159)."""
    return OrderedDict((k[len(scope):], v) for k, v in tensors.items() if k.startswith(scope))


# ---------------------------------------------------------------------- bf16 mixed-precision restatement
def bf16_round(x):
    """float -> nearest bfloat16 (ties to even), returned as float32/float64 of the same shape."""
    a = np.ascontiguousarray(x, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def create_net_bf16(x, params):
    """create_net ('resize') with the rounding points of the HIP mixed-precision inference path
    (fs_bf16.hip, FS_FLAG_BF16): weights, the image, every conv input (after the producer's
    instance-norm + ReLU) and every stored activation are bfloat16; accumulation, instance-norm
    statistics (taken before the output is rounded), the last instance norm and the tanh are full
    precision.  The resize-conv is evaluated in its phase-collapsed form because
    the kernel rounds the COLLAPSED filter.  Test infrastructure for the config-5 path only: the
    parity bar of the project (1e-3 of the pixel range) applies to the fp32 path, not to this one."""
    P = {k: np.asarray(v, np.float64) for k, v in params.items()}
    r = lambda t: bf16_round(t).astype(np.float64)

    def norm_consts(z, name, g="INscale", b="INshift"):
        mean = z.mean(axis=(1, 2), keepdims=True)
        var = z.var(axis=(1, 2), keepdims=True)
        a = P[name + "/" + g] / np.sqrt(var + 1e-3)
        return a, P[name + "/" + b] - mean * a

    h = r(F.reflect_pad(np.asarray(x, np.float64), 40))                      # image pixels -> bf16
    strides = {"initconv_0": 1, "initconv_1": 2, "initconv_2": 2}
    for name in ("initconv_0", "initconv_1", "initconv_2"):
        z = F.conv2d(h, r(P[name + "/W"]), strides[name], "SAME")
        a, b = norm_consts(z, name)
        zs = r(z)                                                              # stored bf16
        h = r(np.maximum(a * zs + b, 0.0))                                     # consumer's staging: relu(a z + b) -> bf16
    skip_raw = (zs, a, b)                                                      # block 0 reads the raw initconv_2 output
    hk = None
    for i in range(5):
        name = "resblock_%d" % i
        z1 = F.conv2d(h, r(P[name + "/W1"]), 1, "VALID")
        a1, b1 = norm_consts(z1, name, "INscale1", "INshift1")
        a1in = r(np.maximum(a1 * r(z1) + b1, 0.0))
        z2 = F.conv2d(a1in, r(P[name + "/W2"]), 1, "VALID")
        a2, b2 = norm_consts(z2, name, "INscale2", "INshift2")
        if i == 0:
            sk = np.maximum(skip_raw[1] * skip_raw[0] + skip_raw[2], 0.0)
        else:
            sk = hk
        hk = r(a2 * r(z2) + b2 + sk[:, 2:-2, 2:-2, :])                         # h_k stored bf16
        h = hk
    for name in ("upsample_0", "upsample_1"):
        w = P[name + "/W"].astype(np.float32)
        ci, co = w.shape[2], w.shape[3]
        R = {(0, 0): (0, 1, 2), (0, 1): (), (1, 0): (0, 1), (1, 1): (2,)}      # (phase, tap) -> source rows/cols
        weff = np.zeros((2, 2, ci, 4, co), np.float32)
        for pa in range(2):
            for dy in range(2):
                for pb in range(2):
                    for dx in range(2):
                        acc = np.zeros((ci, co), np.float32)
                        for kh in range(3):
                            for kw in range(3):
                                if kh in R[(pa, dy)] and kw in R[(pb, dx)]:
                                    acc = acc + w[kh, kw]
                        weff[dy, dx, :, pa * 2 + pb, :] = acc
        weff = r(weff.reshape(2, 2, ci, 4 * co))
        hp = np.pad(h, ((0, 0), (0, 1), (0, 1), (0, 0)))
        zc = F.conv2d(hp, weff, 1, "VALID")                                    # [N,H,W,4*co]
        N, Hh, Ww, _ = zc.shape
        z = zc.reshape(N, Hh, Ww, 2, 2, co).transpose(0, 1, 3, 2, 4, 5).reshape(N, 2 * Hh, 2 * Ww, co)
        a, b = norm_consts(z, name)
        h = r(np.maximum(a * r(z) + b, 0.0))
    name = "upsample_2"
    # the kernel evaluates the 9x9 layer kw-folded (kw = 5b + v): five partial sums, each stored as bf16
    wq = r(P[name + "/W"])
    z = 0.0
    for v in range(5):
        wpart = np.zeros_like(wq)
        for kw in (v, 5 + v):
            if kw < 9:
                wpart[:, kw] = wq[:, kw]
        z = z + r(F.conv2d(h, wpart, 1, "SAME"))
    n, _ = F.inst_norm(z, P[name + "/INscale"], P[name + "/INshift"])
    return F.scaled_tanh(n)
