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


def create_net_bwd(dy, params, cache):
    """Gradients of create_net wrt its 48 parameters, given dL/dy.  Returns a dict with
    the same keys as ``params`` (the input image gets no gradient in train.py)."""
    P, c = params, cache
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
        dn = dh * (c[name + "/n"] > 0)
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
        dn1 = da1 * (c[name + "/n1"] > 0)
        dz1, g[name + "/INscale1"], g[name + "/INshift1"] = F.inst_norm_bwd(dn1, c[name + "/in_cache1"])
        g[name + "/W1"] = F.conv2d_bwd_filter(c[name + "/in"], dz1, 3, 1, "VALID")
        dskip = np.pad(dh, ((0, 0), (2, 2), (2, 2), (0, 0)))
        dh = F.conv2d_bwd_input(dz1, P[name + "/W1"], c[name + "/in"].shape[1:3], 1, "VALID") + dskip
    strides = {"initconv_0": 1, "initconv_1": 2, "initconv_2": 2}
    for name in ("initconv_2", "initconv_1", "initconv_0"):
        k = 9 if name == "initconv_0" else 3
        dn = dh * (c[name + "/n"] > 0)
        dz, g[name + "/INscale"], g[name + "/INshift"] = F.inst_norm_bwd(dn, c[name + "/in_cache"])
        g[name + "/W"] = F.conv2d_bwd_filter(c[name + "/in"], dz, k, strides[name], "SAME")
        if name != "initconv_0":
            dh = F.conv2d_bwd_input(dz, P[name + "/W"], c[name + "/in"].shape[1:3], strides[name], "SAME")
    return OrderedDict((k, g[k]) for k in params)


def strip_scope(tensors, scope="img_t_net/"):
    """Checkpoint names -> oracle names (drops the variable_scope prefix used at
    stylize_image.py:63 / train.py:159)."""
    return OrderedDict((k[len(scope):], v) for k, v in tensors.items() if k.startswith(scope))
