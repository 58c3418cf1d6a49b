"""Perceptual-loss path: numpy restatement of reference libs/vgg16.py, utils.get_grams,
losses.py and the train.py step (test oracle; TRAINING-PATH PARITY IS UNPINNED BY THE
REFERENCE -- see oracle/__init__.py).
"""
from collections import OrderedDict

import numpy as np

from . import nnops as F
from . import tnet

VGG_MEAN = (123.68, 116.779, 103.939)          # vgg16.py:41 (RGB order)
# conv layers that can ever be fetched by train.py (conv5_x is dead code on the path)
VGG_LAYERS = [("conv1_1", 3, 64), ("conv1_2", 64, 64), ("pool1",),
              ("conv2_1", 64, 128), ("conv2_2", 128, 128), ("pool2",),
              ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("pool3",),
              ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512)]


def synthetic_vgg_weights(seed=3, dtype=np.float32):
    """He-normal stand-in for libs/vgg16_weights.npz (absent offline), in the npz key
    convention load_weights relies on (vgg16.py:257-266): ``convX_Y_W`` [3,3,Cin,Cout]
    sorts before ``convX_Y_b`` [Cout]."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for layer in VGG_LAYERS:
        if len(layer) == 3:
            name, ci, co = layer
            w[name + "_W"] = (rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(dtype)
            w[name + "_b"] = (rng.standard_normal((co,)) * 0.05).astype(dtype)
    return w


def vgg16(imgs, weights, upto="conv4_3", keep=False):
    """vgg16.convlayers (vgg16.py:36-220) up to ``upto``: mean-subtract, then
    conv3x3 SAME + bias + ReLU, 2x2/2 SAME max-pools.  Returns {name: post-ReLU tensor}
    (the tensor TF names ``vgg/<name>:0``), plus a cache when ``keep``."""
    t = imgs.dtype
    h = imgs - np.asarray(VGG_MEAN, dtype=t).reshape(1, 1, 1, 3)
    feats, cache = OrderedDict(), {}
    for layer in VGG_LAYERS:
        name = layer[0]
        if len(layer) == 1:
            cache[name + "/in_hw"] = h.shape[1:3]
            h, cache[name + "/idx"] = F.max_pool_2x2(h)
        else:
            cache[name + "/in"] = h
            h = F.bias_relu(F.conv2d(h, weights[name + "_W"], 1, "SAME"), weights[name + "_b"])
            feats[name] = h
        if name == upto:
            break
    return (feats, cache) if keep else feats


def vgg16_bwd(dfeats, feats, weights, cache, upto="conv4_3", masks=None):
    """dL/d(imgs) given dL/d(feature) for any subset of layers (dgrad only: VGG is frozen,
    train.py:198-199).

    ``masks`` (tests only): the NON-DIFFERENTIABLE decisions of the backward pass taken from another evaluation of the same
    forward -- ``masks["conv3_2"]`` a boolean ReLU mask, ``masks["pool1/idx"]`` the arg-max (0..3) of every pooling window --
    instead of this evaluation's own.  A pre-activation within float32 noise of zero (or two window entries within noise
    of each other) is decided differently by a float32 and a float64 forward; with the float32 path's decisions injected,
    everything that remains is smooth and the two gradients agree to rounding."""
    names = [l[0] for l in VGG_LAYERS]
    names = names[:names.index(upto) + 1]
    masks = masks or {}
    dh = None
    for name in reversed(names):
        if name.startswith("pool"):
            dh = F.max_pool_2x2_bwd(dh, masks.get(name + "/idx", cache[name + "/idx"]), cache[name + "/in_hw"])
            continue
        if name in dfeats:
            dh = dfeats[name] if dh is None else dh + dfeats[name]
        dz = dh * masks.get(name, feats[name] > 0)
        dh = F.conv2d_bwd_input(dz, weights[name + "_W"], cache[name + "/in"].shape[1:3], 1, "SAME")
    return dh


def gram(feat):
    """utils.get_grams (utils.py:66-83): per-sample F^T F / (h*w*c)."""
    b, h, w, c = feat.shape
    Fm = feat.reshape(b, h * w, c)
    return np.matmul(Fm.transpose(0, 2, 1), Fm) / feat.dtype.type(h * w * c)


def gram_bwd(dG, feat):
    b, h, w, c = feat.shape
    Fm = feat.reshape(b, h * w, c)
    return (np.matmul(Fm, dG + dG.transpose(0, 2, 1)) / feat.dtype.type(h * w * c)).reshape(feat.shape)


def content_loss(layers, targets, weights):
    """losses.content_loss (losses.py:12-40): w * sum_{b,h,w,c}(diff^2) / (h*w*c)."""
    total, grads = 0.0, []
    for x, tgt, wgt in zip(layers, targets, weights):
        _, h, w, c = x.shape
        d = x - tgt
        total = total + x.dtype.type(wgt) * np.sum(np.square(d), dtype=np.float64).astype(x.dtype) / x.dtype.type(h * w * c)
        grads.append(d * x.dtype.type(2.0 * wgt / (h * w * c)))
    return total, grads


def style_loss(grams, target_grams, weights):
    """losses.style_loss (losses.py:43-67): w * sum_{b,i,j}(G-Gt)^2 / (c*c); the target
    [1,c,c] broadcasts over the batch."""
    total, grads = 0.0, []
    for G, Gt, wgt in zip(grams, target_grams, weights):
        _, c1, c2 = G.shape
        d = G - Gt
        total = total + G.dtype.type(wgt) * np.sum(np.square(d), dtype=np.float64).astype(G.dtype) / G.dtype.type(c1 * c2)
        grads.append(d * G.dtype.type(2.0 * wgt / (c1 * c2)))
    return total, grads


def tv_loss(x):
    """losses.tv_loss (losses.py:70-97): sum of squared forward differences, both axes."""
    v = x[:, :-1] - x[:, 1:]
    h = x[:, :, :-1] - x[:, :, 1:]
    loss = np.sum(np.square(h), dtype=np.float64).astype(x.dtype) + np.sum(np.square(v), dtype=np.float64).astype(x.dtype)
    g = np.zeros_like(x)
    g[:, :-1] += 2 * v
    g[:, 1:] -= 2 * v
    g[:, :, :-1] += 2 * h
    g[:, :, 1:] -= 2 * h
    return loss, g


def target_grams(style_img, vgg_w, style_layers):
    """train.py:144-151: one VGG pass on the style image -> list of [1,c,c] Grams."""
    feats = vgg16(style_img, vgg_w, upto=max(style_layers))
    return [gram(feats[n]) for n in style_layers]


def perceptual_loss(y, content_targets, tgt_grams, vgg_w, content_layers=("conv3_3",),
                    style_layers=("conv1_2", "conv2_2", "conv3_3", "conv4_3"),
                    content_weights=(1.0,), style_weights=(5.0, 5.0, 5.0, 5.0), beta=0.0, masks=None):
    """train.py:164-184: loss = content + style + beta*tv on y (the net output, fed to
    VGG directly).  Returns (dict of loss scalars, dL/dy)."""
    upto = max(list(content_layers) + list(style_layers))
    feats, cache = vgg16(y, vgg_w, upto=upto, keep=True)
    closs, cgrads = content_loss([feats[n] for n in content_layers], content_targets, content_weights)
    grams = [gram(feats[n]) for n in style_layers]
    sloss, sgrads = style_loss(grams, tgt_grams, style_weights)
    dfeats = {}
    for n, g in zip(content_layers, cgrads):
        dfeats[n] = dfeats.get(n, 0) + g
    for n, g in zip(style_layers, sgrads):
        dfeats[n] = dfeats.get(n, 0) + gram_bwd(g, feats[n])
    dy = vgg16_bwd(dfeats, feats, vgg_w, cache, upto=upto, masks=masks)
    tv, dtv = tv_loss(y)
    t = y.dtype.type
    loss = closs + sloss + t(beta) * tv
    if beta != 0.0:
        dy = dy + t(beta) * dtv
    return {"loss": loss, "content_loss": closs, "style_loss": sloss, "tv_loss": t(beta) * tv}, dy


def adam_tf(params, grads, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (train.py:203) TF1 form: lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    theta -= lr_t * m / (sqrt(v) + eps) -- epsilon OUTSIDE the bias correction.
    ``t`` is the 1-based step count.  Updates in place."""
    for k in params:
        dt = params[k].dtype.type
        lr_t = dt(lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
        m[k] = dt(b1) * m[k] + dt(1 - b1) * grads[k]
        v[k] = dt(b2) * v[k] + dt(1 - b2) * np.square(grads[k])
        params[k] -= lr_t * m[k] / (np.sqrt(v[k]) + dt(eps))


def train_step(params, batch, tgt_grams, vgg_w, beta=0.0, **kw):
    """One train.py loop body (train.py:245-275) without the optimiser update:
    content targets from the RAW batch (train.py:250-251 overrides Y with the batch),
    then forward/backward through create_net + VGG.  Returns (losses, grads dict)."""
    content_layers = kw.get("content_layers", ("conv3_3",))
    masks = kw.pop("masks", None) or {}           # {"vgg": ..., "tnet": ...}: see vgg16_bwd / tnet.create_net_bwd
    feats = vgg16(batch, vgg_w, upto=max(content_layers))
    content_targets = [feats[n] for n in content_layers]
    y, cache = tnet.create_net(batch, params, "resize", keep=True)
    losses, dy = perceptual_loss(y, content_targets, tgt_grams, vgg_w, beta=beta, masks=masks.get("vgg"), **kw)
    grads = tnet.create_net_bwd(dy, params, cache, masks=masks.get("tnet"))
    return losses, grads, y
