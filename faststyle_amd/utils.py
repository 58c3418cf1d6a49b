"""Image I/O wrappers with the reference's conventions (reference utils.py:14-52), on PIL.

The reference uses OpenCV 3.1: imread -> BGR->RGB uint8; imwrite of float32 rounds + saturates
to uint8 and writes JPEG quality 95, 4:2:0 chroma.  imresize: INTER_CUBIC up / INTER_AREA down --
since round 6 OpenCV's own resampling restated on the host (faststyle_amd/cvresize.py: A = -0.75
cubic in 11-bit fixed point, coverage-weighted area means, cvRound output sizes; pinned by
hand-derived known answers, tests/test_cvresize.py -- OpenCV itself cannot be installed here).
"""
import os

import numpy as np
from PIL import Image


def imread(path):
    """utils.imread (utils.py:14-22): RGB uint8 [H,W,3]."""
    return np.asarray(Image.open(path).convert("RGB"))


def imresize(img, scale):
    """utils.imresize (utils.py:25-40): cubic for scale>1, area for scale<1, identity at 1."""
    if scale == 1.0:
        return img
    from . import cvresize
    a = np.asarray(img)
    if a.dtype != np.uint8:      # (the reference only ever resizes what imread returned: uint8)
        a = np.clip(np.rint(a), 0, 255).astype(np.uint8)
    if a.ndim == 2:
        return cvresize.resize(a[:, :, None], scale)[:, :, 0]
    return cvresize.resize(a, scale)


def imwrite(path, img):
    """utils.imwrite (utils.py:43-52): float RGB image -> round/saturate u8 -> file
    (JPEG q95 4:2:0 like cv2.imwrite's defaults)."""
    u8 = np.clip(np.rint(np.asarray(img, np.float64)), 0, 255).astype(np.uint8)
    d = os.path.dirname(path)
    if d and not os.path.isdir(d):
        os.makedirs(d)
    ext = os.path.splitext(path)[1].lower()
    if ext in (".jpg", ".jpeg"):
        Image.fromarray(u8).save(path, format="JPEG", quality=95, subsampling="4:2:0")
    else:
        Image.fromarray(u8).save(path)


def get_layers(layer_names, vgg):
    """utils.get_layers (utils.py:55-63): the named VGG tensors (here: of an explicit vgg16 builder)."""
    return vgg.layers([n.split("/")[-1].split(":")[0] for n in layer_names])


def get_grams(layer_names, vgg):
    """utils.get_grams (utils.py:66-83): per-sample Gram matrices [b,c,c] = F^T F / (h*w*c) of the named layers."""
    feats = get_layers(layer_names, vgg)
    if any(getattr(f, "requires_grad", False) for f in feats):     # differentiable (round 6): utils.get_grams inside a graph that is .backward()-ed
        from . import autograd
        return [autograd.gram(f, vgg.engine) for f in feats]
    return [vgg.engine.gram(f) for f in feats]
