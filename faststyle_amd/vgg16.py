"""Host-side mirror of reference libs/vgg16.py: weight-file convention + synthetic stand-in."""
from collections import OrderedDict

import numpy as np

from . import _lib as L


def load_weights(weight_file):
    """vgg16.load_weights (vgg16.py:257-266): np.load(npz); sorted(keys); conv entries only.
    Returns {'conv1_1_W': [3,3,3,64], 'conv1_1_b': [64], ...} for conv1_1..conv4_3."""
    z = np.load(weight_file)
    out = OrderedDict()
    for k in sorted(z.keys()):
        if "fc" in k:
            break
        out[k] = z[k]
    need = [n + s for n in L.VGG_LAYER_NAMES for s in ("_W", "_b")]
    missing = [k for k in need if k not in out]
    if missing:
        raise L.FaststyleError("%s lacks %s" % (weight_file, missing))
    return OrderedDict((k, out[k]) for k in need)


def synthetic_weights(seed=3):
    """He-normal stand-in in the same key convention, for benchmarks when
    libs/vgg16_weights.npz (git-ignored by the reference, fetched by get_vgg16_weights.sh) is
    absent.  Synthetic weights change the loss VALUES, not the work per step."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for n, ci, co in zip(L.VGG_LAYER_NAMES, L.VGG_CIN, L.VGG_COUT):
        w[n + "_W"] = (rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
        w[n + "_b"] = (rng.standard_normal((co,)) * 0.05).astype(np.float32)
    return w


class vgg16(object):
    """Builder with the reference's shape (libs/vgg16.py:26-33): ``vgg16(imgs, weights, engine=...)`` then the
    layers as attributes ``conv1_1 ... conv4_3`` (post-ReLU device tensors, computed on first access through
    fs_vgg_features; conv5_x / fc layers are never used by the reference's scripts and are not built)."""

    def __init__(self, imgs, weights=None, sess=None, engine=None):
        if engine is None:
            raise L.FaststyleError("vgg16 needs the Engine that owns the device")
        self.imgs = imgs
        self.engine = engine
        self._feats = {}
        if weights is not None:
            self.load_weights(weights)

    def load_weights(self, weight_file, sess=None):
        self.engine.vgg_load(load_weights(weight_file) if isinstance(weight_file, str) else weight_file)
        self._feats = {}

    def layers(self, names):
        need = [n for n in names if n not in self._feats]
        if need:
            if getattr(self.imgs, "requires_grad", False):     # differentiable (round 6): the activations carry their graph back to imgs (fs_vgg_dgrad)
                from . import autograd
                outs = autograd.vgg_features(self.imgs, self.engine, need)
            else:
                outs = self.engine.vgg_features(self.imgs, need)
            for n, t in zip(need, outs):
                self._feats[n] = t
        return [self._feats[n] for n in names]

    def __getattr__(self, name):
        if name in L.VGG_LAYER_NAMES:
            return self.layers([name])[0]
        raise AttributeError(name)
