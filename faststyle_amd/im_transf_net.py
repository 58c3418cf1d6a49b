"""Host-side mirror of the reference's transform-net builder (reference im_transf_net.py).

The reference builds a TF1 graph; here ``create_net`` runs the whole net as one HIP launch
sequence (fs_tnet_forward).  Variables live in a flat float32 buffer whose layout is the key
order of the TF checkpoint (``img_t_net/...``, 48 tensors), so ``saver.restore``-compatible
files load and store without any re-layout.
"""
from collections import OrderedDict

import numpy as np

from . import _lib as L

# (layer, kernel, cin, cout) -- im_transf_net.py:37-70
_LAYERS = [("initconv_0", 9, 3, 16), ("initconv_1", 3, 16, 32), ("initconv_2", 3, 32, 64)] + \
          [("resblock_%d" % i, 3, 64, 64) for i in range(5)] + \
          [("upsample_0", 3, 64, 32), ("upsample_1", 3, 32, 16), ("upsample_2", 9, 16, 3)]


def variable_shapes(upsample_method="resize"):
    """name -> shape of the 48 variables create_net defines, sorted like the checkpoint.
    deconv2d stores its filter as [k,k,Cout,Cin] (im_transf_net.py:174)."""
    out = {}
    for name, k, ci, co in _LAYERS:
        if name.startswith("resblock"):
            for s in ("1", "2"):
                out[name + "/INscale" + s] = (co,)
                out[name + "/INshift" + s] = (co,)
                out[name + "/W" + s] = (k, k, ci, co)
        else:
            out[name + "/INscale"] = (co,)
            out[name + "/INshift"] = (co,)
            tr = upsample_method == "deconv" and name.startswith("upsample")
            out[name + "/W"] = (k, k, co, ci) if tr else (k, k, ci, co)
    return OrderedDict(sorted(out.items()))


def initial_variables(seed=0, upsample_method="resize"):
    """The reference's initialisers: conv2d N(0, 0.1) (im_transf_net.py:114), upconv2d / deconv2d
    N(0, 1) (:149, :180), INscale = 1, INshift = 0 (:233-236).  Returns name -> float32 array."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in variable_shapes(upsample_method).items():
        layer, leaf = name.split("/")
        if leaf.startswith("INscale"):
            out[name] = np.ones(shape, np.float32)
        elif leaf.startswith("INshift"):
            out[name] = np.zeros(shape, np.float32)
        else:
            std = 1.0 if layer in ("upsample_0", "upsample_1") else 0.1
            if upsample_method == "deconv" and layer == "upsample_2":
                std = 1.0
            out[name] = (rng.standard_normal(shape) * std).astype(np.float32)
    return out


def create_net(X, upsample_method="deconv", variables=None, engine=None, save_for_bwd=False):
    """create_net(X, upsample_method) -> Y (im_transf_net.py:14).

    X: device tensor [N,H,W,3] float32 (RGB 0..255); ``variables``: flat device parameter
    buffer [424102].  Like the reference the library default of ``upsample_method`` is 'deconv'
    while every script passes 'resize' (train.py:104, stylize_image.py:42).  Both methods run on the
    HIP path; 'deconv' expects the three upsample_* filters in conv2d_transpose layout [k,k,Cout,Cin].
    """
    assert upsample_method in ["deconv", "resize"]          # im_transf_net.py:28
    if engine is None or variables is None:
        raise L.FaststyleError("create_net needs an Engine and the flat variable buffer")
    return engine.tnet_forward(variables, X, save_for_bwd=save_for_bwd, upsample_method=upsample_method)
