#!/usr/bin/env python
"""Winograd vs direct VGG path on the GPU (debug aid): loss and input gradient of fs_perceptual_loss with the Winograd
kernel on / off / on for single layers (FS_VGG_WINO_MASK), against the fp64 oracle, on the shapes of
tests/test_aux_scripts.py::test_slow_style_steps_match_oracle and for several input seeds."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import engine  # noqa: E402
from oracle import perceptual  # noqa: E402


def f64(W):
    return dict((k, v.astype(np.float64)) for k, v in W.items())


def main():
    eng = engine.Engine()
    rng = np.random.default_rng(5)
    Wv = perceptual.synthetic_vgg_weights(seed=3)
    cfg = engine.default_loss_cfg()
    cfg["beta"] = 1e-4
    style = rng.uniform(0, 255, (1, 24, 28, 3)).astype(np.float32)
    cont = rng.uniform(0, 255, (1, 16, 20, 3)).astype(np.float32)
    eng.vgg_load(Wv)
    tg = eng.style_targets(eng.mem.from_numpy(style), cfg)
    tgo = perceptual.target_grams(style.astype(np.float64), f64(Wv), cfg["style_layers"])
    feats = perceptual.vgg16(cont.astype(np.float64), f64(Wv), upto="conv3_3")
    cases = sys.argv[1:] or ["1", "0", "f1"]
    for xseed in (7, 8, 9, 10, 11):
        X = (np.random.RandomState(xseed).rand(*cont.shape) * 255.0).astype(np.float32)
        lo, dXo = perceptual.perceptual_loss(X.astype(np.float64), [feats["conv3_3"]], tgo, f64(Wv), beta=1e-4)
        sc = np.abs(dXo).max()
        for wino in cases:
            os.environ["FS_CONV_WINO"] = "0" if wino == "0" else "1"
            os.environ.pop("FS_VGG_WINO_MASK", None)
            if wino[0] == "f":
                os.environ["FS_VGG_WINO_MASK"] = hex(1 << int(wino[1:]))
            if wino[0] == "d":
                os.environ["FS_VGG_WINO_MASK"] = hex(1 << (16 + int(wino[1:])))
            losses, dX = eng.perceptual_loss(eng.mem.from_numpy(X), eng.mem.from_numpy(cont), tg, cfg)
            l, d = eng.mem.to_numpy(losses).copy(), eng.mem.to_numpy(dX).copy()
            err = np.abs(d - dXo)
            print("seed %d wino=%-3s loss %.6f (oracle %.6f)  dX max err %.3e rms %.3e (rel. to max|dX|), #elements with err > 1e-4: %d"
                  % (xseed, wino, l[0], lo["loss"], err.max() / sc, np.sqrt((err ** 2).mean()) / sc, int((err / sc > 1e-4).sum())),
                  flush=True)


if __name__ == "__main__":
    main()
