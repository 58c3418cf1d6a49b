#!/usr/bin/env python
"""Winograd vs direct conv2d on small VGG-like shapes (debug aid)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import engine  # noqa: E402

eng = engine.Engine()
rng = np.random.default_rng(0)
for (N, H, W, Ci, Co) in [(2, 16, 20, 64, 64), (1, 16, 20, 64, 64), (2, 16, 20, 64, 128), (2, 24, 28, 64, 64), (2, 16, 36, 64, 64)]:
    for scale in (1.0, 100.0):
        x = np.maximum(rng.standard_normal((N, H, W, Ci)), 0).astype(np.float32) * scale
        w = (rng.standard_normal((3, 3, Ci, Co)) * 0.05).astype(np.float32)
        b = rng.standard_normal(Co).astype(np.float32)
        d = eng.mem.to_numpy(eng.conv2d(eng.mem.from_numpy(x), eng.mem.from_numpy(w), 1, "SAME", bias=eng.mem.from_numpy(b), out_relu=1))
        y = eng.mem.to_numpy(eng.conv2d(eng.mem.from_numpy(x), eng.mem.from_numpy(w), 1, "SAME", bias=eng.mem.from_numpy(b), out_relu=1,
                                        winograd=True))
        e = np.abs(y - d)
        print("N%d %dx%d %d->%d x%g: max diff %.3e (rel %.2e) at %s; mask flips %d" % (
            N, H, W, Ci, Co, scale, e.max(), e.max() / np.abs(d).max(), np.unravel_index(e.argmax(), e.shape),
            int(((y > 0) != (d > 0)).sum())), flush=True)
