#!/usr/bin/env python
"""GPU spot check (round 6): transform-net forward at random odd shapes / batches with every streaming kernel forced (thresholds 1) against the float64 oracle,
split-bf16 kernels on and off.  usage: gpu_shape_fuzz.py [cases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for k in ("FS_S16_MIN_TILES", "FS_CSTREAM_MIN_TILES"):
    os.environ[k] = "1"
os.environ["FS_TNET_RES_X6"] = "2"
from faststyle_amd import ckpt, engine
from oracle import tnet

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = ckpt.load_checkpoint(os.path.join(root, "models", "starry_final.ckpt"))
P64 = {k: np.asarray(v, np.float64) for k, v in tnet.strip_scope(W).items()}
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
e = engine.Engine()
flat = e.mem.from_numpy(e.flatten_params(W))
worst = 0.0
for c in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    n = int(rng.integers(1, 4)); h = int(rng.integers(41, 400)); w = int(rng.integers(41, 560))
    x = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    want = tnet.create_net(x.astype(np.float64), P64)
    errs = []
    for split in ("1", "0"):
        for k in ("FS_S16_SPLIT", "FS_CSTREAM_SPLIT"):
            os.environ[k] = split
        e.lib.fs_debug_reload_env()
        e.reset_workspaces()
        y = e.mem.to_numpy(e.tnet_forward(flat, e.mem.from_numpy(x)))
        errs.append(float(np.abs(y - want).max()) / 255.0)
    worst = max(worst, errs[0])
    print("case %d: %d x %d x %d  max err / 255: split %.2e  fp32 %.2e" % (c, n, h, w, errs[0], errs[1]), flush=True)
    assert errs[0] < 2e-5 and errs[1] < 2e-5, errs
print("gpu_shape_fuzz ok, worst %.2e" % worst)
