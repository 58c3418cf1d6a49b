#!/usr/bin/env python
"""Micro-benchmark of single conv launches (tuning aid): times fs_conv2d_fwd with torch events.
usage: micro_conv.py [name ...]   (names from CASES; default all)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import engine  # noqa: E402

# name: (N, H, W, Cin, Cout, K, stride, padding)
CASES = {
    "vgg1_2_n8": (8, 256, 256, 64, 64, 3, 1, "SAME"),
    "vgg2_2_n8": (8, 128, 128, 128, 128, 3, 1, "SAME"),
    "vgg3_2_n8": (8, 64, 64, 256, 256, 3, 1, "SAME"),
    "vgg1_2_n4": (4, 256, 256, 64, 64, 3, 1, "SAME"),
    "vgg2_2_n4": (4, 128, 128, 128, 128, 3, 1, "SAME"),
    "vgg3_2_n4": (4, 64, 64, 256, 256, 3, 1, "SAME"),
    "vgg4_2_n4": (4, 32, 32, 512, 512, 3, 1, "SAME"),
    "vgg4_1_n4": (4, 32, 32, 256, 512, 3, 1, "SAME"),
    "vgg4_d1_n4": (4, 32, 32, 512, 256, 3, 1, "SAME"),
    "res_n4": (4, 80, 80, 64, 64, 3, 1, "VALID"),
    "final9x9": (4, 256, 256, 16, 3, 9, 1, "SAME"),
    "first9x9": (4, 336, 336, 3, 16, 9, 1, "VALID"),
    "first9x9_720p": (1, 800, 1360, 3, 16, 9, 1, "VALID"),
    "final9x9_720p": (1, 720, 1280, 16, 3, 9, 1, "SAME"),
    "res_720p": (1, 196, 336, 64, 64, 3, 1, "VALID"),
    "res_b32": (32, 76, 76, 64, 64, 3, 1, "VALID"),
    # the narrow layers of the transform net at the metric's batch (32 x 256x256)
    "t_first_n32": (32, 344, 344, 3, 16, 9, 1, "VALID"),
    "t_s2a_n32": (32, 336, 336, 16, 32, 3, 2, "SAME"),
    "t_s2b_n32": (32, 168, 168, 32, 64, 3, 2, "SAME"),
    "t_final_n32": (32, 256, 256, 16, 3, 9, 1, "SAME"),
}


def main():
    names = sys.argv[1:] or list(CASES)
    e = engine.Engine()
    iters = int(os.environ.get("ITERS", "20"))
    for nm in names:
        N, H, W, Ci, Co, K, s, pad = CASES[nm]
        x = torch.randn(N, H, W, Ci, device="cuda")
        w = torch.randn(K, K, Ci, Co, device="cuda") * 0.05
        if os.environ.get("ZERO"):            # DVFS probe: all-zero operands draw less power -> higher clock
            x.zero_()
            w.zero_()
        kw = {"want_stats": True} if os.environ.get("STATS") else {}
        if os.environ.get("WINO"):            # eligible 3x3 convs through the Winograd kernel
            kw["winograd"] = {"4t": "4t", "4": 4, "6": 6}.get(os.environ["WINO"], True)
        y = e.conv2d(x, w, s, pad, **kw)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            y = e.conv2d(x, w, s, pad, **kw)
        t1.record()
        if kw:
            y = y[0]
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / iters
        fl = 2.0 * y.numel() * K * K * Ci
        print("%-12s %8.1f us  %7.2f TFLOP/s  (%.2f GFLOP)" % (nm, ms * 1e3, fl / ms / 1e9, fl / 1e9), flush=True)


if __name__ == "__main__":
    main()
