#!/usr/bin/env python
"""Micro-benchmark of single wgrad / Gram launches (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import engine  # noqa: E402

# name: (N, H, W, Cin, Cout, K, stride, padding, per_sample)
CASES = {
    "res_n4": (4, 82, 82, 64, 64, 3, 1, "VALID", False),
    "gram1_2": (4, 256, 256, 64, 64, 1, 1, "SAME", True),
    "gram3_3": (4, 64, 64, 256, 256, 1, 1, "SAME", True),
    "gram4_3": (4, 32, 32, 512, 512, 1, 1, "SAME", True),
    "s2_16_32": (4, 336, 336, 16, 32, 3, 2, "SAME", False),
    "first9x9": (4, 336, 336, 3, 16, 9, 1, "SAME", False),
    "fold_like": (4, 256, 256, 16, 16, 9, 1, "SAME", False),
    # the transform-net filter gradients of a 256x256 step (batch from the BATCH environment variable, default 32)
    "t_res": (0, 78, 78, 64, 64, 3, 1, "VALID", False),
    "t_init1": (0, 336, 336, 16, 32, 3, 2, "SAME", False),
    "t_init2": (0, 168, 168, 32, 64, 3, 2, "SAME", False),
    "t_init0": (0, 336, 336, 3, 16, 9, 1, "SAME", False),
    "t_up0": (0, 64, 64, 64, 128, 2, 1, (0, 0, 64, 64), False),
    "t_up1": (0, 128, 128, 32, 64, 2, 1, (0, 0, 128, 128), False),
    "t_out3x3": (0, 256, 256, 16, 16, 3, 1, "SAME", False),
}


def main():
    names = sys.argv[1:] or list(CASES)
    e = engine.Engine()
    iters = int(os.environ.get("ITERS", "20"))
    for nm in names:
        N, H, W, Ci, Co, K, s, pad, ps = CASES[nm]
        N = N or int(os.environ.get("BATCH", "32"))
        x = torch.randn(N, H, W, Ci, device="cuda")
        if not isinstance(pad, str):
            Ho, Wo = pad[2], pad[3]
        elif pad == "VALID":
            Ho, Wo = H - K + 1, W - K + 1
        else:
            Ho, Wo = -(-H // s), -(-W // s)
        dy = x if ps else torch.randn(N, Ho, Wo, Co, device="cuda")
        dw = e.conv2d_wgrad(x, dy, K, s, pad, per_sample=ps)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            dw = e.conv2d_wgrad(x, dy, K, s, pad, per_sample=ps)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / iters
        fl = 2.0 * N * Ho * Wo * K * K * Ci * Co
        print("%-10s %8.1f us (incl. slab reduce)  %7.2f TFLOP/s  (%.2f GFLOP)" % (nm, ms * 1e3, fl / ms / 1e9, fl / 1e9), flush=True)


if __name__ == "__main__":
    main()
