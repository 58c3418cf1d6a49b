#!/usr/bin/env python
"""Phase cycle counts of conv_wgrad workgroups (tuning aid; needs the -DFS_CONV_TRACE build, see conv_trace.py)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import engine  # noqa: E402
from tools.micro_wgrad import CASES  # noqa: E402


def main():
    names = sys.argv[1:] or ["res_n4", "s2_16_32", "gram1_2", "gram3_3"]
    import ctypes
    e = engine.Engine()
    e.lib.fs_debug_wgrad_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for nm in names:
        N, H, W, Ci, Co, K, s, pad, ps = CASES[nm]
        x = torch.randn(N, H, W, Ci, device="cuda")
        Ho, Wo = (H - K + 1, W - K + 1) if pad == "VALID" else (-(-H // s), -(-W // s))
        dy = x if ps else torch.randn(N, Ho, Wo, Co, device="cuda")
        for _ in range(3):
            e.conv2d_wgrad(x, dy, K, s, pad, per_sample=ps)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            e.conv2d_wgrad(x, dy, K, s, pad, per_sample=ps)
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 100.0
        assert e.lib.fs_debug_wgrad_trace_reset() == 0
        e.conv2d_wgrad(x, dy, K, s, pad, per_sample=ps)
        torch.cuda.synchronize()
        buf = np.zeros((4096, 8), dtype=np.int64)
        assert e.lib.fs_debug_wgrad_trace(buf.ctypes.data, 4096) == 0
        live = buf[buf[:, 6] > 0]
        life = live[:, 6] - live[:, 0]
        span = live[:, 6].max() - live[:, 0].min()
        fl = 2.0 * N * Ho * Wo * K * K * Ci * Co
        print("%s: %.1f us incl. slab reduce (%.1f TFLOP/s); %d workgroups, kernel span %d ticks, lifetime mean %d (min %d max %d)"
              % (nm, us, fl / us / 1e6, len(live), span, life.mean(), life.min(), life.max()), flush=True)
        for i, k in ((3, "stage"), (2, "sweep"), (4, "barrier"), (5, "slab write")):
            v = live[:, i]
            print("   %-10s mean %8.0f  (%5.1f%% of lifetime)  min %7d max %7d" % (k, v.mean(), 100.0 * v.mean() / life.mean(), v.min(), v.max()))
        print("   wave-slot histogram: %s" % np.bincount((live[:, 7] & 0xF).astype(int), minlength=4)[:6].tolist())


if __name__ == "__main__":
    main()
