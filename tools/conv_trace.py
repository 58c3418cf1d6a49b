#!/usr/bin/env python
"""Where a conv_igemm workgroup spends its cycles (tuning aid; needs the DEBUG build of the library):

    hipcc ... -DFS_CONV_TRACE -o exp/libtrace.so ;  FASTSTYLE_HIP_LIB=exp/libtrace.so python tools/conv_trace.py [case ...]

Wave 0 of every workgroup timestamps its phases with the shader clock (s_memtime): prologue (first issue + commit),
per-chunk sweep / commit / barrier wait, epilogue.  Prints the per-workgroup means, the lifetime of a workgroup and
how many workgroup 'rounds' the launch took."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import engine  # noqa: E402
from tools.micro_conv import CASES  # noqa: E402


KW = {"winograd": True} if os.environ.get("WINO") else {}    # WINO=1: the Winograd kernel (build with -DFS_WINO2_TRACE)


def main():
    names = sys.argv[1:] or ["vgg1_2_n4", "vgg2_2_n4", "vgg3_2_n4", "vgg4_2_n4"]
    CASES.setdefault("vgg1_2_n4", (4, 256, 256, 64, 64, 3, 1, "SAME"))
    CASES.setdefault("vgg2_2_n4", (4, 128, 128, 128, 128, 3, 1, "SAME"))
    e = engine.Engine()
    rd = e.lib.fs_debug_conv_trace          # only the -DFS_CONV_TRACE build exports it
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for nm in names:
        N, H, W, Ci, Co, K, s, pad = CASES[nm]
        x = torch.randn(N, H, W, Ci, device="cuda")
        w = torch.randn(K, K, Ci, Co, device="cuda") * 0.05
        for _ in range(3):
            y = e.conv2d(x, w, s, pad, **KW)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        t0.record()
        for _ in range(iters):
            y = e.conv2d(x, w, s, pad, **KW)
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / iters
        assert e.lib.fs_debug_conv_trace_reset() == 0
        y = e.conv2d(x, w, s, pad, **KW)
        torch.cuda.synchronize()
        buf = np.zeros((4096, 8), dtype=np.int64)
        rc = rd(buf.ctypes.data, 4096)
        assert rc == 0, rc
        live = buf[buf[:, 6] > 0]
        span = live[:, 6].max() - live[:, 0].min()
        life = live[:, 6] - live[:, 0]
        fl = 2.0 * y.numel() * K * K * Ci
        print("%s: %.1f us (%.1f TFLOP/s incl. epilogue kernel if split)  %d workgroups traced, launch span %d ticks"
              " -> %.1f ticks/us" % (nm, us, fl / us / 1e6, len(live), span, span / us), flush=True)
        names_ = ["prologue", "sweep", "commit", "barrier", "epilogue"]
        tot = life.mean()
        print("   workgroup lifetime mean %.0f (min %d max %d) ticks; rounds = span/lifetime = %.2f"
              % (tot, life.min(), life.max(), span / tot))
        for i, k in enumerate(names_):
            v = live[:, 1 + i]
            print("   %-9s mean %8.0f  (%5.1f%%)  min %7d max %7d" % (k, v.mean(), 100.0 * v.mean() / tot, v.min(), v.max()))
        hw = live[:, 7]
        print("   HW_ID wave slot histogram (bits 3:0): %s" % np.bincount((hw & 0xF).astype(int), minlength=4)[:8].tolist())
        first = live[np.argsort(live[:, 0])][:512]
        print("   ... of the 512 earliest workgroups: %s" % np.bincount((first[:, 7] & 0xF).astype(int), minlength=4)[:8].tolist())
        order = np.argsort(live[:, 0])
        starts = live[order, 0] - live[:, 0].min()
        print("   start offsets (ticks) pctl 0/25/50/75/100: %s" % np.percentile(starts, [0, 25, 50, 75, 100]).astype(int))


if __name__ == "__main__":
    main()
