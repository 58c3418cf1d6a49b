#!/usr/bin/env python
"""Micro-benchmark of the VGG16 3x3 conv launches at the training shapes (tuning aid): fs_conv2d_fwd through the Winograd
F(4x4,3x3) kernel (fs_wino4.hip) against the F(2x2,3x3) kernels / the direct kernel, filters transformed ONCE outside the timed
loop, HIP events around ITERS launches.   usage: micro_wino4.py [name ...]   env: MODES=4,2,0  ITERS=20  CHECK=1"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import _lib as L, engine  # noqa: E402

# name: (N, H, W, Cin, Cout, epilogue)   -- the launches of a batch-32 train step (forward at 2N = 64 up to conv3_3)
CASES = {
    "conv1_2_fwd_n64": (64, 256, 256, 64, 64, "pool"),
    "conv2_1_fwd_n64": (64, 128, 128, 64, 128, "relu"),
    "conv2_2_fwd_n64": (64, 128, 128, 128, 128, "pool"),
    "conv3_1_fwd_n64": (64, 64, 64, 128, 256, "relu"),
    "conv3_2_fwd_n64": (64, 64, 64, 256, 256, "relu"),
    "conv4_1_fwd_n32": (32, 32, 32, 256, 512, "relu"),
    "conv4_2_fwd_n32": (32, 32, 32, 512, 512, "relu"),
    "conv4_2_dgrad_n32": (32, 32, 32, 512, 512, "mask"),
    "conv3_2_dgrad_n32": (32, 64, 64, 256, 256, "mask"),
    "conv2_2_dgrad_n32": (32, 128, 128, 128, 128, "mask"),
    "conv1_2_dgrad_n32": (32, 256, 256, 64, 64, "none"),
    "conv4_2_fwd_n4": (4, 32, 32, 512, 512, "relu"),
    "conv3_2_fwd_n8": (8, 64, 64, 256, 256, "relu"),
    "conv1_2_fwd_n8": (8, 256, 256, 64, 64, "pool"),
}


def main():
    names = sys.argv[1:] or list(CASES)
    modes = [m if m == "4t" else int(m) for m in os.environ.get("MODES", "4t,4").split(",")]   # 4t: fs_wino4t.hip, 4: fs_wino4.hip, 2: F(2x2), 0: direct
    iters = int(os.environ.get("ITERS", "20"))
    e = engine.Engine()
    p = e.mem.ptr
    for nm in names:
        N, H, W, Ci, Co, epi = CASES[nm]
        x = torch.relu(torch.randn(N, H, W, Ci, device="cuda")) * 10.0
        w = torch.randn(3, 3, Ci, Co, device="cuda") * (2.0 / (9 * Ci)) ** 0.5
        bias = torch.randn(Co, device="cuda")
        mask = torch.randn(N, H, W, Co, device="cuda")
        split_note = ""
        outs = {}
        for mode in modes:
            d = L.fs_conv_desc()
            d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = N, H, W, Ci, Co, 3, 3, 1
            d.pad_mode = L.FS_PAD_SAME
            d.x, d.w = p(x), p(w)
            keep = []
            if epi in ("relu", "pool"):
                d.bias, d.out_relu = p(bias), 1
            if epi == "mask":
                d.mask_src = p(mask)
            if mode == "4t":
                U = e.mem.empty((36, Ci, Co))
                L.check(e.lib, e.lib.fs_wino4t_transform_filter(e.ctx, p(w), Ci, Co, p(U)), "wino4t transform")
                d.w_wino4t = p(U)
                keep.append(U)
            elif mode == 4:
                U = e.mem.empty((36, Ci, Co))
                L.check(e.lib, e.lib.fs_wino4_transform_filter(e.ctx, p(w), Ci, Co, p(U)), "wino4 transform")
                d.w_wino4 = p(U)
                keep.append(U)
            elif mode == 2:
                U = e.mem.empty((16, Ci, Co))
                L.check(e.lib, e.lib.fs_wino_transform_filter(e.ctx, p(w), Ci, Co, p(U)), "wino transform")
                d.w_wino = p(U)
                keep.append(U)
            tiles = ctypes.c_int()
            y = e.mem.empty((N, H, W, Co))
            d.y = p(y)
            if epi == "pool" and mode in (2, 4, "4t") and not (mode == 2 and Ci > 128):
                pool = e.mem.empty((N, H // 2, W // 2, Co))
                d.pool_out = p(pool)
                keep.append(pool)
            L.check(e.lib, e.lib.fs_conv2d_plan(ctypes.byref(d), ctypes.byref(tiles)), "plan")
            for _ in range(2):
                L.check(e.lib, e.lib.fs_conv2d_fwd(e.ctx, ctypes.byref(d)), "fwd")
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(iters):
                e.lib.fs_conv2d_fwd(e.ctx, ctypes.byref(d))
            t1.record()
            torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / iters
            fl = 2.0 * N * H * W * 9 * Ci * Co
            outs[mode] = y
            print("%-18s mode %-2s %8.1f us  direct-equivalent %7.2f TFLOP/s  executed %6.2f TFLOP/s" % (
                nm, mode, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 * ({4: 0.25, "4t": 0.25, 2: 16.0 / 36.0}.get(mode, 1.0))), flush=True)
            buf = None
            if mode == 4 and hasattr(e.lib, "fs_debug_wino4_trace"):   # -DFS_WINO4_TRACE build: phases of the last launch
                import numpy as np
                buf = np.zeros((4096, 8), dtype=np.int64)
                e.lib.fs_debug_wino4_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
                assert e.lib.fs_debug_wino4_trace(buf.ctypes.data, 4096) == 0
            elif mode == "4t":                                          # -DFS_WINO4T_TRACE build
                from micro_wino4t import read_wino4t_trace
                buf = read_wino4t_trace(e.lib)
            if buf is not None:
                live = buf[buf[:, 6] > 0]
                life = (live[:, 6] - live[:, 0]).astype(float)
                steps = live[:, 5].astype(float)
                print("   trace: %d workgroups, lifetime %.0f ticks, steps/wg %.1f, items/wg %.1f | prologue %.1f%% sweeps %.1f%% (%.0f ticks per step) "
                      "barrier %.1f%% epilogue %.1f%% (%.0f ticks per item)" % (
                          len(live), life.mean(), steps.mean(), live[:, 7].mean(), 100 * live[:, 1].mean() / life.mean(),
                          100 * live[:, 2].mean() / life.mean(), (live[:, 2] / steps).mean(), 100 * live[:, 3].mean() / life.mean(),
                          100 * live[:, 4].mean() / life.mean(), (live[:, 4] / live[:, 7]).mean()))
        if os.environ.get("CHECK") and len(outs) > 1:
            ks = list(outs)
            ref = outs[ks[0]].double()
            for k in ks[1:]:
                err = float((outs[k].double() - ref).abs().max() / ref.abs().max())
                print("   max |mode %s - mode %s| / max = %.2e" % (k, ks[0], err))


if __name__ == "__main__":
    main()
