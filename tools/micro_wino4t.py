#!/usr/bin/env python
"""Micro-benchmark of the transform net's residual 3x3 convs (64 -> 64) at the training / inference shapes (tuning aid):
fs_conv2d_fwd through the 16-tile Winograd F(4x4,3x3) kernel (fs_wino4t.hip) against the F(2x2,3x3) kernels, in the forms the
network launches -- VALID + instance norm on load + statistics (forward), 'full' padding + residual gradient (input gradient).
Filters transformed ONCE outside the timed loop, HIP events around ITERS launches.
usage: micro_wino4t.py [name ...]   env: MODES=4t,2  ITERS=20  CHECK=1"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import _lib as L, engine  # noqa: E402

# name: (N, H, W, form)   H, W = conv INPUT extent
CASES = {
    "720p_fwd_aff": (1, 190, 330, "fwd_aff"),
    "720p_fwd": (1, 192, 332, "fwd"),
    "b32_fwd_aff": (32, 74, 74, "fwd_aff"),
    "b32_fwd": (32, 76, 76, "fwd"),
    "b32_dgrad_add": (32, 72, 72, "dgrad_add"),
    "b32_dgrad": (32, 74, 74, "dgrad"),
    "b4_fwd_aff": (4, 74, 74, "fwd_aff"),
    "b4_dgrad_add": (4, 72, 72, "dgrad_add"),
    "1080p_b8_fwd_aff": (8, 280, 490, "fwd_aff"),
}


def read_wino4t_trace(lib):
    """-DFS_WINO4T_TRACE builds: the phase counters of the LAST wino4t launch -- the kernel's instantiations live in four translation
    units with a buffer each (fs_debug_wino4t_trace_1a / 1b / 2a / 2b); the buffer holding the latest end timestamp is the one."""
    import numpy as np
    best = None
    for unit in ("1a", "1b", "1c", "2a", "2b"):
        fn = getattr(lib, "fs_debug_wino4t_trace_" + unit, None)
        if fn is None:
            continue
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
        buf = np.zeros((4096, 8), dtype=np.int64)
        if fn(buf.ctypes.data, 4096) == 0 and (best is None or buf[:, 6].max() > best[:, 6].max()):
            best = buf
    return best


def main():
    names = sys.argv[1:] or list(CASES)
    modes = os.environ.get("MODES", "4t,2").split(",")
    iters = int(os.environ.get("ITERS", "20"))
    e = engine.Engine()
    p = e.mem.ptr
    for nm in names:
        N, H, W, form = CASES[nm]
        x = torch.randn(N, H, W, 64, device="cuda")
        w = torch.randn(3, 3, 64, 64, device="cuda") * (2.0 / (9 * 64)) ** 0.5
        ia = torch.rand(N, 64, device="cuda") + 0.5
        ib = torch.randn(N, 64, device="cuda") * 0.1
        add = torch.randn(N, H - 2, W - 2, 64, device="cuda") if form == "dgrad_add" else None
        outs = {}
        for mode in modes:
            d = L.fs_conv_desc()
            d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = N, H, W, 64, 64, 3, 3, 1
            keep = []
            if form.startswith("fwd"):
                d.pad_mode = L.FS_PAD_VALID
                Ho, Wo = H - 2, W - 2
                if form == "fwd_aff":
                    d.in_a, d.in_b, d.in_per_sample, d.in_relu = p(ia), p(ib), 1, 1
            else:
                d.pad_mode = L.FS_PAD_EXPLICIT
                Ho, Wo = H + 2, W + 2
                d.pad_t, d.pad_l, d.Ho, d.Wo = 2, 2, Ho, Wo
                if form == "dgrad_add":
                    d.add_src, d.add_pad = p(add), 2
            d.x, d.w = p(x), p(w)
            U = e.mem.empty((36, 64, 64))
            if mode == "4t":
                L.check(e.lib, e.lib.fs_wino4t_transform_filter(e.ctx, p(w), 64, 64, p(U)), "wino4t transform")
                d.w_wino4t = p(U)
            elif mode == "2":
                L.check(e.lib, e.lib.fs_wino_transform_filter(e.ctx, p(w), 64, 64, p(U)), "wino transform")
                d.w_wino = p(U)
            tiles = ctypes.c_int()
            L.check(e.lib, e.lib.fs_conv2d_plan(ctypes.byref(d), ctypes.byref(tiles)), "plan")
            y = e.mem.empty((N, Ho, Wo, 64))
            d.y = p(y)
            if form.startswith("fwd"):
                st = e.mem.empty((N, tiles.value, 64, 3))
                d.stats = p(st)
                keep.append(st)
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
            fl = 2.0 * N * Ho * Wo * 9 * 64 * 64
            outs[mode] = y
            print("%-18s mode %-2s %8.1f us  direct-equivalent %7.2f TFLOP/s  executed %6.2f TFLOP/s   (%d items of the plan)" % (
                nm, mode, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 * ({"4t": 0.25, "2": 16.0 / 36.0}.get(mode, 1.0)), N * tiles.value), flush=True)
            buf = read_wino4t_trace(e.lib) if mode == "4t" else None   # -DFS_WINO4T_TRACE build: phases of the last launch
            if buf is not None:
                t_end = buf[:, 6].max()
                live = buf[(buf[:, 6] > 0) & (buf[:, 6] > t_end - 50_000_000)]   # (rows of earlier, larger launches stay in the buffer)
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
