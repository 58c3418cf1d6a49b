#!/usr/bin/env python
"""EXPERIMENT: do the memory-bound kernels of the split-bf16 pipeline (input / output transform) hide behind the GEMM of an independent half batch?
Two chains of L consecutive conv4_2-shaped layers (N = 16 each) on two HIP streams against one chain at N = 32 on one stream.  Same total work.
usage: micro_wino6_overlap.py   env: LAYERS=3 ITERS=10"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import _lib as L, engine  # noqa: E402


class StreamMem(engine.TorchMem):
    def __init__(self, stream):
        super().__init__()
        self._s = stream

    def stream(self):
        return self._s.cuda_stream


def make(e, N, C, layers):
    p = e.mem.ptr
    w = torch.randn(3, 3, C, C, device="cuda") * (2.0 / (9 * C)) ** 0.5
    bias = torch.randn(C, device="cuda")
    U = torch.empty(e.lib.fs_wino6_filter_bytes(C, C) // 4, device="cuda")
    L.check(e.lib, e.lib.fs_wino6_transform_filter(e.ctx, p(w), C, C, p(U)), "t")
    nb = e.lib.fs_wino6_workspace_bytes(N, 32, 32, C, C)
    ws = torch.empty(nb // 4, device="cuda")
    bufs = [torch.relu(torch.randn(N, 32, 32, C, device="cuda")) for _ in range(2)]
    descs = []
    for i in range(layers):
        d = L.fs_conv_desc()
        d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = N, 32, 32, C, C, 3, 3, 1
        d.pad_mode = L.FS_PAD_SAME
        d.x, d.w, d.y = p(bufs[i & 1]), p(w), p(bufs[(i + 1) & 1])
        d.bias, d.out_relu = p(bias), 1
        d.w_wino6, d.w6_ws, d.w6_ws_bytes = p(U), p(ws), nb
        t = ctypes.c_int()
        L.check(e.lib, e.lib.fs_conv2d_plan(ctypes.byref(d), ctypes.byref(t)), "plan")
        descs.append(d)
    return descs, (w, bias, U, ws, bufs)


def main():
    os.environ.setdefault("FS_WINO6_MINCC", "0")
    os.environ.setdefault("FS_WINO6_MINTILES", "1")
    layers, iters, C = int(os.environ.get("LAYERS", "3")), int(os.environ.get("ITERS", "10")), 512
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    e0, e1 = engine.Engine(mem=StreamMem(s0)), engine.Engine(mem=StreamMem(s1))
    full, k0 = make(e0, 32, C, layers)
    ha, k1 = make(e0, 16, C, layers)
    hb, k2 = make(e1, 16, C, layers)
    torch.cuda.synchronize()

    def run_full():
        for d in full:
            e0.lib.fs_conv2d_fwd(e0.ctx, ctypes.byref(d))

    def run_halves(offset):
        if offset:   # chain B starts when chain A's first layer is half way: its first launch waits on an event recorded after A's first conv
            pass
        for i in range(layers):
            e0.lib.fs_conv2d_fwd(e0.ctx, ctypes.byref(ha[i]))
            e1.lib.fs_conv2d_fwd(e1.ctx, ctypes.byref(hb[i]))

    for name, fn in (("one chain, N = 32", run_full), ("two chains, N = 16 + 16, two streams", lambda: run_halves(False))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(torch.cuda.default_stream())
        s0.wait_stream(torch.cuda.default_stream())
        s1.wait_stream(torch.cuda.default_stream())
        for _ in range(iters):
            fn()
        torch.cuda.default_stream().wait_stream(s0)
        torch.cuda.default_stream().wait_stream(s1)
        t1.record(torch.cuda.default_stream())
        torch.cuda.synchronize()
        print("%-40s %8.1f us per layer" % (name, 1e3 * t0.elapsed_time(t1) / iters / layers), flush=True)


if __name__ == "__main__":
    main()
