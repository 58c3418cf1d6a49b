#!/usr/bin/env python
"""Micro-benchmark of the split-bf16 F(4x4,3x3) pipeline (fs_wino6.hip, FS_WINO_V=6) against the fp32 F(4x4) kernel (fs_wino4t.hip) on the VGG16 launches of
the batch-32 training step, through fs_conv2d_fwd with caller-transformed filters (tuning aid).  HIP events around ITERS launches; the three kernels of the
pipeline are one fs_conv2d_fwd call (their split: rocprofv3 --kernel-trace --stats over this script).
usage: micro_wino6.py [name ...]   env: MODES=6,4t  ITERS=20  CHECK=1  FS_WINO6_CHUNK=<tiles>"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import _lib as L, engine  # noqa: E402

# name: (N, H, W, Cin, Cout, form)
CASES = {
    "conv4_2_fwd": (32, 32, 32, 512, 512, "bias_relu"),
    "conv4_2_dgrad": (32, 32, 32, 512, 512, "mask"),
    "conv4_1_fwd": (32, 32, 32, 256, 512, "bias_relu"),
    "conv4_1_dgrad": (32, 32, 32, 512, 256, "raw"),
    "conv3_2_fwd": (64, 64, 64, 256, 256, "bias_relu"),
    "conv3_2_dgrad": (32, 64, 64, 256, 256, "mask"),
    "conv2_2_fwd": (64, 128, 128, 128, 128, "bias_relu"),
}


def main():
    names = sys.argv[1:] or ["conv4_2_fwd", "conv4_2_dgrad", "conv4_1_fwd", "conv4_1_dgrad", "conv3_2_fwd"]
    modes = os.environ.get("MODES", "6,4t").split(",")
    iters = int(os.environ.get("ITERS", "20"))
    os.environ.setdefault("FS_WINO6_MINCC", "0")
    os.environ.setdefault("FS_WINO4T_TB", "2")
    e = engine.Engine()
    p = e.mem.ptr
    for nm in names:
        N, H, W, Cin, Cout, form = CASES[nm]
        x = torch.relu(torch.randn(N, H, W, Cin, device="cuda")) * 50
        w = torch.randn(3, 3, Cin, Cout, device="cuda") * (2.0 / (9 * Cin)) ** 0.5
        bias = torch.randn(Cout, device="cuda")
        mask = torch.randn(N, H, W, Cout, device="cuda")
        outs = {}
        for mode in modes:
            d = L.fs_conv_desc()
            d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = N, H, W, Cin, Cout, 3, 3, 1
            d.pad_mode = L.FS_PAD_SAME
            d.x, d.w = p(x), p(w)
            if form == "bias_relu":
                d.bias, d.out_relu = p(bias), 1
            elif form == "mask":
                d.mask_src = p(mask)
            keep = []
            if mode == "6":
                U = torch.empty(e.lib.fs_wino6_filter_bytes(Cin, Cout) // 4, device="cuda")
                L.check(e.lib, e.lib.fs_wino6_transform_filter(e.ctx, p(w), Cin, Cout, p(U)), "wino6 transform")
                nb = e.lib.fs_wino6_workspace_bytes(N, H, W, Cin, Cout)
                ws = torch.empty(nb // 4, device="cuda")
                d.w_wino6, d.w6_ws, d.w6_ws_bytes = p(U), p(ws), nb
                keep += [U, ws]
            else:
                U = e.mem.empty((36, Cin, Cout))
                L.check(e.lib, e.lib.fs_wino4t_transform_filter(e.ctx, p(w), Cin, Cout, p(U)), "wino4t transform")
                d.w_wino4t = p(U)
                keep.append(U)
            tiles = ctypes.c_int()
            L.check(e.lib, e.lib.fs_conv2d_plan(ctypes.byref(d), ctypes.byref(tiles)), "plan")
            y = e.mem.empty((N, H, W, Cout))
            d.y = p(y)
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
            fl = 2.0 * N * H * W * 9 * Cin * Cout / 4
            outs[mode] = y
            print("%-16s mode %-2s %8.1f us   executed (fp32-equivalent F(4x4) products) %6.2f TFLOP/s%s" % (
                nm, mode, ms * 1e3, fl / ms / 1e9, "   = %.0f TFLOP/s of bf16 products issued" % (6 * fl / ms / 1e9) if mode == "6" else ""), flush=True)
        if os.environ.get("CHECK") and len(outs) > 1:
            ks = list(outs)
            ref = outs[ks[0]].double()
            for k in ks[1:]:
                err = float((outs[k].double() - ref).abs().max() / ref.abs().max())
                print("   max |mode %s - mode %s| / max = %.2e" % (k, ks[0], err))


if __name__ == "__main__":
    main()
