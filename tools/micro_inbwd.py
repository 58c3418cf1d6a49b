#!/usr/bin/env python
"""Micro-benchmark of the instance-norm backward (fs_instnorm_bwd: partial sums, final, apply) at the training shapes (tuning aid).
usage: micro_inbwd.py   env: ITERS=50  FS_INBWD_CHUNK=<px>"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import _lib as L, engine  # noqa: E402

CASES = {"res_b32": (32, 74, 74, 64), "res_b4": (4, 74, 74, 64), "up1_b32": (32, 128, 128, 32), "init1_b32": (32, 128, 128, 32), "up2_b32": (32, 256, 256, 16)}


def main():
    iters = int(os.environ.get("ITERS", "50"))
    e = engine.Engine()
    p = e.mem.ptr
    for nm, (N, H, W, C) in CASES.items():
        g, z = torch.randn(N, H, W, C, device="cuda"), torch.randn(N, H, W, C, device="cuda")
        mean, rstd = torch.randn(N, C, device="cuda") * 0.1, torch.rand(N, C, device="cuda") + 0.5
        a, b = torch.rand(N, C, device="cuda") + 0.5, torch.randn(N, C, device="cuda") * 0.1
        dz = torch.empty_like(z)
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        nbytes = e.lib.fs_instnorm_bwd_workspace_bytes(N, H * W, C)
        ws = torch.empty(nbytes // 4, device="cuda")
        call = lambda: e.lib.fs_instnorm_bwd(e.ctx, p(g), p(z), p(mean), p(rstd), p(a), p(b), 1, N, H * W, C, p(dz), p(dg), p(db), p(ws), nbytes)
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            call()
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) / iters * 1e3
        mb = N * H * W * C * 4 / 1e6
        print("%-10s %6.1f us for the three launches: %5.1f MB tensor, 5 tensor passes (2 reads + 2 reads, 1 write) -> %.2f TB/s "
              "(the operands of a repeated call sit in the 256 MB Infinity Cache: an upper bound for the step)" % (nm, us, mb, 5 * mb / us), flush=True)


if __name__ == "__main__":
    main()
