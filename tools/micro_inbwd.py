#!/usr/bin/env python
"""Micro-benchmark of the instance-norm backward (fs_instnorm_bwd) at the shapes of a training step (tuning aid).

Both forms through the same entry point, alternated in one process:  FS_INBWD_REC=1 (round 5: partial sums -> records, final reduction in the
apply kernel's prologue: 2 launches + one in_bwd_params launch per STEP)  and  FS_INBWD_REC=0 (partial, final, apply: 3 launches).  Every case
rotates over enough tensor sets to exceed the 256 MB Infinity Cache (SETS_MB, default 600), so the figures are HBM figures as in the step, where
z was written long before and g just before.
usage: micro_inbwd.py [case ...]   env: ITERS=20  SETS_MB=600  FS_INBWD_CHUNK=<px>  FS_INBWD_REC_MAXT=<records per sample>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import engine  # noqa: E402

CASES = {"res_b32": (32, 74, 74, 64), "res82_b32": (32, 82, 82, 64), "res_b4": (4, 74, 74, 64), "init0_b32": (32, 336, 336, 16), "init1_b32": (32, 168, 168, 32),
         "up0_b32": (32, 128, 128, 32), "up1_b32": (32, 256, 256, 16), "init0_b4": (4, 336, 336, 16), "up1_b4": (4, 256, 256, 16)}


def main():
    iters = int(os.environ.get("ITERS", "20"))
    sets_mb = float(os.environ.get("SETS_MB", "600"))
    e = engine.Engine()
    p = e.mem.ptr
    names = sys.argv[1:] or list(CASES)
    for nm in names:
        N, H, W, C = CASES[nm]
        mb = N * H * W * C * 4 / 1e6
        K = max(2, int(sets_mb / (3 * mb)) + 1)
        sets = [(torch.randn(N, H, W, C, device="cuda"), torch.randn(N, H, W, C, device="cuda"), torch.empty(N, H, W, C, device="cuda")) for _ in range(K)]
        mean, rstd = torch.randn(N, C, device="cuda") * 0.1, torch.rand(N, C, device="cuda") + 0.5
        a, b = torch.rand(N, C, device="cuda") + 0.5, torch.randn(N, C, device="cuda") * 0.1
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        nbytes = e.lib.fs_instnorm_bwd_workspace_bytes(N, H * W, C)
        ws = torch.empty(nbytes // 4, device="cuda")
        res = {}
        for rec in (1, 0, 1, 0):
            os.environ["FS_INBWD_REC"] = str(rec)
            e.lib.fs_debug_reload_env()
            call = lambda k: e.lib.fs_instnorm_bwd(e.ctx, p(sets[k][0]), p(sets[k][1]), p(mean), p(rstd), p(a), p(b), 1, N, H * W, C, p(sets[k][2]), p(dg), p(db),
                                                   p(ws), nbytes)
            for k in range(min(K, 3)):
                assert call(k) == 0
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for i in range(iters):
                call(i % K)
            t1.record()
            torch.cuda.synchronize()
            res.setdefault(rec, []).append(t0.elapsed_time(t1) / iters * 1e3)
        f = lambda v: "%6.1f / %6.1f us (%.2f TB/s over 5 tensor passes)" % (v[0], v[1], 5 * mb / min(v))
        print("%-10s %6.1f MB x %d sets | records: %s | three launches: %s" % (nm, mb, K, f(res[1]), f(res[0])), flush=True)


if __name__ == "__main__":
    main()
