#!/usr/bin/env python
"""Where a workgroup of the half-item Winograd kernel (fs_wino2h.hip) spends its cycles -- tuning aid, needs the DEBUG build:

    python -c "from faststyle_amd import build as b; b.build(extra_flags=['-DFS_WINO2H_TRACE'], out='exp/libw2htrace.so', objdir='exp/build_w2htrace')"
    FASTSTYLE_HIP_LIB=exp/libw2htrace.so python tools/w2h_trace.py [batch]

Runs the transform-net forward at 256x256 (default batch 4: the residual convs take the half-item kernel) and prints the
phase cycle counts of the LAST wino2h launch (resblock_4, second conv): prologue, sweeps, patch commits, barrier waits,
epilogue, workgroup lifetime, launch span."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import engine, im_transf_net  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    e = engine.Engine()
    rd = e.lib.fs_debug_conv_trace
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    flat = e.mem.from_numpy(e.flatten_params(im_transf_net.initial_variables(seed=0), scope=""))
    x = torch.rand((B, 256, 256, 3), device="cuda") * 255.0
    for _ in range(3):
        e.tnet_forward(flat, x, save_for_bwd=True)
    torch.cuda.synchronize()
    assert e.lib.fs_debug_conv_trace_reset() == 0
    e.tnet_forward(flat, x, save_for_bwd=True)
    torch.cuda.synchronize()
    buf = np.zeros((4096, 8), dtype=np.int64)
    assert rd(buf.ctypes.data, 4096) == 0
    live = buf[buf[:, 6] > 0]
    life = live[:, 6] - live[:, 0]
    span = live[:, 6].max() - live[:, 0].min()
    print("batch %d: %d workgroups traced; lifetime mean %.0f (min %d max %d) ticks, launch span %d ticks" % (B, len(live), life.mean(), life.min(), life.max(), span))
    for i, k in enumerate(["prologue", "sweeps", "patch commit", "barrier wait", "epilogue"]):
        v = live[:, 1 + i]
        print("   %-13s mean %8.0f  (%5.1f%% of lifetime)  min %7d max %7d" % (k, v.mean(), 100.0 * v.mean() / life.mean(), v.min(), v.max()))
    starts = np.sort(live[:, 0]) - live[:, 0].min()
    print("   start offsets (ticks) pctl 0/25/50/75/100: %s" % np.percentile(starts, [0, 25, 50, 75, 100]).astype(int))
    ends = np.sort(live[:, 6]) - live[:, 0].min()
    print("   end offsets   (ticks) pctl 0/25/50/75/100: %s" % np.percentile(ends, [0, 25, 50, 75, 100]).astype(int))
    # tick rate: time a long kernel-free interval is not available here; s_memtime counts at 100 MHz on gfx950


if __name__ == "__main__":
    main()
