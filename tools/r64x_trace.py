#!/usr/bin/env python
"""Where a conv_r64x workgroup spends its cycles (tuning aid; needs the -DFS_R64X_TRACE build: tools/build_abl2.sh fs_cstream.hip FS_R64X_TRACE 1,
FASTSTYLE_HIP_LIB=exp/libabl_1.so FS_CSTREAM_R64X_ALL=1).  Wave 0 timestamps its phases with the shader clock."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import engine  # noqa: E402
from tools.micro_conv import CASES  # noqa: E402

e = engine.Engine()
rd = e.lib.fs_debug_r64x_trace
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
for nm in sys.argv[1:] or ["res_720p", "res_b32"]:
    N, H, W, Ci, Co, K, s, pad = CASES[nm]
    x = torch.randn(N, H, W, Ci, device="cuda")
    w = torch.randn(K, K, Ci, Co, device="cuda") * 0.05
    for _ in range(3):
        e.conv2d(x, w, s, pad, want_stats=True)
    torch.cuda.synchronize()
    buf = np.zeros((4096, 8), dtype=np.int64)
    assert rd(buf.ctypes.data, 4096) == 0
    live = buf[buf[:, 7] > 0]
    names = ["issue", "sweep", "barrier A", "commit", "epilogue", "barrier B", "prologue"]
    t = live[:, 7].astype(float)
    print("%s: %d workgroups, %.1f tiles each; per TILE cycles (mean over workgroups):" % (nm, len(live), t.mean()))
    for i, n in enumerate(names[:6]):
        print("   %-10s %8.0f" % (n, (live[:, i] / t).mean()))
    print("   %-10s %8.0f (once)   total per workgroup %8.0f" % ("prologue", live[:, 6].mean(), live[:, :7].sum(axis=1).mean()))
