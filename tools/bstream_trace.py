#!/usr/bin/env python
"""Where a workgroup of the streaming bf16 kernel (fs_bstream.hip) spends its cycles -- tuning aid, needs the DEBUG build:

    python -c "from faststyle_amd import build as b; b.build(extra_flags=['-DFS_BSTREAM_TRACE'], out='exp/libbstrace.so', objdir='exp/build_bstrace')"
    FASTSTYLE_HIP_LIB=exp/libbstrace.so FS_BSTREAM_MASK=<bit of the instance> python tools/bstream_trace.py [N H W]

The trace holds the LAST bstream launch of a forward pass: with FS_BSTREAM_MASK = 1 << (instance - 1) only that instance
takes the streaming kernel (the others fall back to fs_bf16.hip), so the trace is that instance's (first) launch... the
LAST one of the instance for the residual convs.  Prints mean cycles per tile and phase (wave 0's view)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import ckpt, engine  # noqa: E402


def main():
    N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 1080, 1920)
    e = engine.Engine()
    rd = e.lib.fs_debug_conv_trace
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flat = e.mem.from_numpy(e.flatten_params(ckpt.load_checkpoint(os.path.join(root, "models", "starry_final.ckpt"))))
    x = torch.rand((N, H, W, 3), device="cuda") * 255.0
    for _ in range(2):
        e.tnet_forward(flat, x, bf16=True)
    torch.cuda.synchronize()
    assert e.lib.fs_debug_conv_trace_reset() == 0
    e.tnet_forward(flat, x, bf16=True)
    torch.cuda.synchronize()
    buf = np.zeros((512, 10), dtype=np.int64)
    assert rd(buf.ctypes.data, 512) == 0
    live = buf[buf[:, 8] > 0]
    life = (live[:, 8] - live[:, 0]).astype(np.float64)
    tiles = live[:, 9].astype(np.float64)
    print("mask %s: %d workgroups, %.1f tiles each, lifetime %.0f cycles = %.0f per tile" % (os.environ.get("FS_BSTREAM_MASK"), len(live), tiles.mean(), life.mean(), (life / tiles).mean()))
    for i, k in enumerate(["finalize+issue", "sweep", "barrier A", "commit", "epilogue write", "barrier B", "store"]):
        v = live[:, 1 + i] / tiles
        print("   %-15s %8.0f cycles per tile (%5.1f%%)" % (k, v.mean(), 100.0 * live[:, 1 + i].sum() / life.sum()))


if __name__ == "__main__":
    main()
