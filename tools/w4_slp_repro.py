#!/usr/bin/env python
"""Reproducer / guard for the two things learnt about the F(4x4,3x3) kernels on the GPU (round 4), on the product library:

1. ROCm 7.2 clang's SLP vectoriser and fs_wino4.hip.  Built WITH -fslp-vectorize two of the kernel's three epilogue instantiations return wrong
   values on gfx950 -- a few per cent of the elements, always lanes 12..15 of a row of 16 and the odd channel of a pair -- while the CPU
   emulator build of the same source and the third, equally packed instantiation are right.  The first loop runs the raw / bias + ReLU /
   consumer-mask forms against the float64 oracle and prints WHERE the bad elements sit (rows, columns, channels).  faststyle_amd/build.py
   compiles every MFMA translation unit with -fno-slp-vectorize since round 5 (packed fp32 beside fp32 matrix instructions is an anti-lever
   anyway); to reproduce:  python -c "from faststyle_amd import build; build.build(force=True, extra_flags=['-fslp-vectorize'], out='exp/libslp.so',
   objdir='exp/obj_slp')"  then on the GPU  FASTSTYLE_HIP_LIB=exp/libslp.so python tools/w4_slp_repro.py .
2. Per-sample results of the kernel do not depend on the batch a sample rides in (items are per sample; only the item -> workgroup map changes):
   the second loop compares a sample alone with the same sample in a batch of two, bit for bit -- the reason the data-parallel identity tests
   (tests/test_path_parity.py) can attribute their residual to the instance-norm / filter-gradient partial-sum partitions.

GPU only:  gpurun -- 'python tools/w4_slp_repro.py'"""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.backends import get_engine
from oracle import nnops
eng = get_engine("hip")
rng = np.random.default_rng(12)
for (xs, cout, epi) in [((1, 16, 32, 4), 64, None), ((1, 16, 32, 8), 64, None), ((1, 16, 32, 8), 64, "relu"), ((1, 16, 32, 8), 64, "mask"), ((2, 21, 37, 8), 64, "relu"), ((1, 32, 64, 64), 128, None)]:
    x = rng.standard_normal(xs).astype(np.float32)
    w = (rng.standard_normal((3, 3, xs[3], cout)) * 0.1).astype(np.float32)
    kw = {}
    want = nnops.conv2d(x.astype(np.float64), w.astype(np.float64), 1, "SAME")
    if epi == "relu":
        b = rng.standard_normal(cout).astype(np.float32)
        kw = dict(bias=eng.mem.from_numpy(b), out_relu=1)
        want = np.maximum(want + b, 0)
    if epi == "mask":
        m = rng.standard_normal(want.shape).astype(np.float32)
        kw = dict(mask_src=eng.mem.from_numpy(m))
        want = np.where(m > 0, want, 0)
    y = eng.mem.to_numpy(eng.conv2d(eng.mem.from_numpy(x), eng.mem.from_numpy(w), 1, "SAME", winograd=4, **kw))
    err = np.abs(y - want)
    print(xs, cout, epi, "max err %.3e rel %.3e nan %d badfrac %.4f" % (err.max(), err.max() / np.abs(want).max(), np.isnan(y).sum(), (err > 1e-3).mean()))
    if (err > 1e-3).any():
        bad = err > 1e-3
        print("  bad by row", np.round(bad.mean(axis=(0, 2, 3)), 2)[:20])
        print("  bad by col", np.round(bad.mean(axis=(0, 1, 3)), 2)[:40])
        print("  bad by ch ", np.round(bad.mean(axis=(0, 1, 2)), 2)[:64])
        i = np.argwhere(bad)[0]
        print("  first bad", i, y[tuple(i)], want[tuple(i)])
# per-sample results must not depend on the batch they ride in (items are per sample; only the item -> workgroup map changes)
import torch
for (H, W, ci, co, epi) in [(256, 256, 64, 64, "relu"), (64, 64, 256, 256, "relu"), (32, 32, 512, 512, "mask"), (128, 128, 128, 64, None)]:
    g = torch.Generator(device="cuda").manual_seed(5)
    x1 = torch.randn((1, H, W, ci), device="cuda", generator=g)
    w = torch.randn((3, 3, ci, co), device="cuda", generator=g) * 0.05
    kw1, kw2 = {}, {}
    if epi == "relu":
        b = torch.randn((co,), device="cuda", generator=g)
        kw1 = kw2 = dict(bias=b, out_relu=1)
    if epi == "mask":
        m = torch.randn((1, H, W, co), device="cuda", generator=g)
        kw1, kw2 = dict(mask_src=m), dict(mask_src=torch.cat([m, m]).contiguous())
    y1 = eng.conv2d(x1, w, 1, "SAME", winograd=4, **kw1).clone()
    y2 = eng.conv2d(torch.cat([x1, x1]).contiguous(), w, 1, "SAME", winograd=4, **kw2).clone()
    y1b = eng.conv2d(x1, w, 1, "SAME", winograd=4, **kw1).clone()
    print((H, W, ci, co, epi), "batch-1 twice equal:", bool(torch.equal(y1, y1b)), " batch-2 halves equal:", bool(torch.equal(y2[0], y2[1])),
          " batch-2 == batch-1:", bool(torch.equal(y2[0], y1[0])), " max diff %.3e" % float((y2[0] - y1[0]).abs().max()))
