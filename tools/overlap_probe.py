#!/usr/bin/env python
"""Can a second stream's dense VGG work hide inside the (latency-bound, small-grid) transform-net forward?
Times tnet_forward(b4 256^2) and a VGG+Gram pass on a 512x512 image alone and concurrently (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import engine, im_transf_net, vgg16  # noqa: E402

ea, eb = engine.Engine(), engine.Engine()
flat = ea.mem.from_numpy(ea.flatten_params(im_transf_net.initial_variables(seed=0), scope=""))
x = torch.rand((4, 256, 256, 3), device="cuda") * 255
W = vgg16.synthetic_weights(3)
eb.vgg_load(W)
img = torch.rand((1, 512, 512, 3), device="cuda") * 255
cfg = engine.default_loss_cfg()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
K = 20


def run(a, b):
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    sa.wait_stream(torch.cuda.current_stream())
    sb.wait_stream(torch.cuda.current_stream())
    for _ in range(K):
        if a:
            with torch.cuda.stream(sa):
                ea.tnet_forward(flat, x, save_for_bwd=True)
        if b:
            with torch.cuda.stream(sb):
                eb.style_targets(img, cfg)
    torch.cuda.current_stream().wait_stream(sa)
    torch.cuda.current_stream().wait_stream(sb)
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / K


for _ in range(2):
    run(True, True)
ta, tb, tab = run(True, False), run(False, True), run(True, True)
print("tnet fwd alone %.3f ms, vgg pass alone %.3f ms, concurrent %.3f ms (sum %.3f, hidden %.3f ms)" %
      (ta, tb, tab, ta + tb, ta + tb - tab))
