#!/usr/bin/env python
"""Probe (round 6): one batch-B train step against TWO half-batch steps on two streams (two engines: own context, own workspaces), graph-replayed.
usage: split_step_probe.py [B]   (default 4)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import engine
from oracle import perceptual, tnet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(0)
Wv = perceptual.synthetic_vgg_weights(seed=3)
P = tnet.init_params(0)
style = rng.uniform(0, 255, (1, 256, 256, 3)).astype(np.float32)


def make(n):
    e = engine.Engine()
    e.vgg_load(Wv)
    cfg = engine.default_loss_cfg()
    flat = e.mem.from_numpy(e.flatten_params(P, scope=""))
    tg = e.style_targets(e.mem.from_numpy(style), cfg)
    x = torch.rand((n, 256, 256, 3), device="cuda") * 255
    grads = e.mem.zeros(flat.shape)

    def fb():
        y = e.tnet_forward(flat, x, save_for_bwd=True)
        losses, dy = e.perceptual_loss(y, x, tg, cfg)
        e.tnet_backward(flat, x, dy, grads=grads)
        return losses
    return e, fb, (flat, tg, x, grads)


def capture(fb, st):
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(3):
            fb()
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            fb()
    torch.cuda.synchronize()
    return g


def bench(graphs, streams, iters=100):
    for _ in range(5):
        for g, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for g, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g.replay()
        # (join per step: the gradient sum + Adam would follow here)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
e_full, fb_full, keep_full = make(B)
g_full = capture(fb_full, s0)
t_full = bench([g_full], [s0])
eA, fbA, keepA = make(B // 2)
eB, fbB, keepB = make(B // 2)
gA, gB = capture(fbA, s0), capture(fbB, s1)
t_half1 = bench([gA], [s0])
t_split = bench([gA, gB], [s0, s1])
print("batch %d: one step %.3f ms | one half-batch step alone %.3f ms | two half-batch steps on two streams %.3f ms" % (B, t_full, t_half1, t_split))
