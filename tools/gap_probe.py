#!/usr/bin/env python
"""How long does one DEPENDENT tiny kernel cost on this stack (eager vs hipGraph replay)?
Decides whether launch count is worth optimising (tuning aid)."""
import torch

x = torch.zeros(1024, device="cuda")
n = 400


def chain():
    for _ in range(n):
        x.add_(1.0)


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3 / n


print("eager : %.2f us per dependent tiny kernel" % timeit(chain))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain()
print("graph : %.2f us per dependent tiny kernel" % timeit(g.replay))
