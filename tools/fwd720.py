#!/usr/bin/env python
"""Forward-only loop for profiling: fwd720.py [H W [N [bf16]]] (default 720 1280 1; BASELINE configs 2 and 5)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import ckpt, engine  # noqa: E402

e = engine.Engine()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = ckpt.load_checkpoint(os.path.join(root, "models", "starry_final.ckpt"))
flat = e.mem.from_numpy(e.flatten_params(W))
H, Wd = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (720, 1280)))
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1
BF16 = len(sys.argv) > 4 and sys.argv[4] == "bf16"      # ("fp32": the default path)
x = torch.rand((N, H, Wd, 3), device="cuda") * 255
for _ in range(3):
    e.tnet_forward(flat, x, bf16=BF16, frozen=True)   # one checkpoint, many frames (stylize_image.py / stylize_webcam.py)
torch.cuda.synchronize()
t = time.perf_counter()
it = 20
for _ in range(it):
    e.tnet_forward(flat, x, bf16=BF16, frozen=True)   # one checkpoint, many frames (stylize_image.py / stylize_webcam.py)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / it
print("forward %s %dx%dx%d: %.3f ms  %.1f fps" % ("bf16" if BF16 else "fp32", N, H, Wd, dt * 1e3, N / dt))
