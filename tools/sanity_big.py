#!/usr/bin/env python
"""Large-shape sanity run (BASELINE config 4/5 shapes in fp32): finite outputs, timing."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import ckpt, engine, im_transf_net, trainer, utils, vgg16  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
e = engine.Engine()
W = ckpt.load_checkpoint(os.path.join(root, "models", "starry_final.ckpt"))
flat = e.mem.from_numpy(e.flatten_params(W))
for (N, H, Wd) in [(8, 1080, 1920), (1, 2160, 3840), (3, 301, 517)]:
    x = torch.rand((N, H, Wd, 3), device="cuda") * 255
    y = e.tnet_forward(flat, x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        y = e.tnet_forward(flat, x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print("fwd %dx%dx%d -> %s finite=%s range=[%.2f,%.2f]  %.2f ms (%.1f fps)" % (
        N, H, Wd, tuple(y.shape), bool(torch.isfinite(y).all()), float(y.min()), float(y.max()), dt * 1e3, N / dt), flush=True)
    del x, y
    e._tnet_ws = {}
    torch.cuda.empty_cache()
style = utils.imread(os.path.join(root, "style_images", "starry_night_crop.jpg")).astype(np.float32)[None]
params = e.flatten_params(im_transf_net.initial_variables(seed=0), scope="")
tr = trainer.Trainer(e, params, vgg16.synthetic_weights(3), style, use_graph=True)
for B in (32, 4):
    b = torch.rand((B, 256, 256, 3), device="cuda") * 255
    for _ in range(3):
        l = tr.step(b)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        l = tr.step(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print("train b%d: %.2f ms/step  %.1f img/s  loss %.4g finite=%s" % (B, dt * 1e3, B / dt, float(l[0]), bool(torch.isfinite(l).all())), flush=True)
