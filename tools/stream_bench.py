#!/usr/bin/env python
"""Latency / throughput of the frame-streaming path (u8 in -> stylize -> u8 out), 720p and 1080p, batch 1.
Reports end-to-end (PCIe both ways, host sync per frame) and device-only (graph replay) rates."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import ckpt, engine, stream  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    eng = engine.Engine()
    variables = eng.mem.from_numpy(eng.flatten_params(ckpt.load_checkpoint(os.path.join(ROOT, "models", "starry_final.ckpt"))))
    rng = np.random.default_rng(0)
    for (h, w) in ((720, 1280), (1080, 1920)):
        st = stream.FrameStylizer(eng, variables, h, w)
        frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for _ in range(5):
            st(frame)
        lat = []
        for _ in range(50):
            t0 = time.perf_counter()
            st(frame)
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e3
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(50):
            st._graph.replay()
        t1.record()
        torch.cuda.synchronize()
        dev = t0.elapsed_time(t1) / 50
        print("%dx%d: end-to-end median %.2f ms (%.0f fps, p95 %.2f ms); device-only %.2f ms (%.0f fps)" %
              (w, h, np.median(lat), 1e3 / np.median(lat), np.percentile(lat, 95), dev, 1e3 / dev), flush=True)


if __name__ == "__main__":
    main()
