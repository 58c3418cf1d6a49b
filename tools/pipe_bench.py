#!/usr/bin/env python
"""Throughput of the TFRecord input pipeline alone (decode threads + GPU resize + HBM shuffle queue) on
synthetic COCO-like shards (640x480 JPEG quality 90).  usage: pipe_bench.py [n_images] [threads] [host]
"host": the host half only -- record framing, Example parsing and the JPEG decode pool, decoded images dropped (no engine, no GPU): what a rank's
cores sustain when eight pools run side by side on one box whose single GPU would otherwise be shared by all eight (tools/pipe_bench8.sh)."""
import io
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faststyle_amd import datapipe, engine, tfrecord  # noqa: E402


def main():
    from PIL import Image
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else None
    rng = np.random.default_rng(0)
    d = tempfile.mkdtemp()
    base = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    jpegs = []
    for k in range(16):      # 16 distinct natural-ish images (upsampled noise + gradient), reused
        im = Image.fromarray(np.roll(base, k, axis=1)).resize((640, 480), Image.BICUBIC)
        buf = io.BytesIO()
        im.save(buf, "JPEG", quality=90)
        jpegs.append(buf.getvalue())
    files = []
    for s in range(4):
        p = os.path.join(d, "train-%05d-of-00004" % s)
        with tfrecord.RecordWriter(p) as w:
            for k in range(n // 4):
                w.write(tfrecord.encode_example({"image/encoded": jpegs[k % 16], "image/height": 480, "image/width": 640,
                                                 "image/channels": 3}))
        files.append(p)
    print("shards: %d images, %.1f KB/jpeg" % (n, np.mean([len(j) for j in jpegs]) / 1e3))
    if len(sys.argv) > 3 and sys.argv[3] == "host":
        th = threads or 24
        it = datapipe._prefetch_map(datapipe.decode_jpeg, datapipe._examples(files, 1, np.random.default_rng(0)), th, window=4 * th)
        t0 = time.time()
        k = sum(1 for _ in it)
        dt = time.time() - t0
        print("pipeline: %.0f images/s host half only (%d images decoded in %.2f s, %d decode threads, %d host cores)" % (k / dt, k, dt, th, os.cpu_count()))
        return
    eng = engine.Engine()
    import torch
    it = datapipe.batcher(files, 4, (256, 256), num_epochs=1, min_after_dequeue=256, engine=eng, num_threads=threads)
    next(it)
    torch.cuda.synchronize()
    t0 = time.time()
    nb = 0
    for b in it:
        nb += 1
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("pipeline: %.0f images/s (%d batches of 4 in %.2f s, %s decode threads, %d host cores)" %
          (nb * 4 / dt, nb, dt, threads or "auto", os.cpu_count()))


if __name__ == "__main__":
    main()
