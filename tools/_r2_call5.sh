#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2e
rm -rf $O && mkdir -p $O
( time timeout 200 python - <<'PY'
import time, sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import perceptual, tnet, torch_ref
P = tnet.init_params(0); Wv = perceptual.synthetic_vgg_weights(3)
style = np.random.default_rng(2).uniform(0, 255, (1, 128, 128, 3)).astype(np.float32)
tg = perceptual.target_grams(style, Wv, ("conv1_2", "conv2_2", "conv3_3", "conv4_3"))
Pt = dict((k, torch.tensor(v, requires_grad=True)) for k, v in P.items())
Wt = dict((k, torch.tensor(v)) for k, v in Wv.items())
tgt = [torch.tensor(g) for g in tg]
x4 = torch.rand(4, 256, 256, 3) * 255
x720 = torch.rand(1, 720, 1280, 3) * 255
for th in (32, 64, 16, 128):
    torch.set_num_threads(th)
    t0 = time.perf_counter(); torch_ref.train_step(Pt, x4, tgt, Wt); t1 = time.perf_counter(); torch_ref.train_step(Pt, x4, tgt, Wt); t2 = time.perf_counter()
    with torch.no_grad():
        Pn = dict((k, v.detach()) for k, v in Pt.items())
        torch_ref.tnet(x720, Pn); t3 = time.perf_counter(); torch_ref.tnet(x720, Pn); t4 = time.perf_counter()
    print("threads %d: b4 step warm %.2f s, timed %.2f s; 720p fwd %.2f s" % (th, t1 - t0, t2 - t1, t4 - t3), flush=True)
PY
) > $O/scan.log 2>&1
cat $O/scan.log
