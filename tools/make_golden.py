#!/usr/bin/env python
"""Generates the committed oracle fixtures under tests/golden/ ("golden vectors to commit" (2)/(3)).

    python tools/make_golden.py            # rewrites tests/golden/oracle_forward.npz and tests/golden/oracle_train.npz

The values come from the float64 numpy oracle (oracle/), which is itself pinned to the reference's shipped known answers
(tests/test_oracle_golden.py).  They freeze the oracle: tests/test_golden_fixtures.py holds BOTH the live oracle and the
HIP path to these files, so an oracle regression and a kernel regression in the same direction cannot hide each other.
Inputs are reproducible from seeds (numpy default_rng / the oracle's synthetic_vgg_weights), only outputs are stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faststyle_amd import ckpt  # noqa: E402  (the bundle reader: no device needed)
from oracle import perceptual, tnet  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
STYLE_LAYERS = ("conv1_2", "conv2_2", "conv3_3", "conv4_3")


def f64(d):
    return dict((k, np.asarray(v, np.float64)) for k, v in d.items())


def forward_inputs():
    x = np.random.default_rng(20).uniform(0, 255, (1, 48, 56, 3)).astype(np.float32)
    return x


def train_inputs():
    rng = np.random.default_rng(21)
    x = rng.uniform(0, 255, (1, 64, 64, 3)).astype(np.float32)
    style = rng.uniform(0, 255, (1, 40, 52, 3)).astype(np.float32)
    return x, style


def summary(a):
    """What is stored of a large tensor: its first 16 elements, sum and sum of squares (float64)."""
    a = np.asarray(a, np.float64).ravel()
    return np.concatenate([a[:16], [a.sum(), np.square(a).sum()]])


def make_forward():
    W = f64(tnet.strip_scope(ckpt.load_checkpoint(os.path.join(ROOT, "models", "starry_final.ckpt"))))
    x = forward_inputs().astype(np.float64)
    y, cache = tnet.create_net(x, W, keep=True)
    out = {"y": y.astype(np.float32)}
    for name, act in cache["acts"].items():          # post-activation output of every unit / residual block
        out["act/" + name] = summary(act)
    # 256x256 resized chicago (PIL BICUBIC of the full image): output checksum + a 64x64 crop
    from PIL import Image
    im = Image.open(os.path.join(GOLD, "ref_assets", "chicago.jpg")).convert("RGB").resize((256, 256), Image.BICUBIC)
    xc = np.asarray(im, np.float64)[None]
    yc = tnet.create_net(xc, W)
    out["chicago256/summary"] = summary(yc)
    out["chicago256/crop"] = yc[0, 96:160, 96:160].astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "oracle_forward.npz"), **out)
    return out


def make_train():
    x, style = train_inputs()
    P = f64(tnet.init_params(0))
    Wv = f64(perceptual.synthetic_vgg_weights(3))
    tg = perceptual.target_grams(style.astype(np.float64), Wv, STYLE_LAYERS)
    losses, grads, y = perceptual.train_step(P, x.astype(np.float64), tg, Wv, beta=1e-4)
    feats = perceptual.vgg16(y, Wv)
    out = {"losses": np.array([losses["loss"], losses["content_loss"], losses["style_loss"], losses["tv_loss"]], np.float64)}
    for n in STYLE_LAYERS:
        g = perceptual.gram(feats[n])[0]
        out["gram/" + n] = g.astype(np.float32) if g.shape[0] <= 128 else summary(g)
        out["target_gram/" + n] = summary(tg[STYLE_LAYERS.index(n)])
    flat = np.concatenate([grads[k].ravel() for k in sorted(grads)])
    out["grad/l2"] = np.array([np.linalg.norm(flat)])
    for k in sorted(grads):
        out["grad/" + k] = summary(grads[k])
    m = dict((k, np.zeros_like(v)) for k, v in P.items())
    v = dict((k, np.zeros_like(v)) for k, v in P.items())
    Pc = dict((k, p.copy()) for k, p in P.items())
    for t in (1, 2):                                  # two TF1-Adam steps on the same batch
        _, g, _ = perceptual.train_step(Pc, x.astype(np.float64), tg, Wv, beta=1e-4)
        perceptual.adam_tf(Pc, g, m, v, t)
        for k in sorted(Pc):
            out["adam%d/%s" % (t, k)] = summary(Pc[k])
    np.savez_compressed(os.path.join(GOLD, "oracle_train.npz"), **out)
    return out


if __name__ == "__main__":
    a = make_forward()
    b = make_train()
    print("wrote %d + %d arrays under %s" % (len(a), len(b), GOLD))
