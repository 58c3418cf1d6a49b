#!/usr/bin/env python
"""Randomised shapes through the round-2 kernels on the CPU kernel emulator (no GPU needed), against numpy float64:

    python tools/fuzz_emu.py [cases] [seed]

  * conv_stream_kernel (fs_cstream.hip), all five instances: stride-2 3x3 and 2x2-tap convs with the pixel-shuffle store,
    producer instance norm + ReLU on load, per-tile statistics (merged and compared with the tensor's own mean / variance),
    the residual-gradient addend; ragged tiles, one to several tiles per workgroup (FS_CSTREAM_WGS);
  * gram_stream_kernel / gram_reduce_kernel / gram_bwd_kernel (fs_gram.hip): every channel count they take, pixel counts
    that end inside a tile, several pixel ranges;
  * wino2_conv_kernel with bias / ReLU, wgrad2_kernel, conv3x3_to3_kernel on ragged shapes;
  * wino4_conv_kernel (fs_wino4.hip, Winograd F(4x4,3x3)): raw / bias + ReLU / consumer-mask epilogues, ragged 16x32 blocks;
  * wino4t_conv_kernel (fs_wino4t*.hip): 16- / 32-tile items, paddings 0 / 1 / 2, every epilogue form through fs_conv2d_fwd (round 5: incl. the two
    forms that also leave instance-norm-backward partial sums); its 128-channel
    item form through the whole VGG16 section (fs_perceptual_loss against the oracle's loss and gradient);
  * conv_s16_kernel (fs_s16.hip): the 9x9 3 -> 16 layer with mirror or zero padding and per-tile statistics, VGG conv1_1's
    form (3 -> 64, per-channel affine on load over zero padding, bias + ReLU); ragged tiles, several persistent grid sizes.
Every case prints one line; a mismatch raises.  tests/ holds fixed-shape versions of the same checks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nnops, perceptual  # noqa: E402
from tests.backends import get_engine  # noqa: E402

TOL = 3e-5


def rel(got, want):
    return float(np.abs(np.asarray(got, np.float64) - want).max() / (np.abs(want).max() + 1e-30))


def conv_explicit(x, w, stride, pad_t, pad_l, Ho, Wo):
    """Cross-correlation with explicit top/left zero padding and a given output extent (bottom/right padding as needed)."""
    N, H, W, _ = x.shape
    kh, kw = w.shape[:2]
    need_h, need_w = (Ho - 1) * stride + kh, (Wo - 1) * stride + kw
    xp = np.pad(x, ((0, 0), (pad_t, max(0, need_h - H - pad_t)), (pad_l, max(0, need_w - W - pad_l)), (0, 0)))
    y = np.zeros((N, Ho, Wo, w.shape[3]))
    for i in range(kh):
        for j in range(kw):
            y += np.tensordot(xp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :], w[i, j], axes=1)
    return y


def shuffle2(y):
    """[N,H,W,4*C] (phase q = 2a+b major) -> [N,2H,2W,C]."""
    N, H, W, C4 = y.shape
    C = C4 // 4
    out = np.zeros((N, 2 * H, 2 * W, C))
    for q in range(4):
        out[:, (q >> 1)::2, (q & 1)::2, :] = y[..., q * C:(q + 1) * C]
    return out


def merge_stats(st):
    """[N,T,C,3] {mean, M2, count} per tile -> per (n, c) mean and biased variance."""
    cnt = st[..., 2].sum(axis=1)
    mean = (st[..., 0] * st[..., 2]).sum(axis=1) / cnt
    m2 = (st[..., 1] + st[..., 2] * (st[..., 0] - mean[:, None, :]) ** 2).sum(axis=1)
    return mean, m2 / cnt


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    os.environ["FS_CSTREAM_MIN_TILES"] = "1"
    os.environ["FS_CSTREAM_MASK"] = "31"
    os.environ["FS_GRAM2_MIN_TILES"] = "0"
    os.environ["FS_S16_MIN_TILES"] = "1"
    e = get_engine("emu")
    up, down = e.mem.from_numpy, e.mem.to_numpy
    inst = [(16, 32, 3, 2), (32, 64, 2, 1), (32, 64, 3, 2), (64, 128, 2, 1), (64, 64, 3, 1)]
    for it in range(cases):
        kind = int(os.environ["FUZZ_KIND"]) if "FUZZ_KIND" in os.environ else it % 15   # (FUZZ_KIND=<n>: that kernel family only)
        if kind == 11:        # 16-channel-block streaming kernel (fs_s16.hip)
            n, h, w = int(rng.integers(1, 3)), int(rng.integers(9, 50)), int(rng.integers(9, 50))
            os.environ["FS_S16_WGS"] = str(int(rng.choice([1, 3, 512])))
            e.lib.fs_debug_reload_env()
            x = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
            if rng.integers(2):   # the image layer: 9x9, 3 -> 16, mirror padding by r < min(h, w) or zero padding, statistics
                refl = int(rng.integers(0, min(h, w, 41)))
                wt = (rng.standard_normal((9, 9, 3, 16)) * 0.05).astype(np.float32)
                xv = nnops.reflect_pad(x.astype(np.float64), refl) if refl else x.astype(np.float64)
                want = nnops.conv2d(xv, wt.astype(np.float64), 1, "SAME")
                res = e.conv2d(up(x), up(wt), 1, "SAME", want_stats=True, src_mode=1 if refl else 0, refl=refl)
                y = down(res[0])
                mean, var = merge_stats(down(res[1]).astype(np.float64))
                assert np.abs(mean - want.mean(axis=(1, 2))).max() < 1e-4 * (np.abs(want).max() + 1), "tile statistics: mean"
                assert np.abs(var - want.var(axis=(1, 2))).max() < 1e-4 * (want.var(axis=(1, 2)).max() + 1e-9), "tile statistics: variance"
                what = "9x9 3->16 refl %d" % refl
            else:                 # VGG conv1_1: 3x3, 3 -> 64, image - mean on load (zero padding stays zero), bias + ReLU
                wt = (rng.standard_normal((3, 3, 3, 64)) * 0.1).astype(np.float32)
                b = rng.standard_normal(64).astype(np.float32)
                ia, ib = rng.uniform(0.5, 1.5, 3).astype(np.float32), -rng.uniform(90, 130, 3).astype(np.float32)
                want = np.maximum(nnops.conv2d(x.astype(np.float64) * ia + ib, wt.astype(np.float64), 1, "SAME") + b, 0.0)
                y = down(e.conv2d(up(x), up(wt), 1, "SAME", in_a=up(ia), in_b=up(ib), bias=up(b), out_relu=1))
                what = "3x3 3->64 affine+bias+relu"
            r = rel(y, want)
            print("case %3d conv_s16 %s %s wgs %s  rel %.2e" % (it, what, x.shape, os.environ["FS_S16_WGS"], r), flush=True)
            assert y.shape == want.shape and r < TOL
        elif kind < 5:
            cin, cout, ks, st = inst[kind]
            n = int(rng.integers(1, 4))
            h, w = int(rng.integers(3, 40)), int(rng.integers(3, 44))
            os.environ["FS_CSTREAM_WGS"] = str(int(rng.choice([1, 3, 256])))
            e.lib.fs_debug_reload_env()
            x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
            wt = (rng.standard_normal((ks, ks, cin, cout)) * 0.1).astype(np.float32)
            kw = {}
            xin = x.astype(np.float64)
            if kind in (0, 2) or (kind == 4 and rng.integers(2)):      # producer instance norm + ReLU on load
                a = rng.uniform(0.5, 1.5, (n, cin)).astype(np.float32)
                b = rng.standard_normal((n, cin)).astype(np.float32)
                kw.update(in_a=up(a), in_b=up(b), in_per_sample=1, in_relu=1)
                xin = np.maximum(xin * a[:, None, None, :] + b[:, None, None, :], 0.0)
            if ks == 2:                                               # phase-collapsed form: explicit padding, shuffle store
                pt, pl = int(rng.integers(0, 2)), int(rng.integers(0, 2))
                want = shuffle2(conv_explicit(xin, wt.astype(np.float64), 1, pt, pl, h, w))
                y = down(e.conv2d(up(x), up(wt), 1, (pt, pl, h, w), shuffle=1, **kw))
            elif kind == 4:                                           # 64 -> 64: VALID forward or "full" input gradient (+ addend)
                if h < 3 or w < 3:
                    continue
                full = bool(rng.integers(2))
                if full:
                    ho, wo = h + 2, w + 2
                    want = conv_explicit(xin, wt.astype(np.float64), 1, 2, 2, ho, wo)
                    if rng.integers(2) and ho > 4 and wo > 4:
                        add = rng.standard_normal((n, ho - 4, wo - 4, cout)).astype(np.float32)
                        want[:, 2:-2, 2:-2, :] += add
                        kw.update(add_src=up(add), add_pad=2)
                    y = down(e.conv2d(up(x), up(wt), 1, (2, 2, ho, wo), **kw))
                else:
                    want = nnops.conv2d(xin, wt.astype(np.float64), 1, "VALID")
                    y = down(e.conv2d(up(x), up(wt), 1, "VALID", **kw))
            else:
                want = nnops.conv2d(xin, wt.astype(np.float64), st, "SAME")
                res = e.conv2d(up(x), up(wt), st, "SAME", want_stats=True, **kw)
                y = down(res[0])
                mean, var = merge_stats(down(res[1]).astype(np.float64))
                assert np.abs(mean - want.mean(axis=(1, 2))).max() < 1e-4 * (np.abs(want).max() + 1), "tile statistics: mean"
                assert np.abs(var - want.var(axis=(1, 2))).max() < 1e-4 * (want.var(axis=(1, 2)).max() + 1e-9), "tile statistics: variance"
            assert y.shape == want.shape, (y.shape, want.shape)
            r = rel(y, want)
            print("case %3d conv_stream %d->%d k%d s%d  %s  wgs %s  rel %.2e" % (it, cin, cout, ks, st, x.shape, os.environ["FS_CSTREAM_WGS"], r), flush=True)
            assert r < TOL
        elif kind in (5, 6):
            c = int(rng.choice([64, 128, 256, 384, 512]))
            n = int(rng.integers(1, 4))
            h, w = int(rng.integers(1, 30)), int(rng.integers(1, 30))
            os.environ["FS_GRAM2_ITEMS"] = str(int(rng.choice([8, 64, 768])))
            e.lib.fs_debug_reload_env()
            f = rng.standard_normal((n, h, w, c)).astype(np.float32)
            g = down(e.gram(up(f)))
            want = perceptual.gram(f.astype(np.float64))
            r = rel(g, want)
            print("case %3d gram_stream %s items %s  rel %.2e" % (it, f.shape, os.environ["FS_GRAM2_ITEMS"], r), flush=True)
            assert r < TOL and np.array_equal(g, g.transpose(0, 2, 1))
        elif kind == 12:      # Winograd F(4x4,3x3) (fs_wino4.hip): SAME 3x3, the three epilogue forms, ragged 16x32 blocks, several items per workgroup
            cin, cout = int(rng.choice([4, 8, 20, 64])), int(rng.choice([64, 128]))
            n, h, w = int(rng.integers(1, 3)), int(rng.integers(1, 44)), int(rng.integers(1, 70))
            os.environ["FS_WINO4_WGS"] = str(int(rng.choice([1, 3, 256])))
            e.lib.fs_debug_reload_env()
            x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
            wt = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
            epi = int(rng.integers(3))
            kw, want = {}, nnops.conv2d(x.astype(np.float64), wt.astype(np.float64), 1, "SAME")
            if epi == 1:
                bias = rng.standard_normal((cout,)).astype(np.float32)
                kw = dict(bias=up(bias), out_relu=1)
                want = np.maximum(want + bias, 0.0)
            elif epi == 2:
                mask = rng.standard_normal((n, h, w, cout)).astype(np.float32)
                kw = dict(mask_src=up(mask))
                want = np.where(mask > 0, want, 0.0)
            y = down(e.conv2d(up(x), up(wt), 1, "SAME", winograd=4, **kw))
            r = rel(y, want)
            print("case %3d wino4 %s -> %d epilogue %d wgs %s  rel %.2e" % (it, x.shape, cout, epi, os.environ["FS_WINO4_WGS"], r), flush=True)
            assert r < 5e-5
        elif kind == 13:      # register-fed Winograd F(4x4,3x3) (fs_wino4t.hip): 16- / 32-tile items, padding 0 / 1 / 2, every epilogue form, ragged blocks
            cin, cout = int(rng.choice([8, 16, 24, 64])), int(rng.choice([64, 128]))
            pad = int(rng.integers(3))
            n, h, w = int(rng.integers(1, 3)), int(rng.integers(3 - pad, 40)), int(rng.integers(3 - pad, 70))
            os.environ["FS_WINO4T_WGS"] = str(int(rng.choice([1, 3, 256])))
            os.environ["FS_WINO4T_TB"] = str(int(rng.choice([1, 2])))
            os.environ["FS_WINO4T_FLAT"] = str(int(rng.choice([0, 2])))    # (2: the flattened 16-tile form wherever its epilogue forms allow)
            e.lib.fs_debug_reload_env()
            x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
            wt = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
            ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
            want = nnops.conv2d(np.pad(x.astype(np.float64), ((0, 0), (pad, pad), (pad, pad), (0, 0))), wt.astype(np.float64), 1, "VALID")
            epi = int(rng.integers(7))
            kw, stats = {}, False
            inb = None
            if epi == 1:
                bias = rng.standard_normal((cout,)).astype(np.float32)
                kw = dict(bias=up(bias), out_relu=1)
                want = np.maximum(want + bias, 0.0)
            elif epi == 2:
                mask = rng.standard_normal((n, ho, wo, cout)).astype(np.float32)
                kw = dict(mask_src=up(mask))
                want = np.where(mask > 0, want, 0.0)
            elif epi == 3 and ho > 4 and wo > 4:
                add = rng.standard_normal((n, ho - 4, wo - 4, cout)).astype(np.float32)
                kw = dict(add_src=up(add), add_pad=2)
                want[:, 2:-2, 2:-2, :] += add
            elif epi == 4:
                stats = True
            elif epi >= 5 and cout == 64:   # round 5: raw (5) / residual-gradient (6) epilogue + the instance-norm-backward partial sums of the unit below
                if epi == 6 and ho > 4 and wo > 4:
                    add = rng.standard_normal((n, ho - 4, wo - 4, cout)).astype(np.float32)
                    kw = dict(add_src=up(add), add_pad=2)
                    want[:, 2:-2, 2:-2, :] += add
                zz = (rng.standard_normal((n, ho, wo, cout)) * 2).astype(np.float32)
                mean, rstd = zz.mean(axis=(1, 2)).astype(np.float32), (rng.uniform(0.3, 2.0, (n, cout))).astype(np.float32)
                ia, ib = rng.standard_normal((n, cout)).astype(np.float32), rng.standard_normal((n, cout)).astype(np.float32)
                relu = bool(rng.integers(2))
                inb = (zz, mean, rstd, ia, ib, relu)
                kw["inb"] = (up(zz), up(mean), up(rstd), up(ia), up(ib), relu)
                os.environ["FS_WINO4T_TB"] = "1"    # (the planner takes 16-tile items for these launches whatever the knob says: keep the printout honest)
            out = e.conv2d(up(x), up(wt), 1, (pad, pad, ho, wo), winograd="4t", want_stats=stats, **kw)
            y = down(out[0] if (stats or inb is not None) else out)
            r = rel(y, want)
            if inb is not None:
                zz, mean, rstd, ia, ib, relu = inb
                rec = down(out[1]).astype(np.float64)
                keep = (zz.astype(np.float64) * ia[:, None, None, :] + ib[:, None, None, :] > 0) if relu else np.ones(zz.shape, bool)
                gq = np.where(keep, y.astype(np.float64), 0.0)
                xh = (zz.astype(np.float64) - mean[:, None, None, :]) * rstd[:, None, None, :]
                r = max(r, rel(rec[..., 0].sum(axis=1), gq.sum(axis=(1, 2))) * (np.abs(gq.sum(axis=(1, 2))).max() / max(np.abs(gq).sum(axis=(1, 2)).max(), 1e-30)),
                        rel(rec[..., 1].sum(axis=1), (gq * xh).sum(axis=(1, 2))) * (np.abs((gq * xh).sum(axis=(1, 2))).max() / max(np.abs(gq * xh).sum(axis=(1, 2)).max(), 1e-30)))
            if stats:
                mean, var = merge_stats(down(out[1]))
                r = max(r, rel(mean, want.mean(axis=(1, 2))), rel(var, want.var(axis=(1, 2))))
            print("case %3d wino4t %s -> %d pad %d epilogue %d tb %s flat %s wgs %s  rel %.2e" % (it, x.shape, cout, pad, epi, os.environ["FS_WINO4T_TB"],
                                                                                               os.environ["FS_WINO4T_FLAT"], os.environ["FS_WINO4T_WGS"], r), flush=True)
            assert r < 5e-5
        elif kind == 14:      # the VGG16 section through the big-item forms of fs_wino4t.hip (32 tiles x 64 channels, 16 tiles x 128 channels)
            from faststyle_amd import engine as fs_engine
            os.environ["FS_WINO4T_TB"] = "2"
            os.environ["FS_WINO4T_WGS"] = str(int(rng.choice([3, 256])))
            e.lib.fs_debug_reload_env()
            e.reset_workspaces()
            Wv = perceptual.synthetic_vgg_weights(seed=3)
            e.vgg_load(Wv)
            cfg = fs_engine.default_loss_cfg()
            n, h, w = 1, int(rng.integers(16, 34)), int(rng.integers(16, 42))
            style = rng.uniform(0, 255, (1, 24, 32, 3)).astype(np.float32)
            tg = e.style_targets(up(style), cfg)
            y = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
            xc = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
            losses, dy = e.perceptual_loss(up(y), up(xc), tg, cfg)
            W64 = dict((k, v.astype(np.float64)) for k, v in Wv.items())
            tgo = perceptual.target_grams(style.astype(np.float64), W64, cfg["style_layers"])
            feats = perceptual.vgg16(xc.astype(np.float64), W64, upto="conv3_3")
            lo, dyo = perceptual.perceptual_loss(y.astype(np.float64), [feats["conv3_3"]], tgo, W64, beta=0.0)
            r = abs(float(down(losses)[0]) - lo["loss"]) / abs(lo["loss"])
            g, go = down(dy).ravel().astype(np.float64), dyo.ravel()
            cos = float(np.dot(g, go) / (np.linalg.norm(g) * np.linalg.norm(go)))
            os.environ.pop("FS_WINO4T_TB")
            e.lib.fs_debug_reload_env()
            e.reset_workspaces()
            print("case %3d vgg section, big items %s wgs %s  loss rel %.2e  gradient cos %.7f" % (it, y.shape, os.environ["FS_WINO4T_WGS"], r, cos), flush=True)
            assert r < 1e-4 and cos > 0.9999
        elif kind == 8:       # second-generation Winograd kernel (fs_wino2.hip): SAME 3x3, bias + ReLU epilogue, ragged 16x16 blocks
            cin, cout = int(rng.choice([8, 16, 64, 128])), int(rng.choice([64, 128]))
            n, h, w = int(rng.integers(1, 3)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
            x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
            wt = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
            bias = rng.standard_normal((cout,)).astype(np.float32)
            relu = bool(rng.integers(2))
            y = down(e.conv2d(up(x), up(wt), 1, "SAME", winograd=True, bias=up(bias), out_relu=int(relu)))
            want = nnops.conv2d(x.astype(np.float64), wt.astype(np.float64), 1, "SAME") + bias
            if relu:
                want = np.maximum(want, 0.0)
            r = rel(y, want)
            print("case %3d wino2 %s -> %d relu %s  rel %.2e" % (it, x.shape, cout, relu, r), flush=True)
            assert r < TOL
        elif kind == 9:       # second-generation filter gradient (fs_wgrad2.hip)
            cin, cout = int(rng.choice([16, 32, 64])), int(rng.choice([16, 32, 64]))
            k, st = (3, int(rng.choice([1, 2]))) if rng.integers(2) else (2, 1)
            n, h, w = int(rng.integers(1, 4)), int(rng.integers(4, 36)), int(rng.integers(4, 36))
            x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
            pad = "SAME" if k == 3 else "VALID"
            ho, wo = (-(-h // st), -(-w // st)) if pad == "SAME" else (h - k + 1, w - k + 1)
            dy = rng.standard_normal((n, ho, wo, cout)).astype(np.float32)
            dw = down(e.conv2d_wgrad(up(x), up(dy), k, st, pad))
            want = nnops.conv2d_bwd_filter(x.astype(np.float64), dy.astype(np.float64), k, st, pad)
            r = rel(dw, want)
            print("case %3d wgrad2 %s k%d s%d -> %d  rel %.2e" % (it, x.shape, k, st, cout, r), flush=True)
            assert r < TOL
        elif kind == 10:      # 64 -> 3 channels on the vector ALU (fs_c3.hip)
            n, h, w = int(rng.integers(1, 3)), int(rng.integers(1, 45)), int(rng.integers(1, 45))
            x = rng.standard_normal((n, h, w, 64)).astype(np.float32)
            wt = (rng.standard_normal((3, 3, 64, 3)) * 0.1).astype(np.float32)
            y = down(e.conv2d(up(x), up(wt), 1, "SAME"))
            r = rel(y, nnops.conv2d(x.astype(np.float64), wt.astype(np.float64), 1, "SAME"))
            print("case %3d conv3x3_to3 %s  rel %.2e" % (it, x.shape, r), flush=True)
            assert r < TOL
        else:
            c = int(rng.choice([64, 128, 256]))
            n = int(rng.integers(1, 4))
            h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
            x = rng.standard_normal((n, h, w, c)).astype(np.float32)
            s = (rng.standard_normal((n, c, c)) * 0.1).astype(np.float32)
            kw = {"w_nstride": c * c}
            want = np.einsum("nhwc,ncd->nhwd", x.astype(np.float64), s.astype(np.float64))
            if rng.integers(2):
                add = rng.standard_normal(x.shape).astype(np.float32)
                kw["add_src"] = up(add)
                want = want + add
            y = down(e.conv2d(up(x), up(s.reshape(n, 1, 1, c, c)), 1, "SAME", **kw))
            r = rel(y, want)
            print("case %3d gram_bwd %s add %s  rel %.2e" % (it, x.shape, "add_src" in kw, r), flush=True)
            assert r < TOL
    print("fuzz_emu: %d cases ok" % cases)


if __name__ == "__main__":
    main()
