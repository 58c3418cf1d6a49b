"""Committed oracle fixtures (tests/golden/oracle_*.npz, written by tools/make_golden.py from the float64 oracle;
SURVEY.md section 8c "golden vectors to commit" (2)/(3)).

Two directions:
  * the LIVE oracle must still reproduce the files (an oracle regression shows up here, on CPU);
  * the HIP path (kernel emulator on CPU, product library on the GPU) is held to the FILES, not to the live oracle -- so
    a change that moves the oracle and the kernels the same way cannot pass unnoticed.
Hardening of the training path beyond that: the 256x256 batch-4 step with the real 640x938 style
image over 8 seeds, Winograd on and off, and batches with values outside [0,255] (TF1 bicubic overshoot, train.py:158-160).
"""
import os

import numpy as np
import pytest

from faststyle_amd import ckpt, engine
from oracle import perceptual, tnet
from tests.backends import engine_params, get_engine
from tools import make_golden as mg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(params=engine_params())
def eng(request):
    return get_engine(request.param)


@pytest.fixture(scope="module")
def fwd():
    return np.load(os.path.join(GOLD, "oracle_forward.npz"))


@pytest.fixture(scope="module")
def trn():
    return np.load(os.path.join(GOLD, "oracle_train.npz"))


def starry():
    return tnet.strip_scope(ckpt.load_checkpoint(os.path.join(ROOT, "models", "starry_final.ckpt")))


# ------------------------------------------------------------------ the live oracle against the files
def test_live_oracle_reproduces_forward_fixture(fwd):
    W = mg.f64(starry())
    y, cache = tnet.create_net(mg.forward_inputs().astype(np.float64), W, keep=True)
    np.testing.assert_allclose(y, fwd["y"], atol=1e-4)                       # (stored as float32)
    for name, act in cache["acts"].items():
        np.testing.assert_allclose(mg.summary(act), fwd["act/" + name], rtol=1e-10, atol=1e-10)


def test_live_oracle_reproduces_train_fixture(trn):
    x, style = mg.train_inputs()
    P = mg.f64(tnet.init_params(0))
    Wv = mg.f64(perceptual.synthetic_vgg_weights(3))
    tg = perceptual.target_grams(style.astype(np.float64), Wv, mg.STYLE_LAYERS)
    losses, grads, _ = perceptual.train_step(P, x.astype(np.float64), tg, Wv, beta=1e-4)
    np.testing.assert_allclose([losses["loss"], losses["content_loss"], losses["style_loss"], losses["tv_loss"]], trn["losses"], rtol=1e-12)
    for k in sorted(grads):
        np.testing.assert_allclose(mg.summary(grads[k]), trn["grad/" + k], rtol=1e-9, atol=1e-12)
    for i, n in enumerate(mg.STYLE_LAYERS):
        np.testing.assert_allclose(mg.summary(tg[i]), trn["target_gram/" + n], rtol=1e-10)


# ------------------------------------------------------------------ the HIP path against the files
def test_hip_forward_matches_committed_fixture(eng, fwd):
    W = starry()
    flat = eng.mem.from_numpy(eng.flatten_params(W, scope=""))
    y = eng.mem.to_numpy(eng.tnet_forward(flat, eng.mem.from_numpy(mg.forward_inputs())))
    assert y.shape == fwd["y"].shape
    assert np.abs(y - fwd["y"]).max() / 255.0 < 2e-5                          # north-star budget: 1e-3


@pytest.mark.gpu
def test_hip_chicago_256_matches_committed_fixture(fwd):
    """BASELINE configs[0] input: chicago.jpg resized to 256x256 (PIL BICUBIC), starry weights."""
    from PIL import Image
    e = get_engine("hip")
    im = Image.open(os.path.join(GOLD, "ref_assets", "chicago.jpg")).convert("RGB").resize((256, 256), Image.BICUBIC)
    x = np.asarray(im, np.float32)[None]
    flat = e.mem.from_numpy(e.flatten_params(starry(), scope=""))
    y = e.mem.to_numpy(e.tnet_forward(flat, e.mem.from_numpy(x)))
    assert np.abs(y[0, 96:160, 96:160] - fwd["chicago256/crop"]).max() / 255.0 < 2e-5
    s = mg.summary(y)
    assert abs(s[-2] - fwd["chicago256/summary"][-2]) / fwd["chicago256/summary"][-2] < 1e-6     # sum of all pixels
    assert abs(s[-1] - fwd["chicago256/summary"][-1]) / fwd["chicago256/summary"][-1] < 1e-6     # sum of squares


def test_hip_train_step_matches_committed_fixture(eng, trn):
    """64x64 train step (fixture (3)): Grams, the loss scalars incl. a TV term, all 48 gradients, two TF-Adam steps."""
    x, style = mg.train_inputs()
    Wv = perceptual.synthetic_vgg_weights(3)
    eng.vgg_load(Wv)
    cfg = dict(engine.default_loss_cfg(), beta=1e-4)
    P = tnet.init_params(0)
    flat = eng.mem.from_numpy(eng.flatten_params(P, scope=""))
    tg = eng.style_targets(eng.mem.from_numpy(style), cfg)
    for i, n in enumerate(mg.STYLE_LAYERS):
        np.testing.assert_allclose(mg.summary(eng.mem.to_numpy(tg[i]))[-2:], trn["target_gram/" + n][-2:], rtol=2e-5)
    xd = eng.mem.from_numpy(x)
    m, v = eng.mem.zeros(flat.shape), eng.mem.zeros(flat.shape)
    for t in (1, 2):
        y = eng.tnet_forward(flat, xd, save_for_bwd=True)
        losses, dy = eng.perceptual_loss(y, xd, tg, cfg)
        g = eng.tnet_backward(flat, xd, dy)
        if t == 1:
            np.testing.assert_allclose(eng.mem.to_numpy(losses), trn["losses"], rtol=2e-4)
            feats = eng.vgg_features(y, list(mg.STYLE_LAYERS))
            for n, f in zip(mg.STYLE_LAYERS, feats):
                G = eng.mem.to_numpy(eng.gram(f))[0]
                want = trn["gram/" + n]
                if want.ndim == 2:
                    assert np.abs(G - want).max() / np.abs(want).max() < 2e-5
                else:
                    np.testing.assert_allclose(mg.summary(G)[-2:], want[-2:], rtol=5e-5)
            gh = eng.mem.to_numpy(g)
            assert abs(np.linalg.norm(gh.astype(np.float64)) - trn["grad/l2"][0]) / trn["grad/l2"][0] < 2e-3
            gmax = max(np.abs(trn["grad/" + name][:-2]).max() for name, _, _ in eng.param_table())
            bad = []
            for name, off, shape in eng.param_table():
                want = trn["grad/" + name]
                n_el = int(np.prod(shape))
                got16 = gh[off:off + min(16, n_el)]
                # first elements against the tensor's own scale (floor: tensors whose gradient is 0 by symmetry)
                scale = max(np.sqrt(want[-1] / n_el), 1e-3 * gmax)      # (summary = first <= 16 elements, sum, sum of squares)
                if np.abs(got16 - want[:len(got16)]).max() > 5e-3 * scale + 2e-4 * np.abs(want[:-2]).max():
                    bad.append(name)
            assert not bad, bad
        eng.adam_tf_step(flat, g, m, v, t)
        ph = eng.mem.to_numpy(flat)
        close = total = 0
        for name, off, shape in eng.param_table():
            want = trn["adam%d/%s" % (t, name)][:-2]
            got = ph[off:off + len(want)]
            d = np.abs(got - want)
            assert d.max() < 2.5e-3 * t          # (|update| <= lr per step; a gradient within noise of 0 may flip its sign)
            close += int((d < (2e-5 if t == 1 else 1e-4)).sum())
            total += len(got)
        # (float32-vs-float64 noise in a gradient is amplified by Adam's normalisation where |g| is small; the update arithmetic
        # itself is held to 2e-6 in tests/test_path_parity.py.  Step 2 is chaotic on top of that: the first TF-Adam update is lr * sign(g), so every
        # near-zero gradient element ANYWHERE in the 424 k parameters whose sign differs from the float64 oracle's moves its parameter by 2 lr and
        # perturbs the whole second forward / backward -- round 6: with the 9 x 9 image layer as the split-bf16 kernel (forward error against the
        # oracle 0.9e-3 instead of 1.2e-3 of 255, the layer alone 1.8e-7 instead of 6.6e-7) another set of signs flips and 642 / 716 / 738 of the 742
        # sampled elements are within 5e-5 / 1e-4 / 2e-4 where the fp32 kernel's realisation has 735 / 741 / 741: the step-2 bound is a tenth of lr)
        assert close >= (0.97 if t == 1 else 0.90) * total, (t, close, total)


def test_tnet_and_train_step_with_inputs_outside_0_255(eng):
    """train.py:158-160 / datapipe.py:25: batches are float RGB 'roughly' 0..255 -- TF1's bicubic resize overshoots and
    nothing clips it.  Forward and gradients on such a batch against the float64 oracle."""
    rng = np.random.default_rng(33)
    x = rng.uniform(-40.0, 300.0, (1, 48, 52, 3)).astype(np.float32)
    x[0, :4, :4] = -25.5
    x[0, -4:, -4:] = 280.25
    W = starry()
    flat = eng.mem.from_numpy(eng.flatten_params(W, scope=""))
    xd = eng.mem.from_numpy(x)
    y = eng.tnet_forward(flat, xd, save_for_bwd=True)
    yo, cache = tnet.create_net(x.astype(np.float64), mg.f64(W), keep=True)
    assert np.abs(eng.mem.to_numpy(y) - yo).max() / 255.0 < 2e-5
    Wv = perceptual.synthetic_vgg_weights(3)
    eng.vgg_load(Wv)
    cfg = engine.default_loss_cfg()
    style = rng.uniform(-10, 270, (1, 40, 44, 3)).astype(np.float32)
    tg = eng.style_targets(eng.mem.from_numpy(style), cfg)
    losses, dy = eng.perceptual_loss(y, xd, tg, cfg)
    g = eng.mem.to_numpy(eng.tnet_backward(flat, xd, dy))
    tgo = perceptual.target_grams(style.astype(np.float64), mg.f64(Wv), cfg["style_layers"])
    lo, go, _ = perceptual.train_step(mg.f64(W), x.astype(np.float64), tgo, mg.f64(Wv))
    np.testing.assert_allclose(eng.mem.to_numpy(losses)[:3], [lo["loss"], lo["content_loss"], lo["style_loss"]], rtol=2e-4)
    want = np.concatenate([go[n].ravel() for n, _, _ in eng.param_table()])
    cos = float(np.dot(g, want) / (np.linalg.norm(g) * np.linalg.norm(want)))
    assert cos > 0.9999 and abs(np.linalg.norm(g) / np.linalg.norm(want) - 1.0) < 5e-3, cos


# ------------------------------------------------------------------ BASELINE configs[2] at its real shape
def _grads_close(e, g, want, tol):
    """per-tensor max error relative to the tensor's magnitude (floored at 5 % of the largest gradient: tensors that are
    zero by symmetry hold summation noise only) -- the measure of tests/test_path_parity.py"""
    floor = 5e-2 * max(np.abs(w).max() for w in want.values())
    bad = []
    for name, off, shape in e.param_table():
        n = int(np.prod(shape))
        err = np.abs(g[off:off + n].reshape(shape) - want[name]).max() / max(np.abs(want[name]).max(), floor)
        if not err < tol:
            bad.append((name, float(err)))
    return bad


@pytest.mark.gpu
@pytest.mark.parametrize("seed,mode", [(100, "winograd"), (101, "winograd"), (100, "direct")])
def test_hip_train_step_256_b4_real_style_image_all_48_gradients_tight(knob_hip, mode, seed):
    """256x256, batch 4, style_images/starry_night_crop.jpg (640x938), the reference initialisation: the step of BASELINE
    configs[2] (reference train.py:158-204).  The float64 oracle runs with the HIP path's own ReLU masks and pooling arg-max
    injected (tests/maskinject.py), so nothing non-smooth separates the two: losses 2e-5, EVERY one of the 48 gradient
    tensors 2e-4 of its magnitude, whole-vector relative L2 2e-4 -- once through the Winograd kernels, once through the
    direct kernels (FS_CONV_WINO=0).  (Round 2 held this shape only to cos > 0.9999 / relative L2 < 1.5e-2 and attributed
    the slack to flipped ReLU / max-pool ties; the injected comparison shows that attribution was right -- and leaves a
    real gradient bug nowhere to hide.)"""
    from faststyle_amd import utils
    from tests import maskinject
    e = get_engine("hip")
    knob_hip("FS_CONV_WINO", 1 if mode == "winograd" else 0)
    style = utils.imread(os.path.join(ROOT, "style_images", "starry_night_crop.jpg")).astype(np.float32)[None]
    Wv = perceptual.synthetic_vgg_weights(seed=3)
    cfg = engine.default_loss_cfg()
    P = tnet.init_params(seed=0)
    x = np.random.default_rng(seed).uniform(0, 255, (4, 256, 256, 3)).astype(np.float32)
    lh, lo, g, go, masks = maskinject.step_with_injected_masks(e, P, x, style, Wv, cfg)
    np.testing.assert_allclose(lh[:3], [lo["loss"], lo["content_loss"], lo["style_loss"]], rtol=2e-5)
    bad = _grads_close(e, g, go, 2e-4)
    want = np.concatenate([go[n].ravel() for n, _, _ in e.param_table()])
    cos = float(np.dot(g, want) / (np.linalg.norm(g) * np.linalg.norm(want)))
    l2 = float(np.linalg.norm(g - want) / np.linalg.norm(want))
    l2_own = float("nan")
    if seed == 100 and mode == "winograd":   # what the un-injected comparison is made of: the same oracle with its OWN decisions
        lo_own, go_own, _ = perceptual.train_step(mg.f64(P), x.astype(np.float64),
                                                  perceptual.target_grams(style.astype(np.float64), mg.f64(Wv), cfg["style_layers"]), mg.f64(Wv))
        own = np.concatenate([go_own[n].ravel() for n, _, _ in e.param_table()])
        l2_own = float(np.linalg.norm(g - own) / np.linalg.norm(own))
        assert l2_own < 1.5e-2
    print("256x256 b4 seed %d %s: masks injected cos %.9f relL2 %.2e | oracle's own masks relL2 %.2e" % (seed, mode, cos, l2, l2_own))
    assert bad == [], bad
    assert l2 < 2e-4


@pytest.fixture
def knob_hip(monkeypatch):
    e = get_engine("hip")

    def set_knob(name, value):
        monkeypatch.setenv(name, str(value))
        e.lib.fs_debug_reload_env()
        e.reset_workspaces()
    yield set_knob
    monkeypatch.undo()
    e.lib.fs_debug_reload_env()
    e.reset_workspaces()
