"""The second-generation kernels of round 2 (fs_wino2, fs_wgrad2 and its batched launch, the fused max-pool, the one-tile
Gram, the vector-ALU conv1_1 input gradient) against the first-generation / direct kernels they replaced, on the GPU, over
batch sizes and image shapes the fixed-shape parity tests do not visit (odd batches, non-square and non-multiple-of-16
images, a batch larger than the persistent grids).  Both sides are this library, so the comparison needs no oracle run and
covers shapes the numpy oracle would take minutes for; the oracle pins the generations at the fixed shapes elsewhere."""
import numpy as np
import pytest

from faststyle_amd import engine
from oracle import perceptual, tnet
from tests.backends import get_engine

OLD = {"FS_WGRAD2": 0, "FS_TNET_WGRAD_BATCH": 0, "FS_WINO_V": 1, "FS_VGG_POOL_FUSED": 0, "FS_C3_VALU": 0, "FS_GRAM_SAME": 0}
DIRECT = dict(OLD, FS_CONV_WINO=0)

SHAPES = [(3, 256, 256), (5, 192, 160), (1, 252, 332), (2, 100, 76), (32, 64, 64), (7, 128, 144), (9, 72, 88)]


@pytest.fixture
def knobs(monkeypatch):
    e = get_engine("hip")

    def set_knobs(d):
        monkeypatch.undo()
        for k, v in d.items():
            monkeypatch.setenv(k, str(v))
        e.lib.fs_debug_reload_env()
        e.reset_workspaces()
    yield set_knobs
    monkeypatch.undo()
    e.lib.fs_debug_reload_env()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES, ids=["%dx%dx%d" % s for s in SHAPES])
def test_kernel_generations_agree(knobs, shape):
    e = get_engine("hip")
    n, h, w = shape
    rng = np.random.default_rng(h * 1000 + w + n)
    Wv = perceptual.synthetic_vgg_weights(seed=3)
    cfg = engine.default_loss_cfg()
    cfg["beta"] = 1e-4
    P = tnet.init_params(seed=0)
    style = rng.uniform(0, 255, (1, 90, 70, 3)).astype(np.float32)
    x = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    out = {}
    for name, kn in (("new", {}), ("old", OLD), ("direct", DIRECT)):
        knobs(kn)
        e.vgg_load(Wv)
        flat = e.mem.from_numpy(e.flatten_params(P, scope=""))
        tg = e.style_targets(e.mem.from_numpy(style), cfg)
        xd = e.mem.from_numpy(x)
        y = e.tnet_forward(flat, xd, save_for_bwd=True)
        losses, dy = e.perceptual_loss(y, xd, tg, cfg)
        g = e.mem.to_numpy(e.tnet_backward(flat, xd, dy)).astype(np.float64)
        out[name] = (e.mem.to_numpy(y).astype(np.float64), e.mem.to_numpy(losses).astype(np.float64), g)
    y0, l0, g0 = out["direct"]
    assert np.isfinite(g0).all() and np.linalg.norm(g0) > 0
    for name in ("new", "old"):
        y1, l1, g1 = out[name]
        assert np.abs(y1 - y0).max() < 2e-5 * 255, name                       # forward pixels (north-star budget 1e-3)
        np.testing.assert_allclose(l1, l0, rtol=2e-5, err_msg=name)           # loss, content, style, tv
        cos = float(np.dot(g1, g0) / (np.linalg.norm(g1) * np.linalg.norm(g0)))
        l2 = float(np.linalg.norm(g1 - g0) / np.linalg.norm(g0))
        # (ReLU / max-pool ties flip between summation orders: the same envelope as against the oracle)
        assert cos > 0.9999 and l2 < 1.5e-2, (name, shape, cos, l2)


FWD_SHAPES = [(1, 123, 77), (2, 301, 203), (1, 480, 640), (3, 97, 191), (1, 41, 41), (5, 64, 500)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", FWD_SHAPES, ids=["%dx%dx%d" % s for s in FWD_SHAPES])
def test_forward_kernel_choices_agree_on_arbitrary_frames(knobs, shape):
    """stylize_image.py takes any frame size: the residual convs go through either Winograd generation or the direct kernel
    depending on the grid (FS_TNET_WINO), with ragged 16x16 blocks at the image edge -- all three must give the same pixels."""
    e = get_engine("hip")
    n, h, w = shape
    x = np.random.default_rng(n * 7 + h + w).uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    P = tnet.init_params(seed=1)
    ys = {}
    for name, kn in (("wino2", {"FS_TNET_WINO": 2}), ("wino1", {"FS_TNET_WINO": 2, "FS_WINO_V": 1}), ("direct", {"FS_TNET_WINO": 0})):
        knobs(kn)
        flat = e.mem.from_numpy(e.flatten_params(P, scope=""))
        ys[name] = e.mem.to_numpy(e.tnet_forward(flat, e.mem.from_numpy(x))).astype(np.float64)
    for name in ("wino2", "wino1"):
        assert ys[name].shape == ys["direct"].shape
        assert np.abs(ys[name] - ys["direct"]).max() < 2e-5 * 255, (name, shape)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 100, 76), (5, 64, 80)], ids=["2x100x76", "5x64x80"])
def test_filter_gradient_branch_on_the_second_stream_changes_nothing(knobs, shape):
    """fs_tnet_backward forks the filter gradients onto a second stream for batches of >= FS_SIDE_MIN_PIXELS pixels (the training
    batches; below that the fork costs more than it returns).  The fork only changes WHERE kernels run: forced on at a small shape
    (FS_SIDE_MIN_PIXELS=0) the 48 gradient tensors must equal the single-stream ones bit for bit."""
    e = get_engine("hip")
    n, h, w = shape
    rng = np.random.default_rng(n + h + w)
    P = tnet.init_params(seed=0)
    x = rng.uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    grads = {}
    for name, kn in (("forked", {"FS_SIDE_MIN_PIXELS": 0}), ("one_stream", {"FS_SIDE_MIN_PIXELS": 1 << 30})):
        knobs(kn)
        flat = e.mem.from_numpy(e.flatten_params(P, scope=""))
        xd = e.mem.from_numpy(x)
        y = e.tnet_forward(flat, xd, save_for_bwd=True)
        dy = e.mem.from_numpy(np.random.default_rng(7).standard_normal(tuple(int(v) for v in y.shape)).astype(np.float32))
        grads[name] = e.mem.to_numpy(e.tnet_backward(flat, xd, dy))
    assert np.isfinite(grads["forked"]).all() and np.linalg.norm(grads["forked"]) > 0
    assert np.array_equal(grads["forked"], grads["one_stream"])


@pytest.mark.gpu
@pytest.mark.parametrize("min_pixels", [1 << 30, 0], ids=["one_stream", "forked"])
def test_graph_replayed_train_steps_report_the_eager_losses(knobs, min_pixels):
    """Trainer(use_graph=True) replays ONE captured hipGraph per step; the four reported scalars and the parameters after three steps
    must be those of the eager loop, bit for bit, with and without the second-stream fork inside the graph.  (Regression: the loss
    buffer used to be cleared by hipMemsetAsync -- as a node of a single-stream graph it left garbage in losses[2..3] from the second
    replay on; the training batches always forked, so only `tv_loss` of small-batch runs ever showed it.)"""
    from faststyle_amd import trainer, vgg16, im_transf_net
    e = get_engine("hip")
    knobs({"FS_SIDE_MIN_PIXELS": min_pixels})
    rng = np.random.default_rng(3)
    Wv = vgg16.synthetic_weights(3)
    style = rng.uniform(0, 255, (1, 64, 64, 3)).astype(np.float32)
    params = e.flatten_params(im_transf_net.initial_variables(seed=0), scope="")
    xs = [rng.uniform(0, 255, (2, 128, 128, 3)).astype(np.float32) for _ in range(3)]
    got = {}
    for use_graph in (False, True):
        tr = trainer.Trainer(e, params.copy(), Wv, style, engine.default_loss_cfg(), use_graph=use_graph)
        ls = [e.mem.to_numpy(tr.step(e.mem.from_numpy(x))).copy() for x in xs]
        assert (tr.graph is not None) == use_graph
        got[use_graph] = (np.stack(ls), tr.params_numpy())
    assert np.isfinite(got[True][0]).all() and (got[True][0][:, 3] == 0).all()          # beta = 0: no TV term
    assert np.array_equal(got[True][0], got[False][0]) and np.array_equal(got[True][1], got[False][1])
