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
    depending on the grid (FS_TNET_WINO / FS_TNET_WINO4), with ragged 16x16 blocks at the image edge -- all four must give the same pixels."""
    e = get_engine("hip")
    n, h, w = shape
    x = np.random.default_rng(n * 7 + h + w).uniform(0, 255, (n, h, w, 3)).astype(np.float32)
    P = tnet.init_params(seed=1)
    ys = {}
    # (FS_TNET_WINO4=0 on the F(2x2) rows: with it on -- the default -- every launch of >= 64 items goes to the F(4x4) kernel whatever
    # FS_TNET_WINO says, and the two F(2x2) generations, still reachable for ineligible shapes, would lose their large-grid coverage)
    for name, kn in (("wino4t", {"FS_TNET_WINO": 2, "FS_TNET_WINO4": 2}), ("wino2", {"FS_TNET_WINO": 2, "FS_TNET_WINO4": 0}),
                     ("wino1", {"FS_TNET_WINO": 2, "FS_WINO_V": 1, "FS_TNET_WINO4": 0}), ("direct", {"FS_TNET_WINO": 0})):
        knobs(kn)
        flat = e.mem.from_numpy(e.flatten_params(P, scope=""))
        ys[name] = e.mem.to_numpy(e.tnet_forward(flat, e.mem.from_numpy(x))).astype(np.float64)
    for name in ("wino4t", "wino2", "wino1"):
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


def graph_node_types(cuda_graph):
    """hipGraphNodeType of every node of a captured torch CUDAGraph(keep_graph=True), through libamdhip64 (test-only plumbing)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    g = ctypes.c_void_p(cuda_graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(g, None, ctypes.byref(n)) == 0 and n.value > 0
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(g, nodes, ctypes.byref(n)) == 0
    types = []
    for k in range(n.value):
        t = ctypes.c_int(-1)
        assert hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[k]), ctypes.byref(t)) == 0
        types.append(t.value)
    return types


HIP_NODE_KERNEL, HIP_NODE_MEMCPY, HIP_NODE_MEMSET = 0, 1, 2      # hipGraphNodeType (hip_runtime_api.h)


@pytest.mark.gpu
@pytest.mark.parametrize("min_pixels", [1 << 30, 0], ids=["one_stream", "forked"])
def test_graph_replayed_batch4_steps_equal_the_eager_loop_in_all_losses_and_gradients(knobs, min_pixels):
    """The configuration every GPU of BASELINE configs[3] runs (256 x 256, batch 4), TV term on: over four graph replays all four reported scalars,
    the 48 gradient tensors of EVERY step and the parameters at the end equal the eager loop's bit for bit -- and the captured graph holds
    kernel nodes only: no memset node (the round-4 bug: a 16-byte memset node of a single-stream graph left stale loss scalars from the second
    replay on) and no memcpy node (the staging copies of fs_perceptual_loss are gone: y and the batch live inside its workspace)."""
    from faststyle_amd import trainer, vgg16, im_transf_net
    e = get_engine("hip")
    knobs({"FS_SIDE_MIN_PIXELS": min_pixels})
    rng = np.random.default_rng(11)
    Wv = vgg16.synthetic_weights(3)
    style = rng.uniform(0, 255, (1, 96, 80, 3)).astype(np.float32)
    params = e.flatten_params(im_transf_net.initial_variables(seed=0), scope="")
    cfg = engine.default_loss_cfg()
    cfg["beta"] = 1e-4
    xs = [rng.uniform(0, 255, (4, 256, 256, 3)).astype(np.float32) for _ in range(4)]
    got = {}
    trainer.Trainer.KEEP_GRAPH = True
    try:
        for use_graph in (False, True):
            tr = trainer.Trainer(e, params.copy(), Wv, style, cfg, use_graph=use_graph)
            ls, gs = [], []
            for x in xs:
                ls.append(e.mem.to_numpy(tr.step(e.mem.from_numpy(x))).copy())
                gs.append(e.mem.to_numpy(tr.grads).copy())
            assert (tr.graph is not None) == use_graph
            if use_graph:
                types = graph_node_types(tr.graph)
                assert len(types) > 100 and HIP_NODE_MEMSET not in types and HIP_NODE_MEMCPY not in types, sorted(set(types))
            got[use_graph] = (np.stack(ls), np.stack(gs), tr.params_numpy())
    finally:
        trainer.Trainer.KEEP_GRAPH = False
    L = got[True][0]
    assert np.isfinite(L).all() and (L[:, 1:] > 0).all() and np.allclose(L[:, 0], L[:, 1:].sum(axis=1), rtol=1e-6)   # content, style, beta * tv all live
    for k in range(3):
        assert np.array_equal(got[True][k], got[False][k]), ("losses", "gradients", "parameters")[k]


@pytest.mark.gpu
def test_graph_replayed_720p_frames_equal_the_eager_frames_and_two_stylizers_do_not_mix(knobs):
    """FrameStylizer replays ONE captured hipGraph per frame, with frozen parameters: the graph holds no filter re-layout kernels, they were
    built once into the stylizer's OWN workspace.  (a) three different 720p frames through the graph equal the eager frames byte for byte and
    the graph has kernel nodes only; (b) the advisor's round-4 scenario: a SECOND stylizer of the same shape with ANOTHER checkpoint on the
    same engine, and plain same-shape forwards in between, must not change what the first one's graph produces (a shared per-shape workspace
    would be rewritten under it); (c) freeing the parameter tensor of a dropped stylizer and allocating a new one -- possibly at the same
    address -- gives the new model's frames, not stale ones (fs_tnet_invalidate)."""
    from faststyle_amd import stream
    import torch
    e = get_engine("hip")
    knobs({})
    rng = np.random.default_rng(5)
    P1, P2 = tnet.init_params(seed=1), tnet.init_params(seed=2)
    f1 = e.mem.from_numpy(e.flatten_params(P1, scope=""))
    f2 = e.mem.from_numpy(e.flatten_params(P2, scope=""))
    frames = [rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8) for _ in range(3)]
    stream.FrameStylizer.KEEP_GRAPH = True
    try:
        eager1 = stream.FrameStylizer(e, f1, 720, 1280, use_graph=False)
        want1 = [eager1(f) for f in frames]
        eager2 = stream.FrameStylizer(e, f2, 720, 1280, use_graph=False)
        want2 = [eager2(f) for f in frames]
        assert not np.array_equal(want1[0], want2[0])
        g1 = stream.FrameStylizer(e, f1, 720, 1280)
        assert np.array_equal(g1(frames[0]), want1[0])
        types = graph_node_types(g1._graph)
        assert len(types) > 30 and set(types) == {HIP_NODE_KERNEL}, sorted(set(types))
        g2 = stream.FrameStylizer(e, f2, 720, 1280)                       # same shape, other checkpoint, same engine
        assert np.array_equal(g2(frames[1]), want2[1])
        x = e.mem.from_numpy(frames[2].astype(np.float32)[None])
        e.tnet_forward(f2, x)                                              # a plain forward of that shape (the engine's shared workspace)
        e.tnet_forward(f2, x, frozen=True)
        for k in (1, 2, 0):
            assert np.array_equal(g1(frames[k]), want1[k]), k              # the first graph still stylizes with ITS checkpoint
            assert np.array_equal(g2(frames[k]), want2[k]), k
        # (c) a dropped stylizer's buffers may come back at the same addresses
        del g2, eager2, f2
        torch.cuda.synchronize()
        f3 = e.mem.from_numpy(e.flatten_params(tnet.init_params(seed=3), scope=""))
        want3 = e.mem.to_numpy(e.tnet_forward(f3, x))
        g3 = stream.FrameStylizer(e, f3, 720, 1280, swap_rb=False)
        got3 = g3(frames[2]).astype(np.float32)
        assert np.abs(got3 - np.floor(np.clip(want3[0], 0, 255))).max() <= 1.0   # (u8 truncation of the same forward)
    finally:
        stream.FrameStylizer.KEEP_GRAPH = False


@pytest.mark.gpu
def test_pipelined_stylizer_two_frames_in_flight_equals_the_frame_at_a_time_results(knobs):
    """stream.PipelinedStylizer (round 6): two FrameStylizer lanes on streams of their own, frame i + 1 uploaded and stylized while frame i is still on the
    device.  Seven different frames (an odd count: the lanes end unevenly) come back in order and byte for byte equal to the one-frame-at-a-time stylizer's;
    submit() refuses a third frame in flight; a plain same-shape forward on the engine between two submits changes nothing (every lane owns its workspace)."""
    from faststyle_amd import stream
    e = get_engine("hip")
    knobs({})
    rng = np.random.default_rng(9)
    f1 = e.mem.from_numpy(e.flatten_params(tnet.init_params(seed=4), scope=""))
    H, W = 136, 200
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(7)]
    one = stream.FrameStylizer(e, f1, H, W)
    want = [one(f) for f in frames]
    ps = stream.PipelinedStylizer(e, f1, H, W)
    got = list(ps.run(frames))
    assert len(got) == 7 and all(np.array_equal(g, w) for g, w in zip(got, want))
    ps.submit(frames[3])
    e.tnet_forward(f1, e.mem.from_numpy(frames[0].astype(np.float32)[None]))      # the engine's shared workspace, between two submits
    ps.submit(frames[5])
    with pytest.raises(Exception):
        ps.submit(frames[6])
    assert np.array_equal(ps.fetch(), want[3]) and np.array_equal(ps.fetch(), want[5])
    assert not np.array_equal(want[3], want[5])
