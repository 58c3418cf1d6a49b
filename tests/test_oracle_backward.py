"""Cross-check the numpy oracle (forward AND hand-written backward) against an independent
torch-CPU float64 autograd restatement (oracle/torch_ref.py)."""
import numpy as np
import pytest
import torch

from oracle import nnops, perceptual, tnet
from oracle import torch_ref


def _t(d):
    return {k: torch.tensor(v, dtype=torch.float64) for k, v in d.items()}


@pytest.fixture(scope="module")
def setup():
    rng = np.random.default_rng(7)
    P = tnet.init_params(seed=0, dtype=np.float64)
    for k in P:                                   # non-trivial IN scale/shift
        if "INscale" in k:
            P[k] = 1.0 + 0.2 * rng.standard_normal(P[k].shape)
        if "INshift" in k:
            P[k] = 0.1 * rng.standard_normal(P[k].shape)
    W = perceptual.synthetic_vgg_weights(seed=3, dtype=np.float64)
    x = rng.uniform(0, 255, (2, 44, 52, 3))
    style = rng.uniform(0, 255, (1, 37, 45, 3))    # odd sizes exercise SAME max-pool padding
    return P, W, x, style


def test_forward_matches_torch(setup):
    P, W, x, _ = setup
    y = tnet.create_net(x, P)
    yt = torch_ref.tnet(torch.tensor(x), _t(P)).numpy()
    assert y.shape == (2,) + tnet.out_shape(44, 52) + (3,)
    np.testing.assert_allclose(y, yt, rtol=0, atol=1e-8)


def test_train_step_grads_match_torch_autograd(setup):
    P, W, x, style = setup
    tg = perceptual.target_grams(style, W, ("conv1_2", "conv2_2", "conv3_3", "conv4_3"))
    losses, grads, y = perceptual.train_step(P, x, tg, W, beta=1e-4)

    Pt = {k: v.requires_grad_(True) for k, v in _t(P).items()}
    Wt = _t(W)
    xt = torch.tensor(x)
    with torch.no_grad():
        ct = [torch_ref.vgg(xt, Wt)["conv3_3"]]
        tgt = [torch_ref.gram(torch_ref.vgg(torch.tensor(style), Wt)[n])
               for n in ("conv1_2", "conv2_2", "conv3_3", "conv4_3")]
    for a, b in zip(tg, tgt):
        np.testing.assert_allclose(a, b.numpy(), rtol=1e-10, atol=1e-12)
    yt = torch_ref.tnet(xt, Pt)
    L, cl, sl, tv = torch_ref.loss(yt, ct, tgt, Wt, beta=1e-4)
    L.backward()
    np.testing.assert_allclose(losses["loss"], L.item(), rtol=1e-10)
    np.testing.assert_allclose(losses["content_loss"], cl.item(), rtol=1e-10)
    np.testing.assert_allclose(losses["style_loss"], sl.item(), rtol=1e-10)
    for k in P:
        g = Pt[k].grad.numpy()
        scale = np.abs(g).max() + 1e-30
        assert np.abs(grads[k] - g).max() / scale < 1e-8, k
    # SURVEY.md §8a invariant: dL/dINshift2 identical for every residual block
    for i in range(1, 5):
        np.testing.assert_allclose(grads["resblock_%d/INshift2" % i], grads["resblock_0/INshift2"],
                                   rtol=1e-6, atol=1e-9 * np.abs(grads["resblock_0/INshift2"]).max())


def test_reflect_pad_adjoint():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 45, 47, 2))
    dy = rng.standard_normal((1, 125, 127, 2))
    lhs = np.sum(nnops.reflect_pad(x, 40) * dy)
    rhs = np.sum(x * nnops.reflect_pad_bwd(dy, 40))
    np.testing.assert_allclose(lhs, rhs, rtol=1e-12)


def test_adam_tf_matches_closed_form():
    p = {"a": np.array([1.0, -2.0])}
    g = {"a": np.array([0.5, -0.25])}
    m = {"a": np.zeros(2)}
    v = {"a": np.zeros(2)}
    perceptual.adam_tf(p, g, m, v, 1)
    # step 1: m=(1-b1)g, v=(1-b2)g^2, lr_t=lr*sqrt(1-b2)/(1-b1) -> theta -= lr*g/(|g|+eps*...)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    want = np.array([1.0, -2.0]) - lr_t * (0.1 * g["a"]) / (np.sqrt(0.001 * g["a"] ** 2) + 1e-8)
    np.testing.assert_allclose(p["a"], want, rtol=1e-14)


def test_deconv_method_forward_and_backward_match_torch():
    """--upsample_method deconv (im_transf_net.py:57-63, 158-190): conv2d_transpose layers."""
    rng = np.random.default_rng(11)
    P = tnet.init_params(seed=1, upsample_method="deconv", dtype=np.float64)
    assert P["upsample_0/W"].shape == (3, 3, 32, 64) and P["upsample_2/W"].shape == (9, 9, 3, 16)
    x = rng.uniform(0, 255, (1, 44, 48, 3))
    y, cache = tnet.create_net(x, P, "deconv", keep=True)
    assert y.shape == (1, 44, 48, 3)
    Pt = {k: v.requires_grad_(True) for k, v in _t(P).items()}
    yt = torch_ref.tnet(torch.tensor(x), Pt, "deconv")
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=0, atol=1e-8)
    dy = rng.standard_normal(y.shape)
    (yt * torch.tensor(dy)).sum().backward()
    grads = tnet.create_net_bwd(dy, P, cache)
    for k in P:
        g = Pt[k].grad.numpy()
        assert np.abs(grads[k] - g).max() / (np.abs(g).max() + 1e-30) < 1e-8, k
