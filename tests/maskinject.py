"""TEST INFRASTRUCTURE: the non-differentiable decisions of one HIP train step (ReLU sign patterns, max-pool arg-max),
read back from the tensors the step left in its workspaces (fs_tnet_ws_tensor / fs_perceptual_ws_tensor), in the form
oracle.perceptual.train_step(masks=...) injects.

Why: a pre-activation within float32 noise of zero is rectified differently by the float32 HIP forward and the float64
oracle forward (likewise two pooling-window entries within noise of each other); each such flip moves the gradient by a
whole term, so oracle-vs-HIP gradients could only be held to a loose L2 envelope.  With the HIP path's own decisions fed
to the oracle's backward, everything left is smooth arithmetic and all 48 tensors agree to rounding -- a wrong gradient
can no longer hide inside the envelope."""
import numpy as np

from faststyle_amd import _lib as L
from oracle import nnops, perceptual

RELU_UNITS = {0: "initconv_0", 1: "initconv_1", 2: "initconv_2", 3: "resblock_0", 5: "resblock_1", 7: "resblock_2",
              9: "resblock_3", 11: "resblock_4", 13: "upsample_0", 14: "upsample_1"}
POOLS = {"conv1_2": "pool1", "conv2_2": "pool2", "conv3_3": "pool3"}


def hip_masks(eng, N, H, W, cfg, upsample_method="resize"):
    """Call right after tnet_forward(save_for_bwd=True) + perceptual_loss of the shape (N,H,W) on `eng`."""
    to = eng.mem.to_numpy
    tn = {}
    for unit, name in RELU_UNITS.items():
        z = to(eng.tnet_saved(N, H, W, unit, L.FS_TNET_WS_Z, upsample_method)).astype(np.float64)
        a = to(eng.tnet_saved(N, H, W, unit, L.FS_TNET_WS_A, upsample_method)).astype(np.float64)
        b = to(eng.tnet_saved(N, H, W, unit, L.FS_TNET_WS_B, upsample_method)).astype(np.float64)
        # the kernels rectify fmaf(z, a, b) (fs_elem.hip in_bwd, the consumers' staging loads): the float64 product of two
        # float32 values is exact, so this is the sign the fused multiply-add sees
        tn[name] = (z * a[:, None, None, :] + b[:, None, None, :]) > 0
    Ho, Wo = eng.tnet_out_shape(H, W)
    vg = {}
    lmax = max(L.VGG_LAYER_NAMES.index(n) for n in list(cfg["style_layers"]) + list(cfg["content_layers"]))
    for name in L.VGG_LAYER_NAMES[:lmax + 1]:
        act = to(eng.vgg_saved(N, Ho, Wo, cfg, name))[:N]          # the y half
        vg[name] = act > 0
        if name in POOLS and L.VGG_LAYER_NAMES.index(name) < lmax:
            vg[POOLS[name] + "/idx"] = nnops.max_pool_2x2(act)[1]   # first maximum in window order, as vgg_bwd_route does
    return {"tnet": tn, "vgg": vg}


def _f64(d):
    return {k: np.asarray(v, np.float64) for k, v in d.items()}


def step_with_injected_masks(eng, P, x, style, Wv, cfg):
    """One train step (no optimiser) of `eng` on batch x, and the float64 oracle's step on the same inputs with the
    engine's decisions injected.  Returns (engine losses[4], oracle losses dict, engine flat gradient, oracle gradient
    dict, the masks)."""
    eng.vgg_load(Wv)
    flat = eng.mem.from_numpy(eng.flatten_params(P, scope=""))
    tg = eng.style_targets(eng.mem.from_numpy(style), cfg)
    xd = eng.mem.from_numpy(x)
    y = eng.tnet_forward(flat, xd, save_for_bwd=True)
    losses, dy = eng.perceptual_loss(y, xd, tg, cfg)
    masks = hip_masks(eng, x.shape[0], x.shape[1], x.shape[2], cfg)      # before the backward reuses any buffer
    g = eng.mem.to_numpy(eng.tnet_backward(flat, xd, dy)).astype(np.float64)
    lh = eng.mem.to_numpy(losses).astype(np.float64)
    W64 = _f64(Wv)
    tgo = perceptual.target_grams(style.astype(np.float64), W64, cfg["style_layers"])
    kw = dict(content_layers=tuple(cfg["content_layers"]), style_layers=tuple(cfg["style_layers"]),
              content_weights=tuple(cfg["content_weights"]), style_weights=tuple(cfg["style_weights"]))
    lo, go, _ = perceptual.train_step(_f64(P), x.astype(np.float64), tgo, W64, beta=cfg.get("beta", 0.0), masks=masks, **kw)
    return lh, lo, g, go, masks
