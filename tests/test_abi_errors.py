"""Error paths of the C ABI (include/faststyle_hip.h: "every call returns 0 on success, a negative code on error; fs_last_error()
holds the message") through ctypes -- on the CPU against the emulator build of the same sources, on the GPU against the product library.
Every case returns BEFORE anything is launched or dereferenced, so the tensor arguments may be arbitrary non-null addresses."""
import ctypes

import numpy as np
import pytest

from faststyle_amd import _lib as L
from tests.backends import engine_params, get_engine

P = ctypes.c_void_p


@pytest.fixture(params=engine_params())
def eng(request):
    return get_engine(request.param)


def err(e):
    return (e.lib.fs_last_error() or b"").decode()


def test_tnet_entry_points_reject_bad_arguments(eng):
    e, lib, ctx = eng, eng.lib, eng.ctx
    buf = e.mem.zeros((1024,))
    p = e.mem.ptr(buf)
    need = lib.fs_tnet_workspace_bytes(1, 64, 64, 0)
    assert need > 0
    # null tensor: -1
    assert lib.fs_tnet_forward(ctx, None, p, 1, 64, 64, p, p, need, 0) == -1 and "null" in err(e)
    # REFLECT-40 needs H, W >= 41: -2, the message carries the numbers
    assert lib.fs_tnet_forward(ctx, p, p, 1, 40, 64, p, p, need, 0) == -2 and "41" in err(e) and "40" in err(e)
    assert lib.fs_tnet_forward(ctx, p, p, 0, 64, 64, p, p, need, 0) == -2
    # workspace one byte short: -3 with both sizes
    assert lib.fs_tnet_forward(ctx, p, p, 1, 64, 64, p, p, need - 1, 0) == -3 and str(need) in err(e)
    assert lib.fs_tnet_backward(ctx, p, p, p, 1, 64, 64, p, p, need - 1, 0) == -3 and str(need) in err(e)
    # the bf16 mode is inference-only and covers the resize-conv models
    assert lib.fs_tnet_forward(ctx, p, p, 1, 64, 64, p, p, need, L.FS_FLAG_BF16 | L.FS_FLAG_SAVE_FOR_BWD) == -2 and "BF16" in err(e)
    assert lib.fs_tnet_forward(ctx, p, p, 1, 64, 64, p, p, need, L.FS_FLAG_BF16 | L.FS_FLAG_UPSAMPLE_DECONV) == -2
    assert lib.fs_tnet_workspace_bytes(1, 64, 64, L.FS_FLAG_BF16 | L.FS_FLAG_SAVE_FOR_BWD) == 0
    assert lib.fs_tnet_workspace_bytes(1, 40, 64, 0) == 0
    # inspection entry points
    off, dims = ctypes.c_size_t(), (ctypes.c_int * 4)()
    assert lib.fs_tnet_ws_tensor(1, 64, 64, 0, 16, L.FS_TNET_WS_Z, ctypes.byref(off), ctypes.byref(dims)) == -2 and "16" in err(e)
    assert lib.fs_tnet_ws_tensor(1, 64, 64, 0, 5, L.FS_TNET_WS_H, ctypes.byref(off), ctypes.byref(dims)) == -2
    assert lib.fs_tnet_ws_tensor(1, 64, 64, L.FS_FLAG_BF16, 0, L.FS_TNET_WS_Z, ctypes.byref(off), ctypes.byref(dims)) == -2
    name = ctypes.c_char_p()
    assert lib.fs_tnet_param_info(48, ctypes.byref(name), None, None, None) == -1 and "48" in err(e)
    assert lib.fs_tnet_invalidate(None) == -1


def test_backward_refuses_a_workspace_filled_by_another_forward(eng):
    """fs_tnet_backward reads what fs_tnet_forward left in the caller's workspace; handed a workspace this context filled with another
    upsample method (FS_FLAG_UPSAMPLE_DECONV: other filter layouts, another launch sequence) or another shape it returns -5 instead of
    gradients of garbage."""
    from oracle import tnet
    e, lib, ctx = eng, eng.lib, eng.ctx
    N, H, W = 1, 44, 48
    flat = e.mem.from_numpy(e.flatten_params(tnet.init_params(seed=0), scope=""))
    x = e.mem.from_numpy(np.random.default_rng(0).uniform(0, 255, (N, H, W, 3)).astype(np.float32))
    nbytes = max(lib.fs_tnet_workspace_bytes(N, H, W, L.FS_FLAG_SAVE_FOR_BWD),
                 lib.fs_tnet_workspace_bytes(N, H, W, L.FS_FLAG_SAVE_FOR_BWD | L.FS_FLAG_UPSAMPLE_DECONV),
                 lib.fs_tnet_workspace_bytes(2, H, W, L.FS_FLAG_SAVE_FOR_BWD))
    ws = e.mem.zeros((nbytes // 4,))
    Ho, Wo = e.tnet_out_shape(H, W)
    y, dy, g = e.mem.zeros((N, Ho, Wo, 3)), e.mem.zeros((2, Ho, Wo, 3)), e.mem.zeros((L.FS_TNET_NPARAMS,))
    p = e.mem.ptr
    e._sync_stream()
    assert lib.fs_tnet_forward(ctx, p(flat), p(x), N, H, W, p(y), p(ws), nbytes, L.FS_FLAG_SAVE_FOR_BWD) == 0
    assert lib.fs_tnet_backward(ctx, p(flat), p(x), p(dy), N, H, W, p(g), p(ws), nbytes, L.FS_FLAG_UPSAMPLE_DECONV) == -5
    assert "resize" in err(e) and "deconv" in err(e)
    assert lib.fs_tnet_backward(ctx, p(flat), p(x), p(dy), 2, H, W, p(g), p(ws), nbytes, 0) == -5 and "N=2" in err(e)
    assert lib.fs_tnet_backward(ctx, p(flat), p(x), p(dy), N, H, W, p(g), p(ws), nbytes, 0) == 0          # the matching call runs
    assert np.isfinite(e.mem.to_numpy(g)).all()
    lib.fs_tnet_invalidate(ctx)                                                                           # forgets the record: no refusal
    assert lib.fs_tnet_backward(ctx, p(flat), p(x), p(dy), N, H, W, p(g), p(ws), nbytes, 0) == 0


def test_loss_and_builder_entry_points_reject_bad_arguments(eng):
    e, lib, ctx = eng, eng.lib, eng.ctx
    buf = e.mem.zeros((1024,))
    p = e.mem.ptr(buf)
    vp = (P * L.FS_VGG_NLAYERS)(*([p] * L.FS_VGG_NLAYERS))
    cfg = L.fs_loss_cfg()
    cfg.n_content = 1
    cfg.content_layer[0] = 6
    cfg.n_style = 1
    cfg.style_layer[0] = 10                      # conv4_3 is layer 9: out of range
    assert lib.fs_perceptual_workspace_bytes(1, 64, 64, ctypes.byref(cfg)) == 0
    assert lib.fs_perceptual_loss(ctx, ctypes.byref(vp), ctypes.byref(vp), p, ctypes.byref(cfg), p, p, 1, 64, 64, p, p, p, 1 << 30) == -2
    assert "style layer" in err(e)
    cfg.style_layer[0] = 9
    cfg.n_content = 5
    assert lib.fs_perceptual_loss(ctx, ctypes.byref(vp), ctypes.byref(vp), p, ctypes.byref(cfg), p, p, 1, 64, 64, p, p, p, 1 << 30) == -2
    cfg.n_content = 1
    # a style layer without its target Gram matrix
    assert lib.fs_perceptual_loss(ctx, ctypes.byref(vp), ctypes.byref(vp), p, ctypes.byref(cfg), p, p, 1, 64, 64, p, p, p, 1 << 30) == -2
    assert "target_gram[0]" in err(e)
    cfg.target_gram[0] = p
    need = lib.fs_perceptual_workspace_bytes(1, 64, 64, ctypes.byref(cfg))
    assert need > 0
    assert lib.fs_perceptual_loss(ctx, ctypes.byref(vp), ctypes.byref(vp), p, ctypes.byref(cfg), p, p, 1, 64, 64, p, p, p, need - 4) == -3
    assert lib.fs_perceptual_loss(ctx, ctypes.byref(vp), ctypes.byref(vp), p, ctypes.byref(cfg), p, None, 1, 64, 64, p, p, p, need) == -1
    gp = (P * 4)(p, p, p, p)
    assert lib.fs_style_targets(ctx, ctypes.byref(vp), ctypes.byref(vp), ctypes.byref(cfg), p, 64, 64, ctypes.byref(gp), p, 16) == -3
    lay = (ctypes.c_int * 1)(10)
    op = (P * 1)(p)
    assert lib.fs_vgg_features(ctx, ctypes.byref(vp), ctypes.byref(vp), p, 1, 64, 64, 1, lay, op, p, 1 << 30) == -2 and "10" in err(e)
    assert lib.fs_vgg_features_workspace_bytes(1, 64, 64, 10) == 0
    # Gram matrices: C a multiple of 4, of 128 beyond 128
    assert lib.fs_gram_fwd(ctx, p, 1, 64, 130, p, p, 1 << 20) == -2 and "130" in err(e)
    assert lib.fs_gram_workspace_bytes(1, 64, 130) == 0
    assert lib.fs_gram_bwd(ctx, p, p, 1, 64, 64, p, p, 64) == -3
    # Adam: t is the 1-based step
    assert lib.fs_adam_tf_step(ctx, p, p, p, p, 16, 1e-3, 0.9, 0.999, 1e-8, 0) == -2 and "1-based" in err(e)
    # losses.content_loss / style_loss: n must be whole periods
    assert lib.fs_loss_sqdiff(ctx, p, p, 7, 16, 1.0, p, p) == -2 and "period" in err(e)
    assert lib.fs_loss_sqdiff(ctx, p, p, 0, 16, 1.0, p, p) == -1


def test_single_op_entry_points_reject_bad_arguments(eng):
    e, lib, ctx = eng, eng.lib, eng.ctx
    buf = e.mem.zeros((1024,))
    p = e.mem.ptr(buf)

    def desc(**kw):
        d = L.fs_conv_desc()
        d.x = d.w = d.y = p
        d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = 1, 16, 16, 8, 64, 3, 3, 1
        d.pad_mode = L.FS_PAD_SAME
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    d = desc(Cin=6)                                          # Cin: 3 or a multiple of 4
    assert lib.fs_conv2d_fwd(ctx, ctypes.byref(d)) == -2 and "Cin" in err(e)
    d = desc(shuffle=1, Cout=66)
    assert lib.fs_conv2d_fwd(ctx, ctypes.byref(d)) == -2 and "shuffle" in err(e)
    d = desc(pad_mode=L.FS_PAD_VALID, H=2, W=2)              # a 3x3 VALID conv of a 2x2 image has no output
    assert lib.fs_conv2d_fwd(ctx, ctypes.byref(d)) == -2 and "empty" in err(e)
    d = desc(pool_out=p)                                     # the fused max-pool exists in the Winograd epilogues only
    assert lib.fs_conv2d_fwd(ctx, ctypes.byref(d)) == -2 and "pool_out" in err(e)
    d = desc(x=None)
    assert lib.fs_conv2d_fwd(ctx, ctypes.byref(d)) == -1
    assert lib.fs_conv2d_fwd(None, ctypes.byref(desc())) == -1
    # filter transforms: channel multiples of the kernels' steps
    assert lib.fs_wino4t_transform_filter(ctx, p, 12, 64, p) == -2 and "12" in err(e)      # Cin % 8
    assert lib.fs_wino4t_transform_filter(ctx, p, 8, 32, p) == -2                          # Cout % 64
    assert lib.fs_wino4_transform_filter(ctx, p, 6, 64, p) == -2
    assert lib.fs_wino_transform_filter(ctx, p, 12, 64, p) == -2
    # instance-norm backward
    assert lib.fs_instnorm_bwd(ctx, p, p, p, p, p, p, 1, 1, 64, 260, p, p, p, p, 1 << 30) == -2 and "256" in err(e)
    need = lib.fs_instnorm_bwd_workspace_bytes(2, 4096, 64)
    assert lib.fs_instnorm_bwd(ctx, p, p, p, p, p, p, 1, 2, 4096, 64, p, p, p, p, need - 4) == -3 and str(need) in err(e)
    assert lib.fs_instnorm_bwd(ctx, p, p, p, p, p, p, 3, 2, 4096, 64, p, p, p, p, need) == -2
    # filter gradients
    w = L.fs_wgrad_desc()
    w.x = w.dy = w.dw = p
    w.N, w.H, w.W, w.Cin, w.Cout, w.KH, w.KW, w.stride = 1, 16, 16, 192, 64, 3, 3, 1
    w.pad_mode = L.FS_PAD_SAME
    assert lib.fs_conv2d_wgrad(ctx, ctypes.byref(w), p, 1 << 30) == -2 and "128" in err(e)
    assert lib.fs_conv2d_wgrad_workspace_bytes(ctypes.byref(w)) == 0
    w.Cin = 64
    need = lib.fs_conv2d_wgrad_workspace_bytes(ctypes.byref(w))
    assert need > 0 and lib.fs_conv2d_wgrad(ctx, ctypes.byref(w), p, need - 4) == -3
    # u8 <-> f32 helpers: alignment contract
    assert lib.fs_u8_to_f32(ctx, P(p + 1), 16, p) == -1 and "aligned" in err(e)
    assert lib.fs_resize_bicubic_u8(ctx, p, 0, 4, p, 4, 4) == -1
    assert lib.fs_resize_bicubic_u8x(ctx, p, 4, 4, 2, p, 4, 4) == -2 and "pixel_bytes" in err(e)
    assert lib.fs_resize_bicubic_u8x(ctx, p, 4, 0, 4, p, 4, 4) == -1


def test_round6_named_exports_reject_bad_arguments(eng):
    """The entry points added in round 6 (value + gradient losses, fs_vgg_dgrad, fs_conv2d_dgrad, fs_resizeconv_*, fs_instnorm_apply, fs_wino6_*)."""
    e, lib, ctx = eng, eng.lib, eng.ctx
    buf = e.mem.zeros((4096,))
    p = e.mem.ptr(buf)
    # losses
    assert lib.fs_loss_sqdiff_grad(ctx, p, p, 0, 16, 1.0, p, p, p) == -1
    assert lib.fs_loss_sqdiff_grad(ctx, p, p, 5, 16, 1.0, p, p, p) == -2 and "16" in err(e)            # not whole periods
    assert lib.fs_loss_sqdiff_grad(ctx, p, p, 4, 16, 1.0, p, None, p) == -1
    assert lib.fs_loss_tv_grad(ctx, p, 1, 0, 4, 3, 1.0, p, p, 0, p) == -2
    assert lib.fs_loss_tv_grad(ctx, p, 1, 4, 4, 3, 1.0, p, None, 0, p) == -1
    # VGG input gradient
    vp = (ctypes.c_void_p * L.FS_VGG_NLAYERS)(*([p] * L.FS_VGG_NLAYERS))
    lay = (ctypes.c_int * 2)(6, 6)
    g2 = (ctypes.c_void_p * 2)(p, p)
    need = lib.fs_vgg_dgrad_workspace_bytes(1, 32, 32, 6)
    assert need > 0 and lib.fs_vgg_dgrad_workspace_bytes(1, 32, 32, 10) == 0
    assert lib.fs_vgg_dgrad(ctx, ctypes.byref(vp), ctypes.byref(vp), None, p, 1, 32, 32, 2, lay, g2, p, p, need) == -2 and "twice" in err(e)
    lay = (ctypes.c_int * 2)(6, 11)
    assert lib.fs_vgg_dgrad(ctx, ctypes.byref(vp), ctypes.byref(vp), None, p, 1, 32, 32, 2, lay, g2, p, p, need) == -2 and "11" in err(e)
    lay = (ctypes.c_int * 2)(3, 6)
    assert lib.fs_vgg_dgrad(ctx, ctypes.byref(vp), ctypes.byref(vp), None, p, 1, 32, 32, 2, lay, g2, p, p, need - 4) == -3
    g2 = (ctypes.c_void_p * 2)(p, None)
    assert lib.fs_vgg_dgrad(ctx, ctypes.byref(vp), ctypes.byref(vp), None, p, 1, 32, 32, 2, lay, g2, p, p, need) == -1
    # conv input gradient
    d = L.fs_conv_desc()
    d.w = p
    d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = 1, 16, 16, 8, 64, 3, 3, 1
    d.pad_mode = L.FS_PAD_SAME
    need = lib.fs_conv2d_dgrad_workspace_bytes(ctypes.byref(d))
    assert need == 9 * 8 * 64 * 4
    assert lib.fs_conv2d_dgrad(ctx, ctypes.byref(d), p, p, p, need - 4) == -3
    d.stride = 3
    assert lib.fs_conv2d_dgrad(ctx, ctypes.byref(d), p, p, p, need) == -2 and "stride" in err(e)
    d.stride, d.Cin = 1, 6
    assert lib.fs_conv2d_dgrad(ctx, ctypes.byref(d), p, p, p, need) == -2 and "multiples of 4" in err(e)
    d.Cin = 8
    assert lib.fs_conv2d_dgrad(ctx, ctypes.byref(d), None, p, p, need) == -1
    # resize-conv
    assert lib.fs_resizeconv_workspace_bytes(1, 8, 8, 6, 16) == 0
    need = lib.fs_resizeconv_workspace_bytes(1, 8, 8, 32, 16)
    assert need > 0
    assert lib.fs_resizeconv_fwd(ctx, p, p, 1, 8, 8, 6, 16, p, p, 1 << 20) == -2 and "multiples of 4" in err(e)
    assert lib.fs_resizeconv_dgrad(ctx, p, p, 1, 8, 8, 32, 16, p, p, need - 4) == -3
    assert lib.fs_resizeconv_wgrad(ctx, p, None, 1, 8, 8, 32, 16, p, p, need) == -1
    # instance-norm apply
    assert lib.fs_instnorm_apply(ctx, p, p, p, 1, 4, 4, 64, 3, None, None, None, p) == -2
    assert lib.fs_instnorm_apply(ctx, p, p, p, 1, 4, 4, 64, 1, p, None, None, p) == -2 and "residual" in err(e)     # the skip sum takes mode 0
    assert lib.fs_instnorm_apply(ctx, p, p, p, 1, 4, 4, 64, 0, p, p, None, p) == -2                                # skip_a and skip_b together
    assert lib.fs_instnorm_apply(ctx, p, None, p, 1, 4, 4, 64, 0, None, None, None, p) == -1
    # split-bf16 filter pieces
    assert lib.fs_wino6_filter_bytes(16, 128) == 0 and lib.fs_wino6_filter_bytes(32, 64) == 0 and lib.fs_wino6_filter_bytes(32, 128) == 36 * 32 * 128 * 6
    assert lib.fs_wino6_transform_filter(ctx, p, 48, 128, p) == -2 and "48" in err(e)
    assert lib.fs_wino6_workspace_bytes(0, 8, 8, 32, 128) == 0
