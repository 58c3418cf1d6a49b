"""ctypes binding of libfaststyle_hip.so (include/faststyle_hip.h, include/faststyle_io.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``python -m faststyle_amd.build``
(hipcc, gfx950).  There is NO fallback: if the shared object is missing or does not export the
full C ABI, importing the HIP path raises -- the product never routes through a CPU path.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_size_t, c_uint32, c_uint64, c_void_p)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libfaststyle_hip.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)

FS_TNET_NPARAMS = 424102
FS_TNET_NTENSORS = 48
FS_VGG_NLAYERS = 10
FS_FLAG_SAVE_FOR_BWD = 1
FS_FLAG_UPSAMPLE_DECONV = 2
FS_FLAG_BF16 = 4
FS_FLAG_PARAMS_FROZEN = 8
FS_TNET_WS_Z, FS_TNET_WS_A, FS_TNET_WS_B, FS_TNET_WS_MEAN, FS_TNET_WS_RSTD, FS_TNET_WS_H = 0, 1, 2, 3, 4, 5
FS_PAD_SAME, FS_PAD_VALID, FS_PAD_EXPLICIT = 0, 1, 2
FS_SRC_PLAIN, FS_SRC_REFLECT, FS_SRC_DILATE2 = 0, 1, 2
FS_PROFILE_FAMILIES = 23


def profile_family_names(lib):
    """Row names of fs_profile_end: one per kernel symbol ("" = unused row)."""
    return [(lib.fs_profile_family_name(f) or b"").decode() for f in range(FS_PROFILE_FAMILIES)]


VGG_LAYER_NAMES = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3",
                   "conv4_1", "conv4_2", "conv4_3"]
VGG_CIN = [3, 64, 64, 128, 128, 256, 256, 256, 512, 512]
VGG_COUT = [64, 64, 128, 128, 256, 256, 256, 512, 512, 512]

c_float_p = POINTER(c_float)


class fs_loss_cfg(Structure):
    _fields_ = [("n_content", c_int), ("content_layer", c_int * 4), ("content_weight", c_float * 4),
                ("n_style", c_int), ("style_layer", c_int * 4), ("style_weight", c_float * 4),
                ("target_gram", c_void_p * 4), ("beta", c_float)]


class fs_conv_desc(Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("y", c_void_p),
                ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int),
                ("KH", c_int), ("KW", c_int), ("stride", c_int),
                ("pad_mode", c_int), ("pad_t", c_int), ("pad_l", c_int), ("Ho", c_int), ("Wo", c_int),
                ("src_mode", c_int), ("refl", c_int),
                ("in_a", c_void_p), ("in_b", c_void_p), ("in_per_sample", c_int), ("in_relu", c_int),
                ("bias", c_void_p), ("out_relu", c_int), ("shuffle", c_int), ("stats", c_void_p),
                ("add_src", c_void_p), ("add_pad", c_int), ("w_nstride", c_longlong), ("w_wino", c_void_p),
                ("w_wino4", c_void_p), ("mask_src", c_void_p), ("pool_out", c_void_p),
                ("w_wino4t", c_void_p),
                ("inb_z", c_void_p), ("inb_mean", c_void_p), ("inb_rstd", c_void_p), ("inb_a", c_void_p), ("inb_b", c_void_p),
                ("inb_relu", c_int), ("inb_rec", c_void_p), ("route_src", c_void_p),
                ("w_wino6", c_void_p), ("w6_ws", c_void_p), ("w6_ws_bytes", ctypes.c_size_t)]


class fs_wgrad_desc(Structure):
    _fields_ = [("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p),
                ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int),
                ("KH", c_int), ("KW", c_int), ("stride", c_int), ("pad_mode", c_int),
                ("pad_t", c_int), ("pad_l", c_int), ("Ho", c_int), ("Wo", c_int),
                ("src_mode", c_int), ("refl", c_int),
                ("in_a", c_void_p), ("in_b", c_void_p), ("in_per_sample", c_int), ("in_relu", c_int),
                ("per_sample", c_int), ("scale", c_float)]


_vp10 = c_void_p * FS_VGG_NLAYERS
_vp4 = c_void_p * 4

# name -> (restype, argtypes): every symbol include/faststyle_hip.h declares
PROTOTYPES = {
    "fs_ctx_create": (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "fs_ctx_destroy": (None, [c_void_p]),
    "fs_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "fs_last_error": (c_char_p, []),
    "fs_version": (c_char_p, []),
    "fs_profile_begin": (c_int, [c_void_p]),
    "fs_profile_family_name": (c_char_p, [c_int]),
    "fs_profile_end": (c_int, [c_void_p, POINTER(ctypes.c_double * (3 * FS_PROFILE_FAMILIES))]),
    "fs_tnet_param_info": (c_int, [c_int, POINTER(c_char_p), POINTER(c_int), POINTER(c_int), POINTER(c_int * 4)]),
    "fs_tnet_out_shape": (c_int, [c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "fs_tnet_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "fs_tnet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_int]),
    "fs_tnet_invalidate": (c_int, [c_void_p]),
    "fs_tnet_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                 c_int]),
    "fs_vgg_prepared_floats": (c_size_t, []),
    "fs_vgg_prepare": (c_int, [c_void_p, POINTER(_vp10), c_void_p]),
    "fs_perceptual_workspace_bytes": (c_size_t, [c_int, c_int, c_int, POINTER(fs_loss_cfg)]),
    "fs_perceptual_loss": (c_int, [c_void_p, POINTER(_vp10), POINTER(_vp10), c_void_p, POINTER(fs_loss_cfg), c_void_p,
                                   c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fs_style_targets_workspace_bytes": (c_size_t, [c_int, c_int]),
    "fs_style_targets": (c_int, [c_void_p, POINTER(_vp10), POINTER(_vp10), POINTER(fs_loss_cfg), c_void_p, c_int, c_int,
                                 POINTER(_vp4), c_void_p, c_size_t]),
    "fs_vgg_features_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "fs_vgg_features": (c_int, [c_void_p, POINTER(_vp10), POINTER(_vp10), c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int),
                                POINTER(c_void_p), c_void_p, c_size_t]),
    "fs_gram_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fs_gram_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t]),
    "fs_gram_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t]),
    "fs_tnet_ws_tensor": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t), POINTER(c_int * 4)]),
    "fs_perceptual_ws_tensor": (c_int, [c_int, c_int, c_int, POINTER(fs_loss_cfg), c_int, POINTER(c_size_t), POINTER(c_int * 4)]),
    "fs_perceptual_ws_input": (c_int, [c_int, c_int, c_int, POINTER(fs_loss_cfg), POINTER(c_size_t), POINTER(c_size_t)]),
    "fs_loss_sqdiff": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_float, c_void_p, c_void_p]),
    "fs_loss_tv": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fs_adam_tf_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float,
                                c_float, c_longlong]),
    "fs_conv2d_fwd": (c_int, [c_void_p, POINTER(fs_conv_desc)]),
    "fs_conv2d_plan": (c_int, [POINTER(fs_conv_desc), POINTER(c_int)]),
    "fs_wino_transform_filter": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fs_wino4_transform_filter": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fs_wino4t_transform_filter": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fs_loss_sqdiff_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_float, c_void_p, c_void_p, c_void_p]),
    "fs_loss_tv_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "fs_vgg_dgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "fs_vgg_dgrad": (c_int, [c_void_p, POINTER(_vp10), POINTER(_vp10), c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int), c_void_p, c_void_p,
                             c_void_p, c_size_t]),
    "fs_conv2d_dgrad_workspace_bytes": (c_size_t, [POINTER(fs_conv_desc)]),
    "fs_conv2d_dgrad": (c_int, [c_void_p, POINTER(fs_conv_desc), c_void_p, c_void_p, c_void_p, c_size_t]),
    "fs_resizeconv_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "fs_resizeconv_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t]),
    "fs_resizeconv_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t]),
    "fs_resizeconv_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t]),
    "fs_instnorm_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fs_wino6_filter_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "fs_wino6_transform_filter": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fs_wino6_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "fs_instnorm_finalize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "fs_instnorm_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fs_instnorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fs_conv2d_wgrad_workspace_bytes": (c_size_t, [POINTER(fs_wgrad_desc)]),
    "fs_conv2d_wgrad": (c_int, [c_void_p, POINTER(fs_wgrad_desc), c_void_p, c_size_t]),
    # include/faststyle_io.h
    "fs_crc32c": (c_uint32, [c_void_p, c_size_t]),
    "fs_crc32c_masked": (c_uint32, [c_void_p, c_size_t]),
    "fs_tfrecord_scan": (c_longlong, [c_void_p, c_size_t, c_int, POINTER(c_uint64), POINTER(c_uint64), c_size_t]),
    "fs_tfrecord_frame": (c_size_t, [c_void_p, c_size_t, c_void_p]),
    "fs_example_bytes": (c_int, [c_void_p, c_size_t, c_char_p, POINTER(c_uint64), POINTER(c_uint64)]),
    "fs_example_int64": (c_int, [c_void_p, c_size_t, c_char_p, POINTER(c_longlong)]),
    "fs_resize_bicubic_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int]),
    "fs_resize_bicubic_u8x": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int]),
    "fs_u8_to_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fs_f32_to_u8": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
}


class FaststyleError(RuntimeError):
    pass


def bind(cdll):
    """Attach prototypes; raises if any declared symbol is missing."""
    missing = []
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(cdll, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise FaststyleError("shared library lacks C-ABI symbols: %s" % ", ".join(missing))
    cdll.fs_debug_reload_env.restype = None       # tests / tuning scripts only (not declared in include/*.h)
    cdll.fs_debug_reload_env.argtypes = []
    return cdll


_lib = None


def load():
    """Load the in-tree HIP library (built for gfx950).  Fails loudly when it is absent."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm ships its own libamdhip64; import it FIRST so this library binds to the same
        # HIP runtime instance (two runtimes in one process cannot both own the device).
        import torch  # noqa: F401
        path = os.environ.get("FASTSTYLE_HIP_LIB", LIB_PATH)      # tuning builds (tools/); default: the in-tree library
        if not os.path.exists(path):
            raise FaststyleError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
        _lib = bind(ctypes.CDLL(path))
    return _lib


def check(lib, rc, what=""):
    if rc != 0:
        msg = lib.fs_last_error()
        raise FaststyleError("%s failed (%d): %s" % (what or "faststyle call", rc, msg.decode() if msg else ""))
