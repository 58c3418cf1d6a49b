"""Thin Python driver over the C ABI: owns a context, workspaces and the pointer plumbing.

Device memory, streams and (in ``trainer.py``) ``torch.distributed`` come from PyTorch-ROCm --
plumbing only; every FLOP of the path runs in libfaststyle_hip.so.  The memory provider is
injectable (``mem``) so the parity tests can drive the very same code with host arrays against
the kernel emulator build; the product always uses ``TorchMem`` on a ROCm device.
"""
import ctypes
from collections import OrderedDict

import numpy as np

from . import _lib as L


class TorchMem(object):
    """Device tensors from PyTorch-ROCm on the current HIP stream."""

    supports_graphs = True     # hipGraph capture of a step (trainer.py); host-array providers of the tests have none

    def __init__(self, device="cuda:0"):
        import torch
        if not torch.cuda.is_available():
            raise L.FaststyleError("no ROCm/HIP device visible: the faststyle engine has no CPU path")
        self.torch = torch
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)

    def empty(self, shape):
        return self.torch.empty(tuple(int(s) for s in shape), dtype=self.torch.float32, device=self.device)

    def zeros(self, shape):
        return self.torch.zeros(tuple(int(s) for s in shape), dtype=self.torch.float32, device=self.device)

    def from_numpy(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def to_numpy(self, t):
        return t.detach().cpu().numpy()

    def ptr(self, t):
        if t is None:
            return None
        assert t.is_contiguous() and t.dtype == self.torch.float32 and t.is_cuda
        return t.data_ptr()

    def stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def device_index(self):
        return self.device.index or 0

    def view(self, t, offset, shape):
        n = int(np.prod(shape))
        return t.view(-1)[offset:offset + n].view(*shape)

    # ---- input pipeline (datapipe.py): u8 uploads and the HBM-resident shuffle queue
    def upload_u8(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(self.device, non_blocking=True)

    def ptr_u8(self, t):
        assert t.is_contiguous() and t.dtype == self.torch.uint8 and t.is_cuda
        return t.data_ptr()

    def gather_rows(self, store, idx):
        return store.index_select(0, self.torch.as_tensor(np.asarray(idx, dtype=np.int64), device=self.device))

    def copy_row(self, store, src, dst):
        store[dst].copy_(store[src])


def default_loss_cfg():
    """train.py:52-70 defaults: content conv3_3 x1.0; style conv1_2/2_2/3_3/4_3 x5.0."""
    return dict(content_layers=["conv3_3"], content_weights=[1.0],
                style_layers=["conv1_2", "conv2_2", "conv3_3", "conv4_3"], style_weights=[5.0] * 4, beta=0.0)


class Engine(object):
    # Workspaces are cached per shape (a backward must find the workspace its forward filled; a captured hipGraph replays raw
    # pointers into its own).  The cache is bounded by BYTES, least recently used first: stylizing a directory of mixed
    # resolutions must not pile up one multi-GB workspace per size.  Entries a live hipGraph replays into are pinned by
    # their trainer (pin_workspaces) and are never evicted while pinned.
    MAX_CACHED_BYTES = 24 << 30    # of 288 GB; the largest single shape of the BASELINE configs (1080p batch 8 fp32) needs 17 GB
    MAX_CACHED_SHAPES = 8

    def __init__(self, mem=None, lib=None):
        self.mem = mem if mem is not None else TorchMem()      # initialises the HIP runtime PyTorch ships
        self.lib = lib if lib is not None else L.load()
        ctx = ctypes.c_void_p()
        L.check(self.lib, self.lib.fs_ctx_create(self.mem.device_index(), ctypes.c_void_p(self.mem.stream()),
                                                 ctypes.byref(ctx)), "fs_ctx_create")
        self.ctx = ctx
        self._tnet_ws = {}
        self._perc_ws = {}
        self._pinned = {}          # id(workspace tensor) -> pin count
        self._tnet_nbytes = {}     # (N, H, W, bf16) -> fs_tnet_workspace_bytes, see _tnet_key
        self._perc_io = {}         # perceptual workspace key -> (y offset, content offset), see perceptual_inputs
        self._private = []         # [(weakref to a new_tnet_workspace() tensor, nbytes)]: counted in the byte budget while alive
        self._keep = []

    def close(self):
        if self.ctx:
            self.lib.fs_ctx_destroy(self.ctx)
            self.ctx = None

    def _sync_stream(self):
        self.lib.fs_ctx_set_stream(self.ctx, ctypes.c_void_p(self.mem.stream()))

    # ------------------------------------------------------------------ parameters
    def param_table(self):
        """[(name, offset, shape)] of the 48 tensors in the flat parameter buffer (ckpt order)."""
        out = []
        for i in range(L.FS_TNET_NTENSORS):
            name = ctypes.c_char_p()
            off = ctypes.c_int()
            nd = ctypes.c_int()
            dims = (ctypes.c_int * 4)()
            L.check(self.lib, self.lib.fs_tnet_param_info(i, ctypes.byref(name), ctypes.byref(off), ctypes.byref(nd),
                                                          ctypes.byref(dims)), "fs_tnet_param_info")
            out.append((name.value.decode(), off.value, tuple(dims[k] for k in range(nd.value))))
        return out

    def param_table_for(self, upsample_method="resize"):
        """param_table() with the conv2d_transpose filter shapes [K,K,Cout,Cin] of --upsample_method
        deconv (im_transf_net.py:174) -- same element counts, so the flat offsets are identical."""
        tab = self.param_table()
        if upsample_method == "deconv":
            tab = [(n, o, (s[0], s[1], s[3], s[2]) if n.startswith("upsample_") and len(s) == 4 else s) for n, o, s in tab]
        return tab

    def flatten_params(self, tensors, scope="img_t_net/", upsample_method="resize"):
        """dict name->ndarray (checkpoint names) -> flat float32 vector in table order."""
        flat = np.empty(L.FS_TNET_NPARAMS, dtype=np.float32)
        for name, off, shape in self.param_table_for(upsample_method):
            a = tensors.get(scope + name, tensors.get(name))
            if a is None:
                raise L.FaststyleError("checkpoint lacks tensor %s%s" % (scope, name))
            if tuple(a.shape) != shape:
                raise L.FaststyleError("tensor %s has shape %s, expected %s (wrong --upsample_method?)"
                                       % (name, a.shape, shape))
            flat[off:off + a.size] = np.asarray(a, dtype=np.float32).ravel()
        return flat

    def unflatten_params(self, flat, scope="img_t_net/", upsample_method="resize"):
        flat = np.asarray(flat)
        return OrderedDict((scope + name, flat[off:off + int(np.prod(shape))].reshape(shape).copy())
                           for name, off, shape in self.param_table_for(upsample_method))

    # ------------------------------------------------------------------ transform net
    def tnet_out_shape(self, H, W):
        ho, wo = ctypes.c_int(), ctypes.c_int()
        L.check(self.lib, self.lib.fs_tnet_out_shape(H, W, ctypes.byref(ho), ctypes.byref(wo)), "fs_tnet_out_shape")
        return ho.value, wo.value

    def _evict(self, cache, incoming_bytes):
        """Drop least-recently-used, unpinned entries of `cache` until `incoming_bytes` more fit the budget."""
        def total():
            self._private = [(r, nb) for r, nb in self._private if r() is not None]
            return sum(nb for _, nb in self._tnet_ws.values()) + sum(nb for _, nb in self._perc_ws.values()) + sum(nb for _, nb in self._private)
        dropped = False
        for key in list(cache):
            if len(cache) < self.MAX_CACHED_SHAPES and total() + incoming_bytes <= self.MAX_CACHED_BYTES:
                break
            if self._pinned.get(id(cache[key][0]), 0) == 0:
                cache.pop(key)
                dropped = True
        if dropped:
            self.invalidate_frozen()

    def invalidate_frozen(self):
        """A workspace (or a parameter buffer) the library may remember by ADDRESS went away: the caching allocator can hand the same address
        out again, so FS_FLAG_PARAMS_FROZEN must not trust pointer identity across this point (fs_tnet_invalidate)."""
        if self.ctx:
            self.lib.fs_tnet_invalidate(self.ctx)

    def pin_workspaces(self, entries, on=True):
        """entries: workspace tensors a captured hipGraph replays into; pinned entries survive cache eviction."""
        for t in entries:
            n = self._pinned.get(id(t), 0) + (1 if on else -1)
            if n > 0:
                self._pinned[id(t)] = n
            else:
                self._pinned.pop(id(t), None)

    def pin_last_used(self, tnet=True, perceptual=False):
        """The capture contract: whoever captures engine calls into a hipGraph calls this right after the capture and keeps the
        returned list (the workspace tensors the captured calls used = the most recently used entry of each cache, now
        pinned against eviction AND kept alive by the list) for as long as the graph lives; release_pins() afterwards."""
        held = []
        if tnet and self._tnet_ws:
            held.append(next(reversed(self._tnet_ws.values()))[0])
        if perceptual and self._perc_ws:
            held.append(next(reversed(self._perc_ws.values()))[0])
        self.pin_workspaces(held, True)
        return held

    def release_pins(self, held):
        if held:
            self.pin_workspaces(held, False)
            del held[:]

    def _tnet_key(self, N, H, W, bf16=False):
        # the layout (and with it the size) depends on the library's FS_TNET_WINO knob as the LIBRARY sees it (read once and
        # cached there): ask the library, not os.environ
        # (memoised: the C entry point rebuilds the whole layout, 40-65 us of host time -- per frame in an eager loop;
        # reset_workspaces(), which every knob flip is followed by, drops the memo)
        nbytes = self._tnet_nbytes.get((N, H, W, bf16))
        if nbytes is None:
            nbytes = self.lib.fs_tnet_workspace_bytes(N, H, W, L.FS_FLAG_BF16 if bf16 else L.FS_FLAG_SAVE_FOR_BWD)
            if nbytes == 0:
                raise L.FaststyleError("bad transform-net shape %s" % ((N, H, W, bf16),))
            self._tnet_nbytes[(N, H, W, bf16)] = nbytes
        return (N, H, W, bf16, nbytes)

    def _tnet_workspace(self, N, H, W, bf16=False):
        key = self._tnet_key(N, H, W, bf16)
        if key not in self._tnet_ws:
            self._evict(self._tnet_ws, key[4])
            self._tnet_ws[key] = (self.mem.empty((key[4] // 4,)), key[4])
        else:
            self._tnet_ws[key] = self._tnet_ws.pop(key)                      # mark as most recently used
        return self._tnet_ws[key]

    def reset_workspaces(self):
        """Drop every cached workspace (their sizes were planned under the tuning knobs of the time; tests that flip FS_*
        knobs with fs_debug_reload_env call this, production code never needs to)."""
        self._sync_stream()
        self._tnet_ws.clear()
        self._perc_ws.clear()
        self._pinned.clear()
        self._tnet_nbytes.clear()
        self._perc_io.clear()
        self.invalidate_frozen()

    def new_tnet_workspace(self, N, H, W, bf16=False):
        """A transform-net workspace of the caller's own (not in the shape cache): what a hipGraph capturer with frozen=True replays into --
        the re-laid-out filters inside it belong to ONE parameter set, and no other call of this engine may rewrite them (a shared, per-shape
        workspace would be rewritten by any other same-shape forward: another FrameStylizer with another checkpoint, a Trainer, an eval)."""
        import weakref
        nbytes = self._tnet_key(N, H, W, bf16)[4]
        # a private workspace counts in MAX_CACHED_BYTES for as long as its owner keeps it: make room among the cached (unpinned) entries first,
        # largest cache first -- several stylizers plus a Trainer otherwise add up outside every budget (the caller's workspace itself is never refused)
        for cache in sorted((self._tnet_ws, self._perc_ws), key=lambda c: -sum(nb for _, nb in c.values())):
            self._evict(cache, nbytes)
        self.invalidate_frozen()          # (a fresh allocation may re-use an address the library remembers)
        t = self.mem.empty((nbytes // 4,))
        try:
            self._private.append((weakref.ref(t), nbytes))
        except TypeError:                 # (a memory backend whose buffers take no weak references: not counted)
            pass
        return (t, nbytes)

    @staticmethod
    def _method_flag(upsample_method):
        assert upsample_method in ("resize", "deconv")
        return L.FS_FLAG_UPSAMPLE_DECONV if upsample_method == "deconv" else 0

    def tnet_forward(self, params, x, save_for_bwd=False, upsample_method="resize", bf16=False, frozen=False, workspace=None, out=None):
        """create_net(x, upsample_method): x [N,H,W,3] float32 RGB 0..255 -> y [N,Ho,Wo,3].
        bf16=True: the mixed-precision inference path (FS_FLAG_BF16; ~1e-2 of the pixel range off the fp32 path).
        frozen=True: `params` is not modified between calls (FS_FLAG_PARAMS_FROZEN: the filter re-layouts run once per sequence).
        workspace: a (tensor, nbytes) pair from new_tnet_workspace() instead of the engine's shared per-shape one.
        out: write y into this [N,Ho,Wo,3] tensor (e.g. perceptual_inputs()[0]) instead of a fresh one."""
        self._sync_stream()
        N, H, W, C = (int(s) for s in x.shape)
        assert C == 3
        Ho, Wo = self.tnet_out_shape(H, W)
        ws, nbytes = workspace if workspace is not None else self._tnet_workspace(N, H, W, bf16)
        y = out if out is not None else self.mem.empty((N, Ho, Wo, 3))
        assert tuple(int(v) for v in y.shape) == (N, Ho, Wo, 3)
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_tnet_forward(self.ctx, p(params), p(x), N, H, W, p(y), p(ws), nbytes,
                                                   (L.FS_FLAG_SAVE_FOR_BWD if save_for_bwd else 0) |
                                                   (L.FS_FLAG_BF16 if bf16 else 0) | (L.FS_FLAG_PARAMS_FROZEN if frozen and not bf16 else 0) |
                                                   self._method_flag(upsample_method)), "fs_tnet_forward")
        return y

    def tnet_backward(self, params, x, dy, grads=None, upsample_method="resize"):
        """Gradient of the 48 tensors given dL/dy; must follow tnet_forward(save_for_bwd=True)."""
        self._sync_stream()
        N, H, W, _ = (int(s) for s in x.shape)
        if self._tnet_key(N, H, W) not in self._tnet_ws:
            raise L.FaststyleError("tnet_backward(%dx%dx%d) without a tnet_forward(save_for_bwd=True) of that shape" % (N, H, W))
        ws, nbytes = self._tnet_workspace(N, H, W)
        if grads is None:
            grads = self.mem.empty((L.FS_TNET_NPARAMS,))
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_tnet_backward(self.ctx, p(params), p(x), p(dy), N, H, W, p(grads), p(ws), nbytes,
                                                    self._method_flag(upsample_method)), "fs_tnet_backward")
        return grads

    # ------------------------------------------------------------------ VGG / losses
    def vgg_load(self, weights):
        """weights: mapping in the npz convention of vgg16.load_weights (vgg16.py:257-266):
        'conv1_1_W' [3,3,3,64], 'conv1_1_b' [64], ...  Keeps device copies + dgrad re-layout."""
        self._sync_stream()
        self.vgg_w, self.vgg_b = [], []
        for i, n in enumerate(L.VGG_LAYER_NAMES):
            w = np.asarray(weights[n + "_W"], dtype=np.float32)
            b = np.asarray(weights[n + "_b"], dtype=np.float32)
            if w.shape != (3, 3, L.VGG_CIN[i], L.VGG_COUT[i]) or b.shape != (L.VGG_COUT[i],):
                raise L.FaststyleError("VGG weight %s has shape %s" % (n, w.shape))
            self.vgg_w.append(self.mem.from_numpy(w))
            self.vgg_b.append(self.mem.from_numpy(b))
        self._wp = (ctypes.c_void_p * L.FS_VGG_NLAYERS)(*[self.mem.ptr(t) for t in self.vgg_w])
        self._bp = (ctypes.c_void_p * L.FS_VGG_NLAYERS)(*[self.mem.ptr(t) for t in self.vgg_b])
        self.vgg_prepared = self.mem.empty((self.lib.fs_vgg_prepared_floats(),))
        L.check(self.lib, self.lib.fs_vgg_prepare(self.ctx, ctypes.byref(self._wp), self.mem.ptr(self.vgg_prepared)),
                "fs_vgg_prepare")

    def _cfg(self, cfg, target_grams=None):
        c = L.fs_loss_cfg()
        # the reference asserts equal lengths (losses.py:28, :58 iterate by index over both lists)
        if len(cfg["content_layers"]) != len(cfg["content_weights"]) or len(cfg["style_layers"]) != len(cfg["style_weights"]):
            raise L.FaststyleError("loss layers and weights differ in length: %s / %s, %s / %s" % (
                cfg["content_layers"], cfg["content_weights"], cfg["style_layers"], cfg["style_weights"]))
        c.n_content = len(cfg["content_layers"])
        for i, (n, w) in enumerate(zip(cfg["content_layers"], cfg["content_weights"])):
            c.content_layer[i] = L.VGG_LAYER_NAMES.index(n)
            c.content_weight[i] = w
        c.n_style = len(cfg["style_layers"])
        for i, (n, w) in enumerate(zip(cfg["style_layers"], cfg["style_weights"])):
            c.style_layer[i] = L.VGG_LAYER_NAMES.index(n)
            c.style_weight[i] = w
            if target_grams is not None:
                c.target_gram[i] = self.mem.ptr(target_grams[i])
        c.beta = cfg.get("beta", 0.0)
        return c

    def style_targets(self, style_img, cfg):
        """utils.get_grams on the style image (train.py:144-151): list of [1,C,C] device tensors."""
        self._sync_stream()
        _, H, W, _ = (int(s) for s in style_img.shape)
        c = self._cfg(cfg)
        grams = [self.mem.empty((1, L.VGG_COUT[c.style_layer[i]], L.VGG_COUT[c.style_layer[i]]))
                 for i in range(c.n_style)]
        gp = (ctypes.c_void_p * 4)(*([self.mem.ptr(g) for g in grams] + [None] * (4 - len(grams))))
        nbytes = self.lib.fs_style_targets_workspace_bytes(H, W)
        ws = self.mem.empty((nbytes // 4,))
        L.check(self.lib, self.lib.fs_style_targets(self.ctx, ctypes.byref(self._wp), ctypes.byref(self._bp), ctypes.byref(c),
                                                    self.mem.ptr(style_img), H, W, ctypes.byref(gp), self.mem.ptr(ws), nbytes),
                "fs_style_targets")
        return grams

    def _perceptual_workspace(self, N, H, W, cfg, c):
        key = (N, H, W, tuple(cfg["content_layers"]), tuple(cfg["style_layers"]))
        nbytes = self.lib.fs_perceptual_workspace_bytes(N, H, W, ctypes.byref(c))
        if key not in self._perc_ws or self._perc_ws[key][1] != nbytes:
            self._perc_ws.pop(key, None)
            self._evict(self._perc_ws, nbytes)
            self._perc_ws[key] = (self.mem.empty((nbytes // 4,)), nbytes)
        else:
            self._perc_ws[key] = self._perc_ws.pop(key)
        return self._perc_ws[key]

    def perceptual_inputs(self, N, H, W, cfg, with_ws=False):
        """(y, content) views INSIDE the (cached) perceptual workspace of this shape, where fs_perceptual_loss stages its two inputs
        (fs_perceptual_ws_input): a caller that lets tnet_forward(out=y) write there and keeps its batch in `content` saves both
        staging copies of a step.  content is None when cfg has no content layer.  The views die with the workspace: a caller that keeps
        one across calls (a hipGraph capturer) asks for the workspace tensor too (with_ws=True -> (y, content, ws)) and pins exactly that
        tensor (pin_workspaces([ws]))."""
        c = self._cfg(cfg)
        ws, _ = self._perceptual_workspace(N, H, W, cfg, c)
        key = (N, H, W, tuple(cfg["content_layers"]), tuple(cfg["style_layers"]))
        if key not in self._perc_io:      # (the offsets are a pure function of the key; an eager loop asks every step)
            yo, co = ctypes.c_size_t(), ctypes.c_size_t()
            L.check(self.lib, self.lib.fs_perceptual_ws_input(N, H, W, ctypes.byref(c), ctypes.byref(yo), ctypes.byref(co)), "fs_perceptual_ws_input")
            self._perc_io[key] = (yo.value, None if co.value == ctypes.c_size_t(-1).value else co.value)
        yo, co = self._perc_io[key]
        out = (self.mem.view(ws, yo, (N, H, W, 3)), (None if co is None else self.mem.view(ws, co, (N, H, W, 3))))
        return out + (ws,) if with_ws else out

    def perceptual_loss(self, y, content, target_grams, cfg):
        """loss = content + style + beta*tv (train.py:164-184) and dL/dy.
        Returns (losses[4] device tensor {loss, content, style, beta*tv}, dy)."""
        self._sync_stream()
        N, H, W, _ = (int(s) for s in y.shape)
        if tuple(content.shape) != tuple(y.shape):
            # the reference feeds VGG(batch) into placeholders shaped like VGG(Y) (train.py:171-176, 250-254)
            raise L.FaststyleError("content batch %s and net output %s differ: training sizes must be multiples "
                                   "of 4 (create_net maps H -> 4*ceil(ceil((H+80)/2)/2)-80)" % (tuple(content.shape), tuple(y.shape)))
        c = self._cfg(cfg, target_grams)
        ws, nbytes = self._perceptual_workspace(N, H, W, cfg, c)
        losses = self.mem.empty((4,))
        dy = self.mem.empty((N, H, W, 3))
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_perceptual_loss(self.ctx, ctypes.byref(self._wp), ctypes.byref(self._bp),
                                                      p(self.vgg_prepared), ctypes.byref(c), p(y), p(content), N, H, W,
                                                      p(losses), p(dy), p(ws), nbytes), "fs_perceptual_loss")
        return losses, dy

    # ------------------------------------------------------------------ builder-level pieces (vgg16 / utils / losses)
    def vgg_features(self, x, layer_names):
        """libs/vgg16.py:36-220: post-ReLU activations `convX_Y` of x [N,H,W,3] (RGB 0..255); needs vgg_load()."""
        self._sync_stream()
        N, H, W, _ = (int(s) for s in x.shape)
        ids = [L.VGG_LAYER_NAMES.index(n) for n in layer_names]
        outs, h, w, shapes = [], H, W, {}
        for l in range(max(ids) + 1):
            shapes[l] = (N, h, w, L.VGG_COUT[l])
            if l in (1, 3, 6, 9):                        # max_pool 2x2 s2 SAME after conv1_2 / 2_2 / 3_3 / 4_3
                h, w = (h + 1) // 2, (w + 1) // 2
        outs = [self.mem.empty(shapes[l]) for l in ids]
        nbytes = self.lib.fs_vgg_features_workspace_bytes(N, H, W, max(ids))
        ws = self.mem.empty((nbytes // 4,))
        lay = (ctypes.c_int * len(ids))(*ids)
        op = (ctypes.c_void_p * len(ids))(*[self.mem.ptr(t) for t in outs])
        L.check(self.lib, self.lib.fs_vgg_features(self.ctx, ctypes.byref(self._wp), ctypes.byref(self._bp), self.mem.ptr(x), N, H, W,
                                                   len(ids), lay, op, self.mem.ptr(ws), nbytes), "fs_vgg_features")
        return outs

    def gram(self, feat):
        """utils.get_grams for one layer (utils.py:76-82): feat [N,h,w,c] -> [N,c,c] = F^T F / (h*w*c)  (fs_gram_fwd)."""
        self._sync_stream()
        N, h, w, c = (int(s) for s in feat.shape)
        G = self.mem.empty((N, c, c))
        nbytes = self.lib.fs_gram_workspace_bytes(N, h * w, c)
        ws = self.mem.empty((max(nbytes // 4, 1),))
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_gram_fwd(self.ctx, p(feat), N, h * w, c, p(G), p(ws), nbytes), "fs_gram_fwd")
        return G

    def gram_bwd(self, feat, dG):
        """Gradient through utils.get_grams: dF = F (dG + dG^T) / (h*w*c), feat [N,h,w,c], dG [N,c,c]  (fs_gram_bwd)."""
        self._sync_stream()
        N, h, w, c = (int(s) for s in feat.shape)
        dF = self.mem.empty((N, h, w, c))
        nbytes = self.lib.fs_gram_workspace_bytes(N, h * w, c)
        ws = self.mem.empty((max(nbytes // 4, 1),))
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_gram_bwd(self.ctx, p(feat), p(dG), N, h * w, c, p(dF), p(ws), nbytes), "fs_gram_bwd")
        return dF

    # ------------------------------------------------------------------ inspection of the saved tensors (parity tests)
    def tnet_saved(self, N, H, W, unit, what, upsample_method="resize"):
        """View of a tensor fs_tnet_forward(save_for_bwd=True) left in the (cached) workspace of this shape."""
        off, dims = ctypes.c_size_t(), (ctypes.c_int * 4)()
        L.check(self.lib, self.lib.fs_tnet_ws_tensor(N, H, W, L.FS_FLAG_SAVE_FOR_BWD | self._method_flag(upsample_method), unit, what,
                                                     ctypes.byref(off), ctypes.byref(dims)), "fs_tnet_ws_tensor")
        key = self._tnet_key(N, H, W)
        if key not in self._tnet_ws:
            raise L.FaststyleError("no transform-net workspace of shape %s" % ((N, H, W),))
        shape = tuple(dims) if what in (L.FS_TNET_WS_Z, L.FS_TNET_WS_H) else (dims[0], dims[1])
        return self.mem.view(self._tnet_ws[key][0], off.value, shape)

    def vgg_saved(self, N, H, W, cfg, layer_name):
        """View of the post-ReLU activations fs_perceptual_loss left in its (cached) workspace: [NB,h,w,c], the first N
        samples are the net outputs, the next N (up to the last content layer) the content batch."""
        c = self._cfg(cfg)
        off, dims = ctypes.c_size_t(), (ctypes.c_int * 4)()
        L.check(self.lib, self.lib.fs_perceptual_ws_tensor(N, H, W, ctypes.byref(c), L.VGG_LAYER_NAMES.index(layer_name),
                                                           ctypes.byref(off), ctypes.byref(dims)), "fs_perceptual_ws_tensor")
        key = (N, H, W, tuple(cfg["content_layers"]), tuple(cfg["style_layers"]))
        if key not in self._perc_ws:
            raise L.FaststyleError("no perceptual workspace of shape %s" % ((N, H, W),))
        return self.mem.view(self._perc_ws[key][0], off.value, tuple(dims))

    def loss_sqdiff(self, x, t, scale):
        """scale * sum((x - t)^2) with t broadcast over the leading (batch) axis when smaller; device scalar [1]."""
        self._sync_stream()
        n, period = int(np.prod(x.shape)), int(np.prod(t.shape))
        assert n % period == 0
        out, scratch = self.mem.empty((1,)), self.mem.empty((1024,))
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_loss_sqdiff(self.ctx, p(x), p(t), period, n, float(scale), p(out), p(scratch)), "fs_loss_sqdiff")
        return out

    def loss_tv(self, x):
        self._sync_stream()
        N, H, W, C = (int(s) for s in x.shape)
        out, scratch = self.mem.empty((1,)), self.mem.empty((1024,))
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_loss_tv(self.ctx, p(x), N, H, W, C, p(out), p(scratch)), "fs_loss_tv")
        return out

    # ------------------------------------------------------------------ value + gradient forms (round 6; faststyle_amd/autograd.py builds on these)
    def loss_sqdiff_grad(self, x, t, scale):
        """(scale * sum((x - t)^2) as a device scalar [1], d/dx = 2 * scale * (x - t)) -- losses.py:32-37 / :61-64 with their derivative."""
        self._sync_stream()
        n, period = int(np.prod(x.shape)), int(np.prod(t.shape))
        assert n % period == 0
        out, scratch, grad = self.mem.empty((1,)), self.mem.empty((1024,)), self.mem.empty(tuple(x.shape))
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_loss_sqdiff_grad(self.ctx, p(x), p(t), period, n, float(scale), p(out), p(grad), p(scratch)), "fs_loss_sqdiff_grad")
        return out, grad

    def loss_tv_grad(self, x, scale=1.0, grad=None):
        """(scale * TV(x) [1], scale * dTV/dx); grad given: the gradient is ADDED to it (train.py:184's beta * tv on top of the perceptual gradient)."""
        self._sync_stream()
        N, H, W, C = (int(s) for s in x.shape)
        out, scratch = self.mem.empty((1,)), self.mem.empty((1024,))
        acc = grad is not None
        if grad is None:
            grad = self.mem.empty((N, H, W, C))
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_loss_tv_grad(self.ctx, p(x), N, H, W, C, float(scale), p(out), p(grad), int(acc), p(scratch)), "fs_loss_tv_grad")
        return out, grad

    def vgg_dgrad(self, x, layer_names, dfeats, use_prepared=True):
        """Adjoint of vgg_features: dfeats[i] = dL/d(activation layer_names[i]) -> dL/dx [N,H,W,3] (fs_vgg_dgrad; the forward is recomputed)."""
        self._sync_stream()
        N, H, W, _ = (int(s) for s in x.shape)
        ids = [L.VGG_LAYER_NAMES.index(n) for n in layer_names]
        dx = self.mem.empty((N, H, W, 3))
        nbytes = self.lib.fs_vgg_dgrad_workspace_bytes(N, H, W, max(ids))
        ws = self.mem.empty((nbytes // 4,))
        lay = (ctypes.c_int * len(ids))(*ids)
        gp = (ctypes.c_void_p * len(ids))(*[self.mem.ptr(t) for t in dfeats])
        L.check(self.lib, self.lib.fs_vgg_dgrad(self.ctx, ctypes.byref(self._wp), ctypes.byref(self._bp),
                                                self.mem.ptr(self.vgg_prepared) if use_prepared else None, self.mem.ptr(x), N, H, W,
                                                len(ids), lay, gp, self.mem.ptr(dx), self.mem.ptr(ws), nbytes), "fs_vgg_dgrad")
        return dx

    def conv2d_dgrad(self, dy, w, in_hw, stride=1, padding="SAME"):
        """tf.nn.conv2d_backprop_input: dy [N,Ho,Wo,Cout], w HWIO [K,K,Cin,Cout] -> dx [N,H,W,Cin] (fs_conv2d_dgrad)."""
        self._sync_stream()
        N = int(dy.shape[0])
        K, _, Cin, Cout = (int(s) for s in w.shape)
        H, W = in_hw
        d = L.fs_conv_desc()
        d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = N, H, W, Cin, Cout, K, K, stride
        if isinstance(padding, str):
            d.pad_mode = L.FS_PAD_SAME if padding == "SAME" else L.FS_PAD_VALID
        else:
            d.pad_mode = L.FS_PAD_EXPLICIT
            d.pad_t, d.pad_l, d.Ho, d.Wo = padding
        d.w = self.mem.ptr(w)
        nbytes = self.lib.fs_conv2d_dgrad_workspace_bytes(ctypes.byref(d))
        ws, dx = self.mem.empty((nbytes // 4,)), self.mem.empty((N, H, W, Cin))
        L.check(self.lib, self.lib.fs_conv2d_dgrad(self.ctx, ctypes.byref(d), self.mem.ptr(dy), self.mem.ptr(dx), self.mem.ptr(ws), nbytes), "fs_conv2d_dgrad")
        return dx

    def _resizeconv(self, fn, name, a, b, N, H, W, Cin, Cout, out_shape):
        self._sync_stream()
        nbytes = self.lib.fs_resizeconv_workspace_bytes(N, H, W, Cin, Cout)
        if not nbytes:
            raise L.FaststyleError("%s: Cin and Cout must be multiples of 4 (got %d -> %d)" % (name, Cin, Cout))
        ws, out = self.mem.empty((nbytes // 4,)), self.mem.empty(out_shape)
        p = self.mem.ptr
        L.check(self.lib, fn(self.ctx, p(a), p(b), N, H, W, Cin, Cout, p(out), p(ws), nbytes), name)
        return out

    def resizeconv_fwd(self, x, w):
        """upconv2d's conv (im_transf_net.py:122-155: NEAREST x4 + 3x3 stride-2 SAME), phase-collapsed: x [N,H,W,Cin] -> [N,2H,2W,Cout]."""
        N, H, W, Cin = (int(s) for s in x.shape)
        Cout = int(w.shape[3])
        return self._resizeconv(self.lib.fs_resizeconv_fwd, "fs_resizeconv_fwd", x, w, N, H, W, Cin, Cout, (N, 2 * H, 2 * W, Cout))

    def resizeconv_dgrad(self, dy, w):
        N, H2, W2, Cout = (int(s) for s in dy.shape)
        Cin = int(w.shape[2])
        return self._resizeconv(self.lib.fs_resizeconv_dgrad, "fs_resizeconv_dgrad", dy, w, N, H2 // 2, W2 // 2, Cin, Cout, (N, H2 // 2, W2 // 2, Cin))

    def resizeconv_wgrad(self, x, dy):
        N, H, W, Cin = (int(s) for s in x.shape)
        Cout = int(dy.shape[3])
        return self._resizeconv(self.lib.fs_resizeconv_wgrad, "fs_resizeconv_wgrad", x, dy, N, H, W, Cin, Cout, (3, 3, Cin, Cout))

    def instnorm_apply(self, z, a, b, mode=0, skip=None, skip_a=None, skip_b=None):
        """act(a z + b) materialised (mode 0 none / 1 ReLU / 2 scaled tanh), or the residual block's sum with `skip` [N,H+4,W+4,C] (fs_instnorm_apply)."""
        self._sync_stream()
        N, H, W, C = (int(s) for s in z.shape)
        out = self.mem.empty((N, H, W, C))
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_instnorm_apply(self.ctx, p(z), p(a), p(b), N, H, W, C, int(mode), p(skip), p(skip_a), p(skip_b), p(out)), "fs_instnorm_apply")
        return out

    def resize_bicubic_u8(self, img_u8, out):
        """tf.image.resize_images(img, out.shape[:2], method=2) of TF 1.0 (datapipe.py:24) on the device:
        img_u8 host uint8 [H,W,3] (packed RGB) or [H,W,4] (RGBX as a decoder stores it: datapipe.decode_jpeg; the fourth byte is ignored)
        -> ``out`` (device float32 [Ho,Wo,3], written in place)."""
        self._sync_stream()
        H, W, C = (int(s) for s in img_u8.shape)
        Ho, Wo, Co = (int(s) for s in out.shape)
        assert C in (3, 4) and Co == 3
        src = self.mem.upload_u8(img_u8)
        if C == 3:
            L.check(self.lib, self.lib.fs_resize_bicubic_u8(self.ctx, self.mem.ptr_u8(src), H, W, self.mem.ptr(out), Ho, Wo),
                    "fs_resize_bicubic_u8")
        else:
            L.check(self.lib, self.lib.fs_resize_bicubic_u8x(self.ctx, self.mem.ptr_u8(src), H, W, 4, self.mem.ptr(out), Ho, Wo),
                    "fs_resize_bicubic_u8x")
        return out

    def u8_to_f32(self, src_u8, dst):
        """device u8 buffer -> device float32 (same element order)."""
        self._sync_stream()
        n = int(np.prod(dst.shape))
        L.check(self.lib, self.lib.fs_u8_to_f32(self.ctx, self.mem.ptr_u8(src_u8), n, self.mem.ptr(dst)), "fs_u8_to_f32")
        return dst

    def f32_to_u8(self, src, dst_u8, swap_rb=False):
        """device float32 [...,3] -> device u8 by truncation (numpy astype(uint8)), optional R<->B swap."""
        self._sync_stream()
        npix = int(np.prod(src.shape)) // 3
        L.check(self.lib, self.lib.fs_f32_to_u8(self.ctx, self.mem.ptr(src), npix, 1 if swap_rb else 0,
                                                self.mem.ptr_u8(dst_u8)), "fs_f32_to_u8")
        return dst_u8

    def adam_tf_step(self, p, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
        """tf.train.AdamOptimizer(lr) update (train.py:203), in place; t = 1-based step."""
        self._sync_stream()
        P = self.mem.ptr
        L.check(self.lib, self.lib.fs_adam_tf_step(self.ctx, P(p), P(g), P(m), P(v), int(np.prod(p.shape)), lr, beta1,
                                                   beta2, eps, t), "fs_adam_tf_step")

    # ------------------------------------------------------------------ single ops (tests)
    def conv2d(self, x, w, stride=1, padding="SAME", **kw):
        self._sync_stream()
        d = L.fs_conv_desc()
        N, H, W, Cin = (int(s) for s in x.shape)
        KH, KW, _, Cout = (int(s) for s in w.shape[-4:])
        d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = N, H, W, Cin, Cout, KH, KW, stride
        if isinstance(padding, str):
            d.pad_mode = L.FS_PAD_SAME if padding == "SAME" else L.FS_PAD_VALID
        else:
            d.pad_mode = L.FS_PAD_EXPLICIT
            d.pad_t, d.pad_l, d.Ho, d.Wo = padding
        d.src_mode = kw.get("src_mode", 0)
        d.refl = kw.get("refl", 0)
        p = self.mem.ptr
        d.x, d.w = p(x), p(w)
        for k in ("in_a", "in_b", "bias", "add_src"):
            setattr(d, k, p(kw.get(k)))
        d.in_per_sample = int(kw.get("in_per_sample", 0))
        d.in_relu = int(kw.get("in_relu", 0))
        d.out_relu = int(kw.get("out_relu", 0))
        d.shuffle = int(kw.get("shuffle", 0))
        d.add_pad = int(kw.get("add_pad", 0))
        d.w_nstride = int(kw.get("w_nstride", 0))
        d.mask_src = p(kw.get("mask_src"))
        d.route_src = p(kw.get("route_src"))
        inb = kw.get("inb")      # (z, mean, rstd, a, b, relu): also leave the instance-norm-backward partial sums of the unit that produced z
        if kw.get("winograd") == 6:        # F(4x4,3x3), split-bf16 pipeline (fs_wino6.hip); the output extent is known only after the plan: SAME / explicit pads here
            U = self.mem.empty((self.lib.fs_wino6_filter_bytes(Cin, Cout) // 4,))
            L.check(self.lib, self.lib.fs_wino6_transform_filter(self.ctx, p(w), Cin, Cout, p(U)), "fs_wino6_transform_filter")
            ho, wo = (H, W) if isinstance(padding, str) and padding == "SAME" else ((H - 2, W - 2) if isinstance(padding, str) else (padding[2], padding[3]))
            nb = self.lib.fs_wino6_workspace_bytes(N, ho, wo, Cin, Cout)
            if kw.get("w6_chunk_tiles"):    # a smaller scratch: the launch runs in tile chunks
                nb = 36 * int(kw["w6_chunk_tiles"]) * (Cin + Cout) * 4
            ws6 = self.mem.empty((nb // 4,))
            d.w_wino6, d.w6_ws, d.w6_ws_bytes = p(U), p(ws6), nb
            self._keep = [U, ws6]
        elif kw.get("winograd") == "4t":     # F(4x4,3x3), 16-tile items (fs_wino4t.hip)
            U = self.mem.empty((36, Cin, Cout))
            L.check(self.lib, self.lib.fs_wino4t_transform_filter(self.ctx, p(w), Cin, Cout, p(U)), "fs_wino4t_transform_filter")
            d.w_wino4t = p(U)
            self._keep = [U]
        elif kw.get("winograd") == 4:     # F(4x4,3x3) (fs_wino4.hip)
            U = self.mem.empty((36, Cin, Cout))
            L.check(self.lib, self.lib.fs_wino4_transform_filter(self.ctx, p(w), Cin, Cout, p(U)), "fs_wino4_transform_filter")
            d.w_wino4 = p(U)
            self._keep = [U]
        elif kw.get("winograd"):     # F(2x2,3x3): transform the filter into a caller-owned buffer, hand it over with the conv
            U = self.mem.empty((16, Cin, Cout))
            L.check(self.lib, self.lib.fs_wino_transform_filter(self.ctx, p(w), Cin, Cout, p(U)), "fs_wino_transform_filter")
            d.w_wino = p(U)
            self._keep = [U]
        if inb is not None:      # (set before planning: the launch is then planned with 16 x 16-pixel items whatever the grid)
            d.inb_z, d.inb_mean, d.inb_rstd, d.inb_a, d.inb_b = (p(t) for t in inb[:5])
            d.inb_relu = int(inb[5])
            d.inb_rec = 16       # placeholder until the record count is known
        tiles = ctypes.c_int()
        L.check(self.lib, self.lib.fs_conv2d_plan(ctypes.byref(d), ctypes.byref(tiles)), "fs_conv2d_plan")
        if d.shuffle:
            y = self.mem.empty((N, 2 * d.Ho, 2 * d.Wo, Cout // 4))
        else:
            y = self.mem.empty((N, d.Ho, d.Wo, Cout))
        d.y = p(y)
        pool = None
        if kw.get("want_pool"):
            pool = self.mem.empty((N, d.Ho // 2, d.Wo // 2, Cout))
            d.pool_out = p(pool)
        rec = None
        if inb is not None:
            rec = self.mem.empty((N, tiles.value, Cout, 2))
            d.inb_rec = p(rec)
        stats = None
        if kw.get("want_stats"):
            stats = self.mem.empty((N, tiles.value, Cout, 3))
            d.stats = p(stats)
        L.check(self.lib, self.lib.fs_conv2d_fwd(self.ctx, ctypes.byref(d)), "fs_conv2d_fwd")
        if pool is not None:
            return y, pool
        if rec is not None:
            return y, rec
        return (y, stats, tiles.value) if kw.get("want_stats") else y

    def instnorm_finalize(self, stats, tiles, C, groups, gamma, beta, eps=1e-3):
        self._sync_stream()
        N = int(stats.shape[0])
        out = [self.mem.empty((N, C)) for _ in range(4)]
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_instnorm_finalize(self.ctx, p(stats), N, tiles, C, groups, p(gamma), p(beta), eps,
                                                        *[p(o) for o in out]), "fs_instnorm_finalize")
        return out  # mean, rstd, a, b

    def instnorm_bwd(self, gin, z, mean, rstd, a, b, mode):
        self._sync_stream()
        N, H, W, C = (int(s) for s in z.shape)
        dz = self.mem.empty(z.shape)
        dg, db = self.mem.empty((C,)), self.mem.empty((C,))
        nbytes = self.lib.fs_instnorm_bwd_workspace_bytes(N, H * W, C)
        ws = self.mem.empty((nbytes // 4,))
        p = self.mem.ptr
        L.check(self.lib, self.lib.fs_instnorm_bwd(self.ctx, p(gin), p(z), p(mean), p(rstd), p(a), p(b), mode, N, H * W, C,
                                                   p(dz), p(dg), p(db), p(ws), nbytes), "fs_instnorm_bwd")
        return dz, dg, db

    def conv2d_wgrad(self, x, dy, k, stride=1, padding="SAME", per_sample=False, scale=1.0, **kw):
        self._sync_stream()
        d = L.fs_wgrad_desc()
        N, H, W, Cin = (int(s) for s in x.shape)
        Cout = int(dy.shape[-1])
        d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = N, H, W, Cin, Cout, k, k, stride
        if isinstance(padding, str):
            d.pad_mode = L.FS_PAD_SAME if padding == "SAME" else L.FS_PAD_VALID
        else:
            d.pad_mode = L.FS_PAD_EXPLICIT
            d.pad_t, d.pad_l, d.Ho, d.Wo = padding
        d.src_mode = kw.get("src_mode", 0)
        d.refl = kw.get("refl", 0)
        p = self.mem.ptr
        d.x, d.dy = p(x), p(dy)
        d.in_a, d.in_b = p(kw.get("in_a")), p(kw.get("in_b"))
        d.in_per_sample = int(kw.get("in_per_sample", 0))
        d.in_relu = int(kw.get("in_relu", 0))
        d.per_sample = int(per_sample)
        d.scale = scale
        dw = self.mem.empty((N, Cin, Cout) if per_sample else (k, k, Cin, Cout))
        d.dw = p(dw)
        nbytes = self.lib.fs_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
        ws = self.mem.empty((max(nbytes // 4, 1),))
        L.check(self.lib, self.lib.fs_conv2d_wgrad(self.ctx, ctypes.byref(d), p(ws), nbytes), "fs_conv2d_wgrad")
        return dw
