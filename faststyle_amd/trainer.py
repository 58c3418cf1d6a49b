"""The train.py loop body (reference train.py:245-275) on the HIP engine, data-parallel ready.

One step = content-target pass + transform-net forward + VGG/Gram/loss forward + full backward
(all inside libfaststyle_hip.so) + ONE gradient all-reduce (SUM, RCCL over xGMI through
torch.distributed) + TF-style Adam.  Losses are batch-SUMMED in the reference (losses.py:32,63)
and instance norm is per sample, so SUM-reducing per-rank gradients of local batches reproduces
the single-process gradient of the global batch exactly (up to fp32 summation order).

The ~170 launches of the forward+backward are launch-latency sensitive (many are < 10 us), so they
can be captured ONCE into a hipGraph (``use_graph=True``) and replayed per step; the all-reduce
and the Adam kernel stay outside the graph (the step count changes the Adam scalars every step).
"""
import numpy as np

from . import engine as _engine


class Trainer(object):
    KEEP_GRAPH = False     # tests: keep the captured hipGraph_t (CUDAGraph.raw_cuda_graph()) so that its node types can be inspected

    def __init__(self, eng, params_flat, vgg_weights, style_img, cfg=None, learn_rate=1e-3, dist=None,
                 use_graph=False, upsample_method="resize"):
        """params_flat: np.float32 [424102] (ckpt order); style_img: np [1,Hs,Ws,3] RGB 0..255;
        dist: None or an initialised torch.distributed module (backend nccl == RCCL)."""
        self.eng = eng
        self.method = upsample_method
        self.cfg = cfg or _engine.default_loss_cfg()
        self.lr = learn_rate
        self.dist = dist
        mem = eng.mem
        self.params = mem.from_numpy(np.asarray(params_flat, np.float32))
        if self._world() > 1:
            dist.broadcast(self.params, src=0)            # identical init on every rank
        self.grads = mem.zeros(self.params.shape)
        self.m = mem.zeros(self.params.shape)
        self.v = mem.zeros(self.params.shape)
        self.global_step = 0
        if vgg_weights is not None:          # None: the engine already holds this weight set (a second Trainer on it)
            eng.vgg_load(vgg_weights)
        # train.py:144-151: target Grams of the style image, computed once
        self.target_grams = eng.style_targets(mem.from_numpy(style_img), self.cfg)
        self.use_graph = bool(use_graph) and getattr(mem, "supports_graphs", False)
        self.graph = None
        self._graph_keepalive = []
        self._static_in = None
        self._static_losses = None
        self.ar_events = None        # bench.py: a list -> step() appends an (event before, event after) pair around every all-reduce

    def _dist_on(self):
        d = self.dist
        return d is not None and d.is_initialized()

    def _world(self):
        return self.dist.get_world_size() if self._dist_on() else 1

    def _forward_backward(self, batch):
        e = self.eng
        # the transform net writes y where fs_perceptual_loss stages it (inside the perceptual workspace): no device copy of y; the captured step
        # also keeps its input batch there (self._static_in, see _capture): no copy of the content half either
        N, H, W, _ = (int(v) for v in batch.shape)
        Ho, Wo = e.tnet_out_shape(H, W)
        y_view = e.perceptual_inputs(N, Ho, Wo, self.cfg)[0] if (Ho, Wo) == (H, W) and hasattr(e.mem, "view") else None
        y = e.tnet_forward(self.params, batch, save_for_bwd=True, upsample_method=self.method, out=y_view)
        losses, dy = e.perceptual_loss(y, batch, self.target_grams, self.cfg)
        e.tnet_backward(self.params, batch, dy, grads=self.grads, upsample_method=self.method)
        return losses

    def _capture(self, batch):
        """Capture forward+backward for this batch shape into a hipGraph (after eager warm-up so
        every one-time initialisation inside the library has already happened)."""
        import torch
        self._release_graph()
        N, H, W, _ = (int(v) for v in batch.shape)
        content_view = perc_ws = None
        if self.eng.tnet_out_shape(H, W) == (H, W) and hasattr(self.eng.mem, "view"):
            _, content_view, perc_ws = self.eng.perceptual_inputs(N, H, W, self.cfg, with_ws=True)
        if content_view is not None:          # the step's input lives where the content half of the VGG batch is staged:
            self.eng.pin_workspaces([perc_ws], True)    # pin exactly the tensor self._static_in views, before anything can evict it
            self._graph_keepalive = [perc_ws]
            self._static_in = content_view
            self._static_in.copy_(batch)
        else:
            perc_ws = None
            self._static_in = batch.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._forward_backward(self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(keep_graph=True) if self.KEEP_GRAPH else torch.cuda.CUDAGraph()
        # thread_local: calls made by OTHER threads (e.g. the RCCL watchdog of a data-parallel run) must not
        # invalidate this thread's capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._static_losses = self._forward_backward(self._static_in)
        self.graph = g
        # the graph replays raw pointers into the two workspaces ITS forward/backward used (the most recently used entry of
        # each cache): keep exactly those alive and un-evictable for as long as the graph lives
        self._graph_keepalive = self._graph_keepalive + self.eng.pin_last_used(tnet=True, perceptual=perc_ws is None)

    def _release_graph(self):
        self.eng.release_pins(self._graph_keepalive)
        self._graph_keepalive = []
        self.graph = None

    def __del__(self):
        try:
            self._release_graph()
        except Exception:
            pass

    def step(self, batch):
        """batch: device tensor [B,H,W,3] float32 RGB 0..255 (train.py:158-160).
        Returns the device tensor {loss, content, style, beta*tv} of the LOCAL batch."""
        e = self.eng
        if self.use_graph:
            if self.graph is None or tuple(self._static_in.shape) != tuple(batch.shape):
                try:
                    self._capture(batch)
                except Exception as ex:                     # fall back loudly, once
                    import sys
                    print("faststyle: hipGraph capture failed (%s); running eagerly" % ex, file=sys.stderr)
                    self.use_graph = False
                    self._release_graph()
        if self.use_graph and self.graph is not None:
            self._static_in.copy_(batch)
            self.graph.replay()
            losses = self._static_losses
        else:
            losses = self._forward_backward(batch)
        if self._dist_on():       # also at world_size 1 (torchrun --nproc-per-node 1): the RCCL path is then exercised as is
            if self.ar_events is not None:
                import torch
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            self.dist.all_reduce(self.grads, op=self.dist.ReduceOp.SUM)   # 1,696,408 B, once per step
            if self.ar_events is not None:
                ev[1].record()
                self.ar_events.append(ev)
        self.global_step += 1
        e.adam_tf_step(self.params, self.grads, self.m, self.v, self.global_step, lr=self.lr)
        return losses

    def params_numpy(self):
        return self.eng.mem.to_numpy(self.params).copy()

    def state_tensors(self, full=True):
        """Checkpoint contents by TF variable name.  full=False: the 48 img_t_net tensors (what
        stylize_image.py restores); full=True: what tf.train.Saver() of train.py:224 writes --
        plus every variable's Adam slots ``<name>/Adam`` (m), ``<name>/Adam_1`` (v) and the int64
        ``global_step``."""
        e = self.eng
        tensors = e.unflatten_params(self.params_numpy(), upsample_method=self.method)
        if full:
            for k, val in e.unflatten_params(e.mem.to_numpy(self.m), upsample_method=self.method).items():
                tensors[k + "/Adam"] = val
            for k, val in e.unflatten_params(e.mem.to_numpy(self.v), upsample_method=self.method).items():
                tensors[k + "/Adam_1"] = val
            tensors["global_step"] = np.array(self.global_step, dtype=np.int64)
        return tensors

    def load_state(self, tensors):
        """Inverse of state_tensors: continue a run from a ``training/<name>.ckpt-<step>`` bundle
        (the reference writes these but has no way to read them back, train.py:256-259).  A bundle
        without Adam slots (a ``*_final.ckpt``) restores the weights only and keeps m = v = 0, step 0."""
        e = self.eng
        mem = e.mem
        names = [k for k in tensors if not k.endswith("/Adam") and not k.endswith("/Adam_1") and k != "global_step"]
        weights = dict((k, tensors[k]) for k in names)
        scope = "img_t_net/" if any(k.startswith("img_t_net/") for k in names) else ""
        flat = e.flatten_params(weights, scope=scope, upsample_method=self.method)
        self.params = mem.from_numpy(flat)
        has_slots = all((k + "/Adam") in tensors and (k + "/Adam_1") in tensors for k in names)
        if has_slots:
            self.m = mem.from_numpy(e.flatten_params(dict((k, tensors[k + "/Adam"]) for k in names), scope=scope,
                                                     upsample_method=self.method))
            self.v = mem.from_numpy(e.flatten_params(dict((k, tensors[k + "/Adam_1"]) for k in names), scope=scope,
                                                     upsample_method=self.method))
            self.global_step = int(np.asarray(tensors.get("global_step", 0)).reshape(-1)[0])
        else:
            self.m = mem.zeros(self.params.shape)
            self.v = mem.zeros(self.params.shape)
            self.global_step = 0
        self._release_graph()                  # a captured graph holds the old buffers
        if self._world() > 1:
            for t in (self.params, self.m, self.v):
                self.dist.broadcast(t, src=0)
        return self.global_step
