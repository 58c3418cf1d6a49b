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
        eng.vgg_load(vgg_weights)
        # train.py:144-151: target Grams of the style image, computed once
        self.target_grams = eng.style_targets(mem.from_numpy(style_img), self.cfg)
        self.use_graph = use_graph
        self.graph = None
        self._static_in = None
        self._static_losses = None

    def _world(self):
        d = self.dist
        return d.get_world_size() if (d is not None and d.is_initialized()) else 1

    def _forward_backward(self, batch):
        e = self.eng
        y = e.tnet_forward(self.params, batch, save_for_bwd=True, upsample_method=self.method)
        losses, dy = e.perceptual_loss(y, batch, self.target_grams, self.cfg)
        e.tnet_backward(self.params, batch, dy, grads=self.grads, upsample_method=self.method)
        return losses

    def _capture(self, batch):
        """Capture forward+backward for this batch shape into a hipGraph (after eager warm-up so
        every one-time initialisation inside the library has already happened)."""
        import torch
        self._static_in = batch.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._forward_backward(self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: calls made by OTHER threads (e.g. the RCCL watchdog of a data-parallel run) must not
        # invalidate this thread's capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._static_losses = self._forward_backward(self._static_in)
        self.graph = g

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
                    self.graph = None
        if self.use_graph and self.graph is not None:
            self._static_in.copy_(batch)
            self.graph.replay()
            losses = self._static_losses
        else:
            losses = self._forward_backward(batch)
        if self._world() > 1:
            self.dist.all_reduce(self.grads, op=self.dist.ReduceOp.SUM)   # 1,696,408 B, once per step
        self.global_step += 1
        e.adam_tf_step(self.params, self.grads, self.m, self.v, self.global_step, lr=self.lr)
        return losses

    def params_numpy(self):
        return self.eng.mem.to_numpy(self.params).copy()
