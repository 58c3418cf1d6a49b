"""Frame-streaming stylizer: the loop body of the reference's stylize_webcam.py (:84-96) on the HIP
engine -- u8 frame in, u8 frame out, everything in between on the GPU.

Per frame: upload u8 (3 B/pixel over PCIe instead of 12) -> fs_u8_to_f32 -> fs_tnet_forward ->
fs_f32_to_u8 (``astype(np.uint8)`` truncation + the BGR<->RGB swap of cv2.cvtColor) -> download u8.
The device part is captured ONCE into a hipGraph (fixed frame size, fixed weights) and replayed per
frame: a batch-1 frame is ~45 short launches, so replay removes the per-launch host latency from
the frame time.  Reference quirk kept: the webcam frame is BGR but is fed to the RGB-trained net
as is, and the *output* channels are swapped before display (stylize_webcam.py:88-95).
"""
import numpy as np

from . import _lib as L


class FrameStylizer(object):
    KEEP_GRAPH = False     # tests: keep the captured hipGraph_t so that its node types can be inspected

    def __init__(self, eng, variables, height, width, upsample_method="resize", batch=1, swap_rb=True, use_graph=True,
                 bf16=False):
        self.eng = eng
        self.variables = variables
        self.method = upsample_method
        self.bf16 = bf16                 # FS_FLAG_BF16 mixed-precision path (~2x the frame rate, ~52 dB vs fp32)
        self.shape = (int(batch), int(height), int(width), 3)
        self.swap_rb = swap_rb
        mem = eng.mem
        Ho, Wo = eng.tnet_out_shape(height, width)
        self.out_shape = (int(batch), Ho, Wo, 3)
        self._in_u8 = mem.upload_u8(np.zeros(self.shape, np.uint8))
        self._in_f32 = mem.empty(self.shape)
        self._out_u8 = mem.upload_u8(np.zeros(self.out_shape, np.uint8))
        self._graph = None
        # A workspace of this stylizer's OWN: with frozen=True the re-laid-out filters of self.variables live inside it and the captured
        # graph no longer rebuilds them -- nobody else may write there (the engine's shared per-shape workspace would be rewritten by any
        # other same-shape forward, and every later replay would silently mix two models).
        self._ws = eng.new_tnet_workspace(self.shape[0], self.shape[1], self.shape[2], bf16)
        self._use_graph = use_graph and hasattr(mem, "torch")
        self._y = None

    def _device_pass(self):
        e = self.eng
        e.u8_to_f32(self._in_u8, self._in_f32)
        self._y = e.tnet_forward(self.variables, self._in_f32, upsample_method=self.method, bf16=self.bf16, frozen=True,   # one checkpoint, many frames
                                 workspace=self._ws)
        e.f32_to_u8(self._y, self._out_u8, swap_rb=self.swap_rb)

    def _capture(self):
        torch = self.eng.mem.torch
        self.eng.invalidate_frozen()                       # the warm-up below rebuilds the filters in self._ws whatever the library remembers
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up outside capture (one-time initialisation)
            self._device_pass()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(keep_graph=True) if self.KEEP_GRAPH else torch.cuda.CUDAGraph()
        # thread_local: calls made by OTHER threads (e.g. the RCCL watchdog of a data-parallel run) must not
        # invalidate this thread's capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._device_pass()
        self._graph = g                                    # (replays raw pointers into self._ws, which lives as long as this object)

    def release(self):
        """Drop the captured graph."""
        self._graph = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def __call__(self, frames_u8):
        """frames_u8: host uint8 [H,W,3] (or [B,H,W,3]) -> host uint8 stylized frame(s) of the net's output size."""
        a = np.asarray(frames_u8)
        single = a.ndim == 3
        if single:
            a = a[np.newaxis]
        if a.shape != self.shape or a.dtype != np.uint8:
            raise L.FaststyleError("frame shape %s dtype %s, stylizer was built for uint8 %s" % (a.shape, a.dtype, self.shape))
        mem = self.eng.mem
        if hasattr(mem, "torch"):
            self._in_u8.copy_(mem.torch.from_numpy(np.ascontiguousarray(a)), non_blocking=True)
        else:
            self._in_u8[...] = a
        if self._use_graph:
            if self._graph is None:
                self._capture()
            self._graph.replay()
        else:
            self._device_pass()
        out = np.array(mem.to_numpy(self._out_u8), copy=True)     # (synchronises; the device buffer is reused next frame)
        return out[0] if single else out


class PipelinedStylizer(object):
    """`depth` FrameStylizer lanes on streams of their own (round 6): frame i + 1 is uploaded and stylized while frame i is still on the device.

    A batch-1 frame is ~45 DEPENDENT launches, 28 of them statistics / residual-add kernels of a few microseconds each: one frame at a time leaves the chip
    idle between them.  Independent frames need no collective and no shared state (BASELINE: "inference shards independent frames") -- two frame graphs on two
    streams fill each other's gaps: 720p 1277 -> 1890 frames/s on one MI355X (bench.py: stylize_720p.two_frames_in_flight), for one frame of latency.  Every
    lane owns its buffers, its workspace (the re-laid-out filters live inside it) and its captured graph; results are bit-identical to FrameStylizer's.

        ps = PipelinedStylizer(eng, variables, H, W)            # same arguments as FrameStylizer, + depth
        for out in ps.run(frames):                              # any iterable of uint8 [H,W,3] frames, results in order
            ...
    or submit(frame) / fetch() by hand (at most `depth` frames between them)."""

    def __init__(self, eng, variables, height, width, depth=2, **kw):
        if not hasattr(eng.mem, "torch"):
            raise L.FaststyleError("PipelinedStylizer needs the GPU engine (streams); use FrameStylizer on the emulator")
        import collections
        torch = eng.mem.torch
        self.torch = torch
        self.depth = int(depth)
        self.lanes = [FrameStylizer(eng, variables, height, width, **kw) for _ in range(self.depth)]
        self.streams = [torch.cuda.Stream() for _ in self.lanes]
        self.events = [torch.cuda.Event() for _ in self.lanes]
        self.host_in = [torch.empty(ln.shape, dtype=torch.uint8, pin_memory=True) for ln in self.lanes]
        self.host_out = [torch.empty(ln.out_shape, dtype=torch.uint8, pin_memory=True) for ln in self.lanes]
        self._pending = collections.deque()
        self._n = 0
        self._single = collections.deque()

    def submit(self, frames_u8):
        if len(self._pending) >= self.depth:
            raise L.FaststyleError("PipelinedStylizer: %d frames in flight already -- fetch() one first" % self.depth)
        a = np.asarray(frames_u8)
        single = a.ndim == 3
        if single:
            a = a[np.newaxis]
        k = self._n % self.depth
        ln, st, torch = self.lanes[k], self.streams[k], self.torch
        if a.shape != ln.shape or a.dtype != np.uint8:
            raise L.FaststyleError("frame shape %s dtype %s, stylizer was built for uint8 %s" % (a.shape, a.dtype, ln.shape))
        self.host_in[k].copy_(torch.from_numpy(np.ascontiguousarray(a)))       # (host -> pinned host; the lane's previous frame was fetched: its buffers are free)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            ln._in_u8.copy_(self.host_in[k], non_blocking=True)
            if ln._use_graph:
                if ln._graph is None:
                    ln._capture()
                ln._graph.replay()
            else:
                ln._device_pass()
            self.host_out[k].copy_(ln._out_u8, non_blocking=True)
            self.events[k].record(st)
        self._pending.append(k)
        self._single.append(single)
        self._n += 1

    def fetch(self):
        """The oldest submitted frame's result (host uint8), waiting for its lane only."""
        k = self._pending.popleft()
        single = self._single.popleft()
        self.events[k].synchronize()
        out = self.host_out[k].numpy().copy()
        return out[0] if single else out

    def run(self, frames):
        for f in frames:
            if len(self._pending) >= self.depth:
                yield self.fetch()
            self.submit(f)
        while self._pending:
            yield self.fetch()

    def release(self):
        for ln in self.lanes:
            ln.release()
