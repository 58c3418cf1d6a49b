#!/usr/bin/env python
"""Headline benchmark of the faststyle hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a torchrun environment: bench.py starts the N ranks itself -- it re-runs the same command as
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...`
  (one process per GPU, RCCL); launched that way by the caller it just joins.  `--launch` does the same for --gpus 1
  (RCCL all-reduce at world size 1); `--rendezvous-only` is a preflight of the launcher alone.  Rank 0 prints the ONE JSON
  line; `value` is the whole-job rate (max-over-ranks time), `per_rank_images_per_sec` every rank's own.

A "step" is one full train.py loop body (reference train.py:245-275) on a synthetic 256x256 batch: content-target VGG
pass, transform-net forward, VGG16 + Gram + losses, full backward, ONE RCCL all-reduce (SUM) of the 424,102 gradients
when launched under torch.distributed, TF-Adam.  fp32 end to end (the reference's dtype); every convolution / Gram
contraction runs on the fp32 matrix cores.  Forward + backward replay one hipGraph per step.

  value                the step at batch 32 PER GPU -- BASELINE.json's metric configuration ("256x256 b32") on one GPU;
                       weak scaling: global batch 32 N
  train_b4_per_gpu     the same step at batch 4 per GPU (BASELINE configs[2]; at N=8 this is configs[3], global batch 32),
                       also hipGraph-replayed, with its own per-kernel table
  roofline             dominant kernel of the b32 step: FLOPs EXECUTED / HIP-event time on the launch stream (an eager
                       pass of the same step right after the timed region -- a replayed hipGraph has no room for events
                       between its nodes), per-kernel table, HBM bytes per launch from profiles/ (rocprofv3 PMC passes)
  gram                 Gram forward (F^T F, utils.py:76-82) + backward GFLOP/s and % of the fp32 MFMA peak
  vgg_gram_substep     fs_perceptual_loss as a whole (VGG16 forward incl. the content-target half, Grams, losses, VGG
                       input gradients): wall time of the section by HIP events, FLOPs executed and as written
  cpu_baseline         BASELINE.md section 3: the numpy oracle (as-written algorithm) and a torch-CPU (oneDNN) restatement
                       on this box's host cores, warm-up + >= 5 timed runs, median/min; a 1-core figure; the 720p forward
  stylize_*            configs[1] (720p, batch 1, fp32) and configs[4] (1080p, batch 8 per GPU, bf16) forward rates
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per image at 256x256 (FLOP = 2*MAC, convs / Grams only), AS WRITTEN
GF_TNET_FWD, GF_TNET_BWD, GF_VGG_CONTENT, GF_VGG_FWD, GF_VGG_DGRAD, GF_GRAM = 7.069, 13.260, 24.386, 36.465, 36.465, 4.295
GF_STEP = GF_TNET_FWD + GF_TNET_BWD + GF_VGG_CONTENT + GF_VGG_FWD + GF_VGG_DGRAD + GF_GRAM          # 121.94
GF_VGG_GRAM = GF_VGG_CONTENT + GF_VGG_FWD + GF_VGG_DGRAD + GF_GRAM                                   # 101.61
# transform-net forward: executed (phase-collapsed resize-conv) GFLOP and minimum fp32 HBM traffic (MB) per image
FWD_WORK = {(720, 1280): (70.756, 83.496, 1524.6), (1080, 1920): (155.023, 183.688, 3335.5), (256, 256): (6.163, 7.069, 134.4)}
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: bf16 MFMA, dense (no sparsity)
PEAK_HBM_TBS = 8.0                    # spec; 6.29 measured (float4 copy)


def vgg_algorithmic_bytes(B, S=256, cmax=6, lmax=9, layers=None):
    """Algorithmic HBM bytes of the F(4x4) VGG16 launches of one train step (conv1_2 ... conv4_3 forward on [y ; content] up to conv3_3 and on y
    beyond, their nine input gradients): every input, ReLU-mask source and output ONCE, the content half's full-resolution conv1_2 / conv2_2 outputs
    not at all (nothing reads them), filters not counted (0.24 GB, L2 / Infinity-Cache resident).  Returns (forward reads, forward writes,
    input-gradient reads incl. masks, input-gradient writes) in bytes -- 2.72 + 2.82 + 2.89 + 1.44 = 9.87 GB at batch 32 over the 18 launches (DESIGN.md
    section 4); `layers`: restrict to those conv layers (1 = conv1_2 ... 9 = conv4_3), e.g. range(1, 7) = the 12 launches that stay on the fp32 kernel
    when conv4_x runs on the split-bf16 pipeline."""
    cin = [3, 64, 64, 128, 128, 256, 256, 256, 512, 512]
    cout = [64, 64, 128, 128, 256, 256, 256, 512, 512, 512]
    pool_after = lambda l: l in (1, 3, 6)
    h, hl = S, []
    for l in range(lmax + 1):
        hl.append(h)
        if pool_after(l) and l < lmax:
            h //= 2
    fr = fw = dr = dw = 0
    for l in (range(1, lmax + 1) if layers is None else layers):
        nb, px = (2 * B if l <= cmax else B), hl[l] * hl[l]
        fr += nb * px * cin[l] * 4
        if pool_after(l) and l < lmax:
            fw += (B if l < cmax else nb) * px * cout[l] * 4 + nb * (px // 4) * cout[l] * 4
        else:
            fw += nb * px * cout[l] * 4
        dr += B * px * cout[l] * 4 + (0 if pool_after(l - 1) else B * px * cin[l] * 4)
        dw += B * px * cin[l] * 4
    return fr, fw, dr, dw


def newest_profile(stem):
    """profiles/rNN_<stem>: the newest round's file (the rocprofv3 PMC summaries bench.py quotes HBM traffic from)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + stem)))
    return c[-1] if c else None


def same_build(profile_json):
    """A stored counter profile describes the running library only if it was collected from the same kernel sources
    (tools/pmc_traffic.py stores faststyle_amd.build.source_digest()); otherwise its numbers are not quoted."""
    try:
        from faststyle_amd import build as fsbuild
        return profile_json.get("csrc_sha16") is not None and profile_json.get("csrc_sha16") == fsbuild.source_digest()
    except Exception:
        return False


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--b4-steps", type=int, default=200, help="timed steps of the batch-4-per-GPU leg")
    ap.add_argument("--profile-steps", type=int, default=5, help="eager steps of the per-kernel / per-section pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-quick", action="store_true", help="one warm-up + one timed run per CPU figure (contract test)")
    ap.add_argument("--no-stylize", action="store_true")
    ap.add_argument("--no-b4", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--launch", action="store_true",
                    help="start the ranks through torch.distributed.run even for --gpus 1 (RCCL all-reduce at world size 1); "
                         "--gpus N > 1 without a torchrun environment always does")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher preflight: every rank joins the process group (RCCL, or gloo when no GPU is visible), SUM-"
                         "all-reduces its rank, rank 0 prints one JSON line; no engine, no kernels")
    return ap.parse_args()


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_cmd(n, argv, port):
    """The driver's command form for N ranks on one node (one process per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--launch"]


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: re-run this command under torch.distributed.run with N
    ranks; stdout (rank 0's ONE JSON line) and the return code pass through."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = launch_cmd(args.gpus, sys.argv[1:], free_port())
    print("bench: launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def rendezvous_only(args):
    import torch
    import torch.distributed as dist
    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    gpu = torch.cuda.is_available()
    if gpu:
        torch.cuda.set_device(local)
    saved_fd = os.dup(1)       # (RCCL's banner must not land on stdout)
    os.dup2(2, 1)
    try:
        dist.init_process_group("nccl" if gpu else "gloo", **({"device_id": torch.device("cuda", local)} if gpu else {}))
        t = torch.tensor([float(rank)], device="cuda" if gpu else "cpu")
        dist.all_reduce(t)
        if gpu:
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    assert float(t.item()) == world * (world - 1) / 2, (float(t.item()), world)
    if rank == 0:
        print(json.dumps({"rendezvous_ok": True, "n_gpus": world, "backend": "RCCL" if gpu else "gloo", "allreduce_sum_of_ranks": float(t.item())}))
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------- CPU baseline
_QUICK = False


def _timed(fn, warm, n, what=""):
    if _QUICK:
        warm, n = min(warm, 1), 1
    t_start = time.perf_counter()
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    print("bench: cpu_baseline %s: %d+%d runs in %.1f s" % (what, warm, n, time.perf_counter() - t_start), file=sys.stderr, flush=True)
    return statistics.median(ts), min(ts)


def cpu_baseline(size):
    """BASELINE.md section 3.  TF1 cannot be installed here or on the GPU box: the figures are the build's own CPU
    restatements of the reference path (kind "port"), timed on the host cores of this box."""
    import torch
    from threadpoolctl import threadpool_limits
    from oracle import perceptual, tnet, torch_ref           # allowed here: the CPU-baseline leg only
    ncpu = os.cpu_count() or 1
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = ncpu
    rng = np.random.default_rng(1)
    P = tnet.init_params(0)
    Wv = perceptual.synthetic_vgg_weights(3)
    style = np.random.default_rng(2).uniform(0, 255, (1, 128, 128, 3)).astype(np.float32)
    layers = ("conv1_2", "conv2_2", "conv3_3", "conv4_3")
    tg = perceptual.target_grams(style, Wv, layers)
    x1 = rng.uniform(0, 255, (1, size, size, 3)).astype(np.float32)
    x4 = rng.uniform(0, 255, (4, size, size, 3)).astype(np.float32)
    x720 = np.random.default_rng(0).uniform(0, 255, (1, 720, 1280, 3)).astype(np.float32)

    # (i) numpy/OpenBLAS oracle, the as-written algorithm (materialised x4 upsample, unfused instance norm), all cores
    med, mn = _timed(lambda: perceptual.train_step(P, x1, tg, Wv), 2, 5, "numpy train step")
    out = {"value": round(1.0 / med, 4), "unit": "images/sec", "cores": int(ncores), "kind": "port",
           "sample": "train step (fwd+bwd, no Adam) of 1x%dx%dx3, numpy/OpenBLAS float32 oracle of the reference path "
                     "(as-written algorithm); 2 warm-up + 5 timed runs, median; host has %d logical CPUs" % (size, size, ncpu),
           "min_time_value": round(1.0 / mn, 4)}

    # (ii) torch-CPU (oneDNN) float32 + autograd: the "strong CPU" figure, batch 4 as train.py's default
    Pt = dict((k, torch.tensor(v, requires_grad=True)) for k, v in P.items())
    Wt = dict((k, torch.tensor(v)) for k, v in Wv.items())
    tgt = [torch.tensor(g) for g in tg]
    # torch's CPU convolutions stop scaling early on this class of host (measured on the 256-thread GPU box: batch-4 step
    # 1.05 s with 16 threads, 1.43 s with 32, 2.1 s with 64, 6.1 s with 128, > 40 s with 256): use the fastest setting
    tthreads = min(ncores, 16)
    torch.set_num_threads(tthreads)
    xt4, xt1, xt720 = torch.tensor(x4), torch.tensor(x1), torch.tensor(x720)
    med, mn = _timed(lambda: torch_ref.train_step(Pt, xt4, tgt, Wt), 3, 5, "torch train step b4")
    out["torch_cpu"] = {"value": round(4.0 / med, 3), "min_time_value": round(4.0 / mn, 3), "unit": "images/sec", "cores": int(tthreads),
                        "sample": "train step of 4x%dx%dx3, torch %s CPU (oneDNN) float32 autograd restatement, %d threads (more "
                                  "threads are slower on this host); 3 warm-up + 5 timed, median" % (size, size, torch.__version__, tthreads)}
    # (iii) one core, for per-core normalisation
    torch.set_num_threads(1)
    with threadpool_limits(limits=1):
        med, mn = _timed(lambda: torch_ref.train_step(Pt, xt1, tgt, Wt), 1, 3, "torch train step 1 core")
    out["one_core"] = {"value": round(1.0 / med, 4), "min_time_value": round(1.0 / mn, 4), "unit": "images/sec", "cores": 1,
                       "sample": "train step of 1x%dx%dx3, torch CPU float32, 1 thread; 1 warm-up + 3 timed, median" % (size, size)}
    torch.set_num_threads(tthreads)
    # (iv) 720p forward (configs[1])
    with torch.no_grad():
        Pn = dict((k, v.detach()) for k, v in Pt.items())
        med, mn = _timed(lambda: torch_ref.tnet(xt720, Pn), 3, 5, "torch 720p forward")
    s720 = {"torch_cpu_fps": round(1.0 / med, 3), "torch_cpu_min_time_fps": round(1.0 / mn, 3), "torch_cpu_cores": int(tthreads),
            "numpy_oracle_cores": int(ncores),
            "sample": "create_net forward of 1x720x1280x3; torch CPU float32 3 warm-up + 5 timed; numpy oracle 1 + 3; median"}
    med, mn = _timed(lambda: tnet.create_net(x720, P), 1, 3, "numpy 720p forward")
    s720["numpy_oracle_fps"] = round(1.0 / med, 3)
    s720["numpy_oracle_min_time_fps"] = round(1.0 / mn, 3)
    out["stylize_720p"] = s720
    return out


# ----------------------------------------------------------------------------------------------- helpers
def fam_table(prof, steps, names):
    """fs_profile_end output -> {kernel: {launches_per_step, gflop_per_step, tflops, ms_per_step, frac}}"""
    tab = {}
    for f, nm in enumerate(names):
        n, fl, ms = prof[3 * f], prof[3 * f + 1], prof[3 * f + 2]
        if ms > 0 and nm:
            tf = fl / (ms * 1e-3) / 1e12
            tab[nm] = {"launches_per_step": round(n / steps, 1), "gflop_per_step": round(fl / steps / 1e9, 2),
                       "ms_per_step": round(ms / steps, 3), "avg_launch_us": round(1e3 * ms / n, 1),
                       "tflops": round(tf, 2), "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}
            if nm.startswith("wino6"):   # the split-bf16 pipeline: its FLOP count is the fp32-equivalent F(4x4) products; six bf16 products are issued for each
                tab[nm]["frac_note"] = "fp32-EQUIVALENT products executed / the fp32 MFMA peak (157.3); the launch issues 6 bf16 products per fp32 one"
                tab[nm]["bf16_tflops_issued"] = round(6 * tf, 1)
                tab[nm]["frac_of_bf16_mfma_peak_issued"] = round(6 * tf / PEAK_BF16_MFMA_TFLOPS, 4)
            if nm in ("conv_s16_kernel", "conv_stream_kernel", "gram_stream_kernel"):   # round 6: the launches of these rows run as split-bf16 kernels unless FS_S16_SPLIT / FS_CSTREAM_SPLIT / FS_GRAM_SPLIT = 0
                split = os.environ.get({"conv_s16_kernel": "FS_S16_SPLIT", "conv_stream_kernel": "FS_CSTREAM_SPLIT"}.get(nm, "FS_GRAM_SPLIT"), "1") != "0"
                if split:
                    tab[nm]["frac_note"] = ("fp32-EQUIVALENT FLOPs / the fp32 MFMA peak (157.3): the launches of this row issue their products as six exact bf16 pieces "
                                            "on the bf16 matrix cores (conv_s16x / conv_s16c3x / conv_stream X6 / conv_r64x / gram_streamx; VGG16 conv1_1 in the conv_s16 row and the 64-channel Gram stay fp32)")
    return tab


def main():
    global _QUICK
    args = parse()
    _QUICK = args.cpu_quick
    if "RANK" not in os.environ and (args.gpus > 1 or args.launch):
        raise SystemExit(self_launch(args))
    if args.rendezvous_only:
        return rendezvous_only(args)
    import torch
    import torch.distributed as dist
    from faststyle_amd import _lib, engine, im_transf_net, trainer, utils, vgg16

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ       # under torch.distributed.run (any world size)
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d inside a %d-rank torch.distributed environment" % (args.gpus, world))
    # FS_DIST_BACKEND=gloo + FS_DIST_SHARE_GPU=1 (tests only): N ranks time-share the ONE GPU of a test box -- RCCL refuses two
    # ranks on one device, gloo stages the 1.7 MB gradient through the host.  The line then says so in config.collective.
    backend = os.environ.get("FS_DIST_BACKEND", "nccl")
    share_gpu = os.environ.get("FS_DIST_SHARE_GPU") == "1"
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on the process's stdout when the communicator comes up; stdout carries exactly ONE
        # JSON line here, so file descriptor 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
            warm = torch.zeros(1, device="cuda")
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    ddist = dist if launched else None
    eng = engine.Engine(engine.TorchMem("cuda:%d" % local))
    lib = eng.lib
    NF = _lib.FS_PROFILE_FAMILIES
    names = _lib.profile_family_names(lib)        # one row per kernel symbol
    # row indices by NAME (the library's own table, fs_profile_family_name), not by a hand-kept copy of its enum
    fam = dict((n, i) for i, n in enumerate(names))
    F_WINO, F_WINO2_VGG, F_WINO2_TNET = fam["wino_conv_kernel"], fam["wino2_conv_kernel (VGG16 convs)"], fam["wino2_conv_kernel (transform-net residual convs)"]
    F_WINO4 = fam["wino4_conv_kernel"]
    F_WINO4T_TNET, F_WINO4T_VGG = fam["wino4t_conv_kernel (transform-net residual convs)"], fam["wino4t_conv_kernel (VGG16 convs)"]
    WINO_F4 = (F_WINO4, F_WINO4T_TNET, F_WINO4T_VGG)
    F_GRAM_FWD = (fam["gram_stream_kernel"], fam["conv_wgrad_kernel (Gram forward)"])
    F_GRAM_BWD = (fam["gram_bwd_kernel"], fam["conv_igemm_kernel (Gram backward, 1x1 per-sample filters)"])
    WINO_F2 = (F_WINO, F_WINO2_VGG, F_WINO2_TNET)
    F_WINO6 = next((f for f, n in enumerate(names) if n.startswith("wino6")), None)

    B, S = args.batch_per_gpu, args.size
    params = eng.flatten_params(im_transf_net.initial_variables(seed=0), scope="")
    npz = os.path.join(ROOT, "libs", "vgg16_weights.npz")
    real_vgg = os.path.exists(npz)
    vgg_w = vgg16.load_weights(npz) if real_vgg else vgg16.synthetic_weights(seed=3)
    style = utils.imread(os.path.join(ROOT, "style_images", "starry_night_crop.jpg")).astype(np.float32)[None]

    g = torch.Generator(device="cuda")
    g.manual_seed(100 + rank)

    def sync():
        torch.cuda.synchronize()
        if launched:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(v):
        if world > 1:
            t = torch.tensor([v], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return v

    def all_ranks(v):
        """[v of rank 0, v of rank 1, ...] on every rank"""
        if world > 1:
            t = torch.zeros(world, device="cuda", dtype=torch.float64)
            t[rank] = v
            dist.all_reduce(t)
            return [float(x) for x in t.tolist()]
        return [float(v)]

    def train_leg(batch, steps, warmup, tr):
        """Timed region + the eager per-section / per-kernel passes of one batch size."""
        pool = [torch.rand((batch, S, S, 3), device="cuda", generator=g) * 255.0 for _ in range(4)]
        n_warm = max(warmup, 2 if tr.use_graph else 1)             # (graph mode: first step captures, second replays)
        for i in range(n_warm):
            tr.step(pool[i % len(pool)])
        sync()
        graphed = bool(tr.use_graph and tr.graph is not None)
        t0 = time.perf_counter()
        for i in range(steps):
            losses = tr.step(pool[(n_warm + i) % len(pool)])
        sync()
        mine = time.perf_counter() - t0
        elapsed = max_over_ranks(mine)
        res = {"elapsed": elapsed, "graphed": graphed, "loss": float(losses[0].item()),
               "per_rank_images_per_sec": [round(steps * batch / t, 2) for t in all_ranks(mine)]}
        # ---- where the all-reduce + Adam (outside the hipGraph) and the host leave the GPU idle: 20 more steps of the timed
        # kind with an event before and after each; gap = end of step i -> start of step i+1 on the device time line
        Gs = 20
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(Gs)]
        tr.ar_events = []
        for i in range(Gs):
            evs[i][0].record()
            tr.step(pool[i % len(pool)])
            evs[i][1].record()
        sync()
        res["allreduce_us"] = float(np.median([a.elapsed_time(b) for a, b in tr.ar_events])) * 1e3 if tr.ar_events else None
        tr.ar_events = None
        busy = [evs[i][0].elapsed_time(evs[i][1]) for i in range(Gs)]
        gaps = [evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(Gs - 1)]
        res["step_device_ms"] = float(np.median(busy))
        res["inter_step_gap_us"] = float(np.median(gaps)) * 1e3
        # ---- eager passes of the same step: (1) section wall times by HIP events on the launch stream, no per-kernel events
        e, P = tr.eng, args.profile_steps
        secs = ["tnet_forward", "perceptual_loss", "tnet_backward", "allreduce_adam"]
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(P)]

        def sections(i, hook_begin=None, hook_end=None):
            x = pool[i % len(pool)]
            out = []
            for k in range(4):
                if hook_begin:
                    hook_begin()
                else:
                    ev[i][k].record()
                if k == 0:
                    y = e.tnet_forward(tr.params, x, save_for_bwd=True)
                elif k == 1:
                    _, dy = e.perceptual_loss(y, x, tr.target_grams, tr.cfg)
                elif k == 2:
                    e.tnet_backward(tr.params, x, dy, grads=tr.grads)
                else:
                    if tr._dist_on():
                        tr.dist.all_reduce(tr.grads, op=tr.dist.ReduceOp.SUM)
                    tr.global_step += 1
                    e.adam_tf_step(tr.params, tr.grads, tr.m, tr.v, tr.global_step, lr=tr.lr)
                if hook_end:
                    out.append(hook_end())
            if not hook_begin:
                ev[i][4].record()
            return out
        for i in range(P):
            sections(i)
        sync()
        res["sections_ms"] = dict((secs[k], round(sum(ev[i][k].elapsed_time(ev[i][k + 1]) for i in range(P)) / P, 3)) for k in range(4))
        # (2) per-kernel: HIP events around every MFMA launch, collected per section
        acc = np.zeros((4, 3 * NF))

        def hb():
            lib.fs_profile_begin(e.ctx)

        def he():
            buf = (ctypes.c_double * (3 * NF))()
            lib.fs_profile_end(e.ctx, ctypes.byref(buf))
            return np.array(buf[:])
        for i in range(P):
            for k, v in enumerate(sections(i, hb, he)):
                acc[k] += v
        sync()
        res["prof"] = acc
        res["psteps"] = P
        del pool
        return res

    # ------------------------------------------------------------------ the metric's configuration: batch 32 per GPU
    tr = trainer.Trainer(eng, params, vgg_w, style, learn_rate=1e-3, dist=ddist, use_graph=not args.no_graph)
    main_leg = train_leg(B, args.steps, args.warmup, tr)
    b4_leg = None
    if not args.no_b4 and B != 4:
        tr4 = trainer.Trainer(eng, params, None, style, learn_rate=1e-3, dist=ddist, use_graph=not args.no_graph)
        b4_leg = train_leg(4, args.b4_steps, args.warmup, tr4)
        del tr4
    del tr

    # ------------------------------------------------------------------ configs[1] / configs[4]: forward only, independent frames
    fwd = {}
    if not args.no_stylize:
        from faststyle_amd import ckpt
        Wc = ckpt.load_checkpoint(os.path.join(ROOT, "models", "starry_final.ckpt"))
        flat = eng.mem.from_numpy(eng.flatten_params(Wc))

        def fwd_leg(shape, bf16, warm, iters, graph):
            x = torch.rand(shape, device="cuda", generator=g) * 255.0
            # inference: the checkpoint does not change between frames (frozen) -- in a workspace of this leg's own, as the streaming driver
            # does: the captured graph skips the filter re-layouts, so nobody else may write where they live
            wsp = eng.new_tnet_workspace(shape[0], shape[1], shape[2], bf16)
            run = lambda: eng.tnet_forward(flat, x, bf16=bf16, frozen=True, workspace=wsp)
            for _ in range(warm):
                run()
            sync()
            if graph and not args.no_graph:
                # one hipGraph per frame, as the streaming driver does (faststyle_amd/stream.py): a batch-1 frame is ~45
                # short launches and the host's launch rate should not be part of the number
                try:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        run()
                    torch.cuda.current_stream().wait_stream(side)
                    sync()
                    fg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(fg, capture_error_mode="thread_local"):
                        run()
                    run = fg.replay
                    run()
                    sync()
                except Exception as ex:
                    print("bench: frame graph capture failed (%s); eager launches" % ex, file=sys.stderr)
            t0 = time.perf_counter()
            for _ in range(iters):
                run()
            sync()
            dt = max_over_ranks(time.perf_counter() - t0)
            del wsp
            eng.invalidate_frozen()
            n = shape[0]
            fps = world * iters * n / dt
            gf_direct, gf_written, mb = FWD_WORK[(shape[1], shape[2])]
            if bf16:
                mb = mb / 2
            # FLOPs EXECUTED: on the fp32 path the ten residual convs run as Winograd F(4x4,3x3) (fs_wino4t.hip; every shape timed here has
            # >= 64 items) -- 36 products per 4x4 outputs instead of 144.  (Until round 4 this figure counted them in direct form.)
            gf_exec = gf_direct
            if not bf16:
                hr, wr = (shape[1] + 80) // 4, (shape[2] + 80) // 4            # residual input extent (im_transf_net.py:34-45)
                for k in range(1, 11):
                    ho, wo = hr - 2 * k, wr - 2 * k
                    gf_exec += (2.0 * -(-ho // 4) * -(-wo // 4) * 36 - 2.0 * ho * wo * 9) * 64 * 64 / 1e9
            per_gpu = fps / world
            rep = {"fps": round(fps, 1), "ms_per_batch": round(1e3 * dt / iters, 3), "batch_per_gpu": n, "iters": iters,
                   "tflops_executed": round(gf_exec * per_gpu / 1e3, 2), "tflops_direct_form": round(gf_direct * per_gpu / 1e3, 2),
                   "tflops_as_written": round(gf_written * per_gpu / 1e3, 2),
                   "min_traffic_TBps": round(mb * per_gpu / 1e6, 3), "frac_hbm_peak_min_traffic": round(mb * per_gpu / 1e6 / PEAK_HBM_TBS, 4),
                   "hip_graph": bool(graph and not args.no_graph)}
            if bf16:     # two roofs: the bf16 matrix cores (2.5 PFLOP/s dense) and HBM -- this leg is built around bytes
                rep["frac_bf16_mfma_peak_executed"] = round(gf_exec * per_gpu / 1e3 / PEAK_BF16_MFMA_TFLOPS, 4)
            else:
                rep["frac_f32_mfma_peak_executed"] = round(gf_exec * per_gpu / 1e3 / PEAK_F32_MFMA_TFLOPS, 4)
            # HBM bytes actually moved per batch, from the rocprofv3 PMC passes of the same shape (tools/collect_profiles.sh)
            tp = newest_profile("hbm_traffic_%dp_%s.json" % (shape[1], "b%d_bf16" % n if bf16 else "fp32")) if (n == 1 or bf16) else None
            if tp:
                tj = json.load(open(tp))
                passes = tj.get("forward_passes", 23)
                if tj.get("kernels") and not same_build(tj):
                    rep["hbm_counter_source"] = os.path.relpath(tp, ROOT) + " (NOT quoted: collected from other kernel sources than the running build)"
                elif tj.get("kernels"):
                    tot_b = sum(k["traffic_bytes_per_launch"] * k["launches_sampled"] for k in tj["kernels"].values()) / passes
                    rep["hbm_counter_bytes_per_batch"] = int(tot_b)
                    rep["hbm_counter_TBps"] = round(tot_b / (dt / iters) / 1e12, 3)
                    rep["frac_hbm_peak_counters"] = round(tot_b / (dt / iters) / 1e12 / PEAK_HBM_TBS, 4)
                    rep["hbm_counter_source"] = os.path.relpath(tp, ROOT)
            return rep
        fwd["stylize_720p"] = fwd_leg((1, 720, 1280, 3), False, 10, 50, True)

        def two_in_flight(shape, warm, iters, bf16=False, depth=int(os.environ.get("FS_BENCH_FRAMES_IN_FLIGHT", "2"))):
            """INFORMATIONAL (round 6; not the metric's number): the same batch-1 frame graph twice -- two frames, two workspaces, two streams -- replayed
            alternately, as a video / webcam pipeline that accepts one frame of latency would run it: the ~45 dependent launches of a frame leave the chip
            idle between them (a 720p frame's 28 statistics / residual-add launches are 5-8 us each for microseconds of work), a second frame fills the gaps."""
            try:
                streams = [torch.cuda.Stream() for _ in range(depth)]
                graphs, keep = [], []
                for st in streams:
                    x = torch.rand(shape, device="cuda", generator=g) * 255.0
                    wsp = eng.new_tnet_workspace(shape[0], shape[1], shape[2], bf16)
                    run = lambda x=x, wsp=wsp: eng.tnet_forward(flat, x, bf16=bf16, frozen=True, workspace=wsp)
                    st.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(st):
                        for _ in range(warm):
                            run()
                    sync()
                    with torch.cuda.stream(st):
                        fg = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(fg, stream=st, capture_error_mode="thread_local"):
                            run()
                    sync()
                    graphs.append(fg)
                    keep.append((x, wsp))
                for _ in range(3):
                    for st, fg in zip(streams, graphs):
                        with torch.cuda.stream(st):
                            fg.replay()
                sync()
                t0 = time.perf_counter()
                for _ in range(iters):
                    for st, fg in zip(streams, graphs):
                        with torch.cuda.stream(st):
                            fg.replay()
                sync()
                dt = time.perf_counter() - t0
                del graphs, keep
                eng.invalidate_frozen()
                return {"fps": round(depth * iters * shape[0] / dt, 1), "frames_in_flight": depth,
                        "note": "informational: two frame-batch graphs on two streams, alternated; the leg's own fps (one batch at a time) is the configured figure"}
            except Exception as ex:
                return {"error": "%s: %s" % (type(ex).__name__, ex)}
        if world == 1 and not args.no_graph:
            fwd["stylize_720p"]["two_frames_in_flight"] = two_in_flight((1, 720, 1280, 3), 3, 50)
        fwd["stylize_1080p_b8_bf16"] = fwd_leg((8, 1080, 1920, 3), True, 3, 20, False)
        if world == 1 and not args.no_graph:
            fwd["stylize_1080p_b8_bf16"]["two_batches_in_flight"] = two_in_flight((8, 1080, 1920, 3), 2, 12, bf16=True)
        fwd["stylize_1080p_b8_fp32"] = fwd_leg((8, 1080, 1920, 3), False, 2, 8, False)

    if rank == 0:
        def leg_report(leg, batch, steps):
            n_img = steps * batch * world
            value = n_img / leg["elapsed"]
            P = leg["psteps"]
            acc = leg["prof"]
            tot = acc.sum(axis=0)
            per_kernel = fam_table(tot, P, names)
            fams = [(tot[3 * f + 2], f) for f in range(NF)]
            di = max(fams)[1]                                              # dominant = most GPU time
            exec_gflop = sum(tot[3 * f + 1] for f in range(NF)) / P / 1e9     # FLOPs executed per step (all MFMA kernels)
            mfma_ms = sum(tot[3 * f + 2] for f in range(NF)) / P
            step_s = leg["elapsed"] / steps
            pl = acc[1]
            perc_gflop = sum(pl[3 * f + 1] for f in range(NF)) / P / 1e9
            perc_ms = leg["sections_ms"]["perceptual_loss"]
            gsum = lambda fams, k: sum(pl[3 * f + k] for f in fams)
            gram_fl = (gsum(F_GRAM_FWD, 1) + gsum(F_GRAM_BWD, 1)) / P
            gram_ms = (gsum(F_GRAM_FWD, 2) + gsum(F_GRAM_BWD, 2)) / P
            tfl = lambda fams: round(gsum(fams, 1) / (gsum(fams, 2) * 1e-3) / 1e12, 2) if gsum(fams, 2) else None
            rep = {
                "images_per_sec": round(value, 2), "ms_per_step": round(1e3 * step_s, 3), "steps": steps, "batch_per_gpu": batch,
                "global_batch": batch * world, "hip_graph": leg["graphed"], "final_loss": leg["loss"],
                "per_rank_images_per_sec": leg["per_rank_images_per_sec"],
                "sections_ms_eager": leg["sections_ms"],
                # the step as the device sees it (events around graph replay + all-reduce + Adam) and the idle time between two
                # steps: what keeping the all-reduce and the optimiser outside the captured graph costs
                "step_device_ms": round(leg["step_device_ms"], 3), "inter_step_gap_us": round(leg["inter_step_gap_us"], 1),
                # HIP events around the one all-reduce(SUM) of the flat gradient buffer (None without a process group: a plain `python bench.py`)
                "allreduce_us": round(leg["allreduce_us"], 1) if leg.get("allreduce_us") is not None else None,
                "host_gap_frac_of_step": round(leg["inter_step_gap_us"] * 1e-3 / (leg["elapsed"] / steps * 1e3), 5),
                "step_tflops_as_written": round(GF_STEP * value / 1e3, 2),
                "step_frac_as_written": round(GF_STEP * value / 1e3 / world / PEAK_F32_MFMA_TFLOPS, 4),
                "step_gflop_executed": round(exec_gflop, 1),
                "step_tflops_executed": round(exec_gflop / step_s / 1e3, 2),
                "step_frac_executed": round(exec_gflop / step_s / 1e3 / PEAK_F32_MFMA_TFLOPS, 4),
                "all_mfma_kernels_tflops": round(exec_gflop / mfma_ms, 2) if mfma_ms else None,
                "mfma_kernel_ms_per_step": round(mfma_ms, 3),
                "gram": {"gflop_per_step": round(gram_fl / 1e9, 2), "ms_per_step": round(gram_ms, 3),
                         "tflops": round(gram_fl / (gram_ms * 1e-3) / 1e12, 2) if gram_ms else None,
                         "frac_of_f32_mfma_peak": round(gram_fl / (gram_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if gram_ms else None,
                         "forward_tflops": tfl(F_GRAM_FWD), "backward_tflops": tfl(F_GRAM_BWD),
                         "gflop_as_written_per_step": round(4.295 * batch, 2),
                         "tflops_as_written": round(4.295 * batch / gram_ms, 2) if gram_ms else None,
                         "note": "G = F^T F/(hwc) per sample on 4 layers (utils.py:66-83) and dF = F (dG+dG^T); 4.295 GFLOP/img as written. "
                                 "gflop_per_step / tflops / frac count the FLOPs EXECUTED: the forward kernel multiplies 10 of the 16 "
                                 "32x32 blocks of a diagonal 128x128 tile (the result is symmetric), everything else in full"},
                "vgg_gram_substep": {"ms": perc_ms, "gflop_executed": round(perc_gflop, 1),
                                     "tflops_executed": round(perc_gflop / perc_ms, 2),
                                     "frac_executed": round(perc_gflop / perc_ms / PEAK_F32_MFMA_TFLOPS, 4),
                                     "gflop_as_written": round(GF_VGG_GRAM * batch, 1),
                                     "frac_as_written": round(GF_VGG_GRAM * batch / perc_ms / PEAK_F32_MFMA_TFLOPS, 4),
                                     "note": "fs_perceptual_loss: VGG16 fwd on [y;content], 4 Grams, losses, VGG input gradients; wall "
                                             "time of the section by HIP events in an eager pass (every kernel of it, not only MFMA ones)"},
                "per_kernel": per_kernel,
            }
            # all 3x3 VGG16 conv launches of the step (conv1_2 .. conv4_3 forward + input gradients), whichever kernels they took: the fp32 F(4x4) kernel and,
            # since round 6, the split-bf16 pipeline for conv4_x -- FLOPs = fp32-equivalent products executed
            vf = [f for f in (F_WINO4T_VGG, F_WINO6, F_WINO4, F_WINO, F_WINO2_VGG) if f is not None and tot[3 * f + 2] > 0]
            v_fl, v_ms, v_n = (sum(tot[3 * f + k] for f in vf) for k in (1, 2, 0))
            if v_ms:
                rep["vgg16_3x3_convs"] = {"kernels": [names[f] for f in vf], "launches_per_step": round(v_n / P, 1), "gflop_per_step": round(v_fl / P / 1e9, 1),
                                          "ms_per_step": round(v_ms / P, 3), "tflops": round(v_fl / (v_ms * 1e-3) / 1e12, 2),
                                          "frac": round(v_fl / (v_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
            wf = [F_WINO4T_VGG, F_WINO4T_TNET, F_WINO4, F_WINO, F_WINO2_VGG, F_WINO2_TNET]
            w_fl, w_ms, w_n = (sum(tot[3 * f + k] for f in wf) for k in (1, 2, 0))
            if w_ms:
                rep["winograd_family"] = {"kernels": [names[f] for f in wf if tot[3 * f + 2] > 0], "launches_per_step": round(w_n / P, 1),
                                          "gflop_per_step": round(w_fl / P / 1e9, 1), "ms_per_step": round(w_ms / P, 3),
                                          "tflops": round(w_fl / (w_ms * 1e-3) / 1e12, 2),
                                          "frac": round(w_fl / (w_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
            return rep, di, per_kernel
        rep, di, per_kernel = leg_report(main_leg, B, args.steps)
        dom = per_kernel[names[di]]
        traffic, traffic_src = None, None
        sym = names[di].split(" (")[0]                     # the kernel symbol of the dominant row
        fam_key = None
        if sym == "wino4t_conv_kernel":                    # (its big-item VGG16 instances <2>, <3> and the transform-net instances <1>, <4> are separate rows)
            fam_key = "wino4t_conv_kernel (VGG16 big items)" if di == F_WINO4T_VGG else "wino4t_conv_kernel (transform net)"
            sym = fam_key
        tpath = newest_profile("hbm_traffic_pmc.json")
        alg_launch = None
        if di == F_WINO4T_VGG and S == 256:
            # (18 launches: all of conv1_2 .. conv4_3; 12: conv4_x -- layers 7..9 -- runs on the split-bf16 pipeline and the row holds conv1_2 .. conv3_3)
            on_row = None if dom["launches_per_step"] > 17.5 else range(1, 7)
            alg_launch = int(sum(vgg_algorithmic_bytes(B, layers=on_row)) / max(dom["launches_per_step"], 1))
        if tpath:
            tj = json.load(open(tpath))
            k = tj.get("families", {}).get(fam_key) if fam_key else tj.get("kernels", {}).get(sym)
            if k and tj.get("batch_per_gpu") == B and same_build(tj):
                traffic, traffic_src = k["traffic_bytes_per_launch"], os.path.relpath(tpath, ROOT)
            elif k and tj.get("batch_per_gpu") == B:
                traffic_src = os.path.relpath(tpath, ROOT) + " (NOT quoted: collected from other kernel sources than the running build)"
        out = {
            "metric": "images/sec train-step 256x256 b32 (+ Gram GFLOPs % MFMA peak); 720p stylize fps",
            "value": rep["images_per_sec"], "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": rep["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "per_rank_images_per_sec": rep["per_rank_images_per_sec"],
            "data": "synthetic (uniform[0,255) images; %s VGG16 weights; random-init transform net)"
                    % ("real" if real_vgg else "synthetic He-normal"),
            "config": {"workload": "train.py step: %dx%d, batch %d per GPU (BASELINE metric 'b32' on one GPU; global batch %d), VGG16 "
                                   "conv1_2/2_2/3_3/4_3 Gram style loss + conv3_3 content loss, resize-conv transform net, TF-Adam, "
                                   "hipGraph-replayed; secondary line train_b4_per_gpu = the same step at batch 4 per GPU "
                                   "(BASELINE configs[2]; configs[3] at N=8)" % (S, S, B, B * world),
                       "global_batch": B * world, "batch_per_gpu": B, "image_size": [S, S], "parallelism": "dp%d" % world,
                       "collective": ("one %s all-reduce(SUM) of 1,696,408 B per step (world size %d)%s" % (
                           "RCCL" if backend == "nccl" else backend, world,
                           "; TEST MODE: the ranks time-share one GPU" if share_gpu else "")) if launched else "none (single process)",
                       "style_image": "starry_night_crop.jpg 640x938"},
            "roofline": {"bound": "mfma",
                         "kernel": names[di] + (": fp32 MFMA, Winograd F(4x4,3x3) 3x3 conv; achieved = FLOPs EXECUTED (36 products per "
                                                "4x4 outputs instead of 144: a quarter of the direct form's, 0.5625 of F(2x2,3x3)'s)" if di in WINO_F4 else
                                                ": fp32 MFMA, Winograd F(2x2,3x3) 3x3 conv; achieved = FLOPs EXECUTED (16 products per "
                                                "2x2 outputs instead of 36)" if di in WINO_F2 else ": fp32 MFMA"),
                         "achieved": dom["tflops"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac"],
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch of the kernel symbol %s (2*FETCH_SIZE+WRITE_SIZE, KiB "
                                                             "counters, separate rocprofv3 --pmc passes)" % sym,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_launch,
                         "traffic_over_algorithmic": round(traffic / alg_launch, 3) if (traffic and alg_launch) else None,
                         "algorithmic_bytes_note": "VGG16: every input, mask source and output of the row's launches once (bench.py vgg_algorithmic_bytes: "
                                                   "9.87 GB per batch-32 step over all 18 launches, 8.99 GB over the 12 of conv1_2 .. conv3_3), / launches of the row" if alg_launch else None,
                         "timed_with": "HIP events on the launch stream, eager pass of %d steps right after the timed region; every row "
                                       "of per_kernel is ONE kernel symbol (template instances summed), so row ms = launches x the "
                                       "average duration of that symbol in profiles/*kernel_stats*" % args.profile_steps,
                         "direct_form_equivalent_tflops": round(dom["tflops"] * (4.0 if di in WINO_F4 else 2.25), 2) if di in WINO_F2 + WINO_F4 else None,
                         "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_launch_us"],
                         "winograd_family": rep.get("winograd_family"),
                         "vgg16_3x3_convs": rep.get("vgg16_3x3_convs"),
                         "per_kernel": per_kernel},
            "gram": rep["gram"], "vgg_gram_substep": rep["vgg_gram_substep"],
            "step_tflops_as_written": rep["step_tflops_as_written"], "step_frac_of_f32_mfma_peak": rep["step_frac_as_written"],
            "step_gflop_executed": rep["step_gflop_executed"], "step_tflops_executed": rep["step_tflops_executed"],
            "step_frac_executed": rep["step_frac_executed"], "all_mfma_kernels_tflops": rep["all_mfma_kernels_tflops"],
            "mfma_kernel_ms_per_step": rep["mfma_kernel_ms_per_step"], "sections_ms_eager": rep["sections_ms_eager"],
            "final_loss": rep["final_loss"], "hip_graph": rep["hip_graph"],
            "step_device_ms": rep["step_device_ms"], "inter_step_gap_us": rep["inter_step_gap_us"], "allreduce_us": rep["allreduce_us"],
            "host_gap_frac_of_step": rep["host_gap_frac_of_step"],
        }
        if b4_leg is not None:
            r4, d4, pk4 = leg_report(b4_leg, 4, args.b4_steps)
            r4["dominant_kernel"] = names[d4]
            r4["config"] = ("BASELINE configs[2] (single GPU, batch 4)" if world == 1 else
                            "BASELINE configs[3] shape: %d ranks x batch 4, global batch %d, one RCCL all-reduce(SUM) of 1,696,408 B "
                            "per step" % (world, 4 * world))
            out["train_b4_per_gpu"] = r4
        for k, v in fwd.items():
            out[k] = v
        if fwd:
            out["stylize_720p_fps"] = fwd["stylize_720p"]["fps"]
            out["stylize_1080p_b8_bf16_fps"] = fwd["stylize_1080p_b8_bf16"]["fps"]     # FS_FLAG_BF16: ~52 dB PSNR vs the fp32 path
            out["stylize_1080p_b8_fp32_fps"] = fwd["stylize_1080p_b8_fp32"]["fps"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S)
        # the headline figures once more as the LAST key of the line (and as the last stderr line): what a log tail shows
        dg = {"b32_images_per_sec": out["value"], "b32_ms_per_step": out["ms_per_step"], "n_gpus": world,
              "roofline_frac": out["roofline"]["frac"], "roofline_kernel": names[di].split(" (")[0],
              "vgg_gram_frac_executed": out["vgg_gram_substep"]["frac_executed"], "step_frac_executed": out["step_frac_executed"]}
        if b4_leg is not None:
            dg["b4_per_gpu_images_per_sec"] = out["train_b4_per_gpu"]["images_per_sec"]
            dg["b4_ms_per_step"] = out["train_b4_per_gpu"]["ms_per_step"]
        if fwd:
            dg["stylize_720p_fps"] = out["stylize_720p_fps"]
            if "fps" in fwd["stylize_720p"].get("two_frames_in_flight", {}):
                dg["stylize_720p_two_frames_in_flight_fps"] = fwd["stylize_720p"]["two_frames_in_flight"]["fps"]   # (informational: two streams, stream.PipelinedStylizer)
            if "fps" in fwd["stylize_1080p_b8_bf16"].get("two_batches_in_flight", {}):
                dg["stylize_1080p_b8_bf16_two_batches_in_flight_fps"] = fwd["stylize_1080p_b8_bf16"]["two_batches_in_flight"]["fps"]
            dg["stylize_1080p_b8_bf16_fps"] = out["stylize_1080p_b8_bf16_fps"]
            dg["stylize_1080p_b8_fp32_fps"] = out["stylize_1080p_b8_fp32_fps"]
        if "cpu_baseline" in out:
            dg["cpu_baseline_images_per_sec"] = out["cpu_baseline"].get("value")
        out["digest"] = dg
        print(json.dumps(out))
        sys.stdout.flush()
        print("bench digest: " + json.dumps(dg), file=sys.stderr)
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
