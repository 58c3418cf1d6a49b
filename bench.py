#!/usr/bin/env python
"""Headline benchmark of the faststyle hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one full train.py loop body (reference train.py:245-275) on a synthetic 256x256 batch
of 4 images per GPU (the reference's --batch_size 4; global batch 4N, 32 at N=8 = BASELINE.json's
"256x256 b32"): content-target VGG pass, transform-net forward, VGG16+Gram+losses, full backward,
one RCCL all-reduce (SUM) of the 424,102 gradients, TF-Adam.  fp32 end to end (the reference's
dtype); every convolution / Gram contraction runs on the fp32 matrix cores.

Prints ONE JSON line on rank 0 (contract: see the task statement); extra keys:
  roofline      dominant kernel (conv_igemm<32,2,2>, the VGG16 / residual 3x3 convs): algorithmic
                FLOP / HIP-event time, measured live on the launch stream during the timed steps
  cpu_baseline  the numpy oracle (a port of the reference path; TF1 itself cannot be installed)
                timed on this box's host cores on a bounded sample, rank 0 / N=1 only
  train_b32_one_gpu_images_per_sec  the same train step at batch 32 on ONE GPU (N=1 runs only; eager launches)
  stylize_720p_fps  config[1]: im_transf_net forward on a 720p frame, batch 1, fp32
  stylize_1080p_b8_bf16_fps / _fp32_fps  config[4]: 1080p, batch 8 per GPU, the bf16 mixed-precision path
                (bf16 MFMA, fp32 statistics; NOT the parity path) beside the fp32 path on the same input
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md §8d algorithmic work (FLOP = 2*MAC, convs/Grams only), per image at 256x256
GFLOP_PER_IMG_AS_WRITTEN = 121.94
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=4)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stylize", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    return ap.parse_args()


def cpu_baseline(n_images, size):
    """Oracle (numpy float32 port of the reference path, as-written algorithm) train-step rate."""
    from oracle import perceptual, tnet            # allowed here: the CPU-baseline leg only
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    rng = np.random.default_rng(1)
    P = tnet.init_params(0)
    Wv = perceptual.synthetic_vgg_weights(3)
    style = rng.uniform(0, 255, (1, 128, 128, 3)).astype(np.float32)
    tg = perceptual.target_grams(style, Wv, ("conv1_2", "conv2_2", "conv3_3", "conv4_3"))
    t0 = time.perf_counter()
    for _ in range(n_images):
        x = rng.uniform(0, 255, (1, size, size, 3)).astype(np.float32)
        perceptual.train_step(P, x, tg, Wv)
    dt = time.perf_counter() - t0
    return {"value": round(n_images / dt, 4), "unit": "images/sec", "cores": int(threads), "kind": "port",
            "sample": "%d train steps of 1x%dx%dx3 (fwd+bwd, no Adam), numpy/OpenBLAS float32 oracle of the "
                      "reference path; host has %d logical cores" % (n_images, size, size, os.cpu_count() or 0)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from faststyle_amd import engine, im_transf_net, trainer, utils, vgg16

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d"
                             % (args.gpus, args.gpus))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    eng = engine.Engine(engine.TorchMem("cuda:%d" % local))

    B, S = args.batch_per_gpu, args.size
    params = eng.flatten_params(im_transf_net.initial_variables(seed=0), scope="")
    npz = os.path.join(ROOT, "libs", "vgg16_weights.npz")
    real_vgg = os.path.exists(npz)
    vgg_w = vgg16.load_weights(npz) if real_vgg else vgg16.synthetic_weights(seed=3)
    style = utils.imread(os.path.join(ROOT, "style_images", "starry_night_crop.jpg")).astype(np.float32)[None]
    tr = trainer.Trainer(eng, params, vgg_w, style, learn_rate=1e-3, dist=dist if world > 1 else None,
                         use_graph=not args.no_graph)

    # synthetic COCO: uniform [0,255) float32 batches, pre-generated on the device, fresh per step
    g = torch.Generator(device="cuda")
    g.manual_seed(100 + rank)
    pool = [torch.rand((B, S, S, 3), device="cuda", generator=g) * 255.0 for _ in range(8)]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    import ctypes
    for i in range(args.warmup):
        tr.step(pool[i % len(pool)])
    sync()
    graphed = bool(tr.use_graph and tr.graph is not None)
    if not graphed:
        eng.lib.fs_profile_begin(eng.ctx)      # HIP events around every MFMA launch of the timed steps
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = tr.step(pool[(args.warmup + i) % len(pool)])
    sync()
    elapsed = time.perf_counter() - t0
    prof = (ctypes.c_double * 21)()
    loss_val = float(losses[0].item())
    if graphed:
        # the timed steps replayed a hipGraph (no room for events between its nodes): time the SAME
        # kernels with HIP events on the launch stream in an eager pass of the same K steps
        tr.use_graph = False
        eng.lib.fs_profile_begin(eng.ctx)
        for i in range(args.steps):
            tr.step(pool[(args.warmup + i) % len(pool)])
        sync()
        tr.use_graph = True
    eng.lib.fs_profile_end(eng.ctx, ctypes.byref(prof))
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the metric's "b32" on ONE GPU (N=1 only; at N=8 the timed region above already is global batch 32)
    b32 = None
    if world == 1 and not args.no_stylize and B != 32:
        big = [torch.rand((32, S, S, 3), device="cuda", generator=g) * 255.0 for _ in range(2)]
        tr.use_graph = False
        for i in range(2):
            tr.step(big[i % 2])
        sync()
        tb = time.perf_counter()
        for i in range(6):
            tr.step(big[i % 2])
        sync()
        b32 = 6 * 32 / (time.perf_counter() - tb)
        tr.use_graph = not args.no_graph
        del big

    # config[1]: 720p stylize, batch 1 per GPU, independent frames (no collective)
    fps = None
    if not args.no_stylize:
        from faststyle_amd import ckpt
        W = ckpt.load_checkpoint(os.path.join(ROOT, "models", "starry_final.ckpt"))
        flat = eng.mem.from_numpy(eng.flatten_params(W))
        frame = torch.rand((1, 720, 1280, 3), device="cuda", generator=g) * 255.0
        for _ in range(3):
            eng.tnet_forward(flat, frame)
        sync()
        # one hipGraph per frame, as the streaming driver does (faststyle_amd/stream.py): a batch-1 frame is ~45 short
        # launches and the host's launch rate should not be part of the number
        run_frame = lambda: eng.tnet_forward(flat, frame)
        if not args.no_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    eng.tnet_forward(flat, frame)
                torch.cuda.current_stream().wait_stream(side)
                sync()
                fg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(fg, capture_error_mode="thread_local"):
                    eng.tnet_forward(flat, frame)
                run_frame = fg.replay
            except Exception as ex:
                print("bench: frame graph capture failed (%s); eager launches" % ex, file=sys.stderr)
        run_frame()
        sync()
        t1 = time.perf_counter()
        iters = 20
        for _ in range(iters):
            run_frame()
        sync()
        dt = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        fps = world * iters / dt

        # config[4]: bf16 mixed-precision 1080p inference, batch 8 per GPU, independent frames (no collective)
        def timed_fwd(x, bf16, iters):
            for _ in range(2):
                eng.tnet_forward(flat, x, bf16=bf16)
            sync()
            t0 = time.perf_counter()
            for _ in range(iters):
                eng.tnet_forward(flat, x, bf16=bf16)
            sync()
            d = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([d], device="cuda", dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                d = float(tt.item())
            return world * iters * x.shape[0] / d
        hd = torch.rand((8, 1080, 1920, 3), device="cuda", generator=g) * 255.0
        fps_1080_bf16 = timed_fwd(hd, True, 10)
        fps_1080_f32 = timed_fwd(hd, False, 5)
        del hd

    if rank == 0:
        n_img = args.steps * B * world
        value = n_img / elapsed
        fam = [[prof[f * 3 + k] for k in range(3)] for f in range(7)]
        names = ["conv_igemm_kernel<32,2,2>", "conv_igemm_kernel<32,2,1>", "conv_igemm_kernel<16,4,1>",
                 "conv_wgrad_kernel", "conv_igemm_kernel<32,1,2>", "conv_igemm_kernel<32,1,1>", "wino_conv_kernel"]
        di = max(range(7), key=lambda f: fam[f][2])       # dominant = most GPU time in the timed region
        dom = fam[di]
        achieved = dom[1] / (dom[2] * 1e-3) / 1e12 if dom[2] > 0 else 0.0
        # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process, so the
        # figure comes from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")
        if os.path.exists(tpath):
            k = json.load(open(tpath)).get("kernels", {}).get(names[di])
            if k:
                traffic, traffic_src = k["traffic_bytes_per_launch"], "profiles/r01_hbm_traffic_pmc.json"
        mfma_flops = sum(f[1] for f in fam)
        mfma_ms = sum(f[2] for f in fam)
        out = {
            "metric": "images/sec train-step 256x256 (global batch 4/GPU x N; b32 at N=8)",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (uniform[0,255) images; %s VGG16 weights; random-init transform net)"
                    % ("real" if real_vgg else "synthetic He-normal"),
            "config": {"workload": "train.py step: 256x256, batch %d/GPU, VGG16 conv1_2/2_2/3_3/4_3 Gram style loss + "
                                   "conv3_3 content loss, resize-conv transform net, TF-Adam" % B,
                       "global_batch": B * world, "image_size": [S, S], "parallelism": "dp%d" % world,
                       "style_image": "starry_night_crop.jpg 640x938"},
            "roofline": {"bound": "mfma",
                         "kernel": names[di] + (" (fp32 MFMA, Winograd F(2x2,3x3) 3x3 conv; achieved = FLOPs EXECUTED, "
                                                "16 products per 2x2 outputs instead of 36)" if di == 6
                                                else " (fp32 MFMA implicit-GEMM conv)"),
                         "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (2*FETCH_SIZE+WRITE_SIZE, KiB counters)", "traffic_source": traffic_src,
                         "timed_with": "HIP events on the launch stream, " + ("eager pass of the same K steps right after "
                                       "the hipGraph-replayed timed region" if graphed else "inside the timed region"),
                         "direct_form_equivalent_tflops": round(achieved * 2.25, 2) if di == 6 else None,
                         "launches_per_step": round(dom[0] / args.steps, 1),
                         "avg_launch_us": round(1e3 * dom[2] / dom[0], 2) if dom[0] else None,
                         "all_mfma_kernels_tflops": round(mfma_flops / (mfma_ms * 1e-3) / 1e12, 2) if mfma_ms else None,
                         "mfma_kernel_ms_per_step": round(mfma_ms / args.steps, 3),
                         "per_kernel": {names[f]: {"launches_per_step": round(fam[f][0] / args.steps, 1),
                                                   "tflops": round(fam[f][1] / (fam[f][2] * 1e-3) / 1e12, 2),
                                                   "ms_per_step": round(fam[f][2] / args.steps, 3)}
                                        for f in range(7) if fam[f][2] > 0}},
            "step_tflops_as_written": round(GFLOP_PER_IMG_AS_WRITTEN * value / 1e3, 2),
            "step_frac_of_f32_mfma_peak": round(GFLOP_PER_IMG_AS_WRITTEN * value / 1e3 / world / PEAK_F32_MFMA_TFLOPS, 4),
            "final_loss": loss_val, "hip_graph": graphed,
        }
        if b32 is not None:
            out["train_b32_one_gpu_images_per_sec"] = round(b32, 1)      # same step at batch 32 on this one GPU
        if fps is not None:
            out["stylize_720p_fps"] = round(fps, 2)
            out["stylize_1080p_b8_bf16_fps"] = round(fps_1080_bf16, 1)     # FS_FLAG_BF16: ~52 dB PSNR vs the fp32 path
            out["stylize_1080p_b8_fp32_fps"] = round(fps_1080_f32, 1)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_images, S)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
