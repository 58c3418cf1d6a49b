#!/usr/bin/env python
"""Drop-in for the reference's train.py (same flags, checkpoint names, logging cadence):
trains the image-transform net against the VGG16 perceptual loss -- every FLOP in
libfaststyle_hip.so on MI355X, optionally data-parallel (one process per GPU, launched with
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P train.py ...``
-- the launcher ``bench.py --gpus N`` uses for itself; ``--batch_size`` is then the per-GPU batch and gradients are
SUM-all-reduced once per step over RCCL).  The forward + backward of a step replay ONE captured hipGraph, exactly
what bench.py times (``--no_graph`` launches the ~170 kernels eagerly instead).

Differences from the reference that are forced by the environment and stated, not hidden:
  * ``--train_dir`` is, as in the reference, a directory of ``train-*`` TFRecord shards written by
    tfrecords_writer.py (read by faststyle_amd/datapipe.py: native record/Example reader, threaded
    JPEG decode, TF1-bicubic resize kernel, HBM-resident shuffle queue of ``--num_pipe_buffer``
    images).  Two conveniences on top: a directory of plain JPEG/PNG files (PIL decode + PIL
    bicubic), and the literal ``synthetic`` (uniform [0,255) images, MS-COCO train2014 count
    82,783 per epoch) for benchmarking without a dataset;
  * next to the TensorBoard event file (``summaries/train/<run_name>/events.out.tfevents.*`` with the
    reference's four scalars ``summaries/{loss,style_loss,content_loss,tv_loss}`` at the same steps,
    train.py:185-189, 260-272; no graph definition in it) the same values go to ``scalars.jsonl``.
"""
import glob
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def setup_parser():
    """The reference flag surface (train.py:23-105), defined in faststyle_amd/cli.py."""
    from faststyle_amd import cli
    return cli.train_parser()


COCO_TRAIN2014 = 82783


def batcher(train_dir, batch_size, resize, n_epochs, buffer_size, seed, rank, world):
    """Yields float32 [B,H,W,3] RGB 0..255 numpy batches until n_epochs are exhausted
    (role of reference datapipe.batcher, datapipe.py:55-78: decode -> bicubic resize ->
    shuffle buffer -> batches; the trailing partial batch is dropped like shuffle_batch does)."""
    rng = np.random.default_rng(seed + rank)
    H, W = resize
    if train_dir == "synthetic":
        per_rank = COCO_TRAIN2014 * n_epochs // world
        for _ in range(per_rank // batch_size):
            yield rng.uniform(0, 255, (batch_size, H, W, 3)).astype(np.float32)
        return
    from PIL import Image
    files = sorted(os.path.join(train_dir, f) for f in os.listdir(train_dir)
                   if f.lower().endswith((".jpg", ".jpeg", ".png")))
    if not files:
        raise SystemExit("no images under --train_dir %s" % train_dir)
    files = files[rank::world]
    buf = []

    def load(path):
        im = Image.open(path).convert("RGB").resize((W, H), Image.BICUBIC)
        return np.asarray(im, dtype=np.float32)

    def drain(min_keep):
        while len(buf) > min_keep and len(buf) >= batch_size:
            idx = rng.choice(len(buf), batch_size, replace=False)
            out = np.stack([buf[i] for i in idx])
            for i in sorted(idx, reverse=True):
                buf.pop(i)
            yield out

    for _ in range(n_epochs):
        for f in files:
            buf.append(load(f))
            if len(buf) >= buffer_size + 3 * batch_size:
                for b in drain(buffer_size):
                    yield b
    for b in drain(0):
        yield b


def main(args):
    import torch
    import torch.distributed as dist
    from faststyle_amd import ckpt, datapipe, engine, im_transf_net, tbevents, trainer, utils, vgg16

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FS_DIST_BACKEND=gloo + FS_DIST_SHARE_GPU=1: test switches -- N ranks on the ONE GPU of a test box (RCCL refuses two ranks
    # on one device; gloo stages the 1.7 MB gradient through the host).  Production: RCCL, one GPU per rank.
    backend = os.environ.get("FS_DIST_BACKEND", "nccl")
    if os.environ.get("FS_DIST_SHARE_GPU") == "1":
        local = local_rank = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
    eng = engine.Engine(engine.TorchMem("cuda:%d" % local))

    # Load in style image that will define the model (train.py:135-137).
    style_img = utils.imread(args.style_img_path)
    style_img = utils.imresize(style_img, args.style_target_resize)
    style_img = style_img[np.newaxis, :].astype(np.float32)

    cfg = dict(content_layers=args.loss_content_layers, content_weights=args.content_weights,
               style_layers=args.loss_style_layers, style_weights=args.style_weights, beta=args.beta)
    if rank == 0:
        print('Precomputing target style layers.')
    vgg_w = vgg16.load_weights('libs/vgg16_weights.npz')          # train.py:148, 239 (path relative to CWD)
    method = args.upsample_method
    params = eng.flatten_params(im_transf_net.initial_variables(seed=0, upsample_method=method), scope="",
                                upsample_method=method)
    tr = trainer.Trainer(eng, params, vgg_w, style_img, cfg, learn_rate=args.learn_rate,
                         dist=dist if world > 1 else None, upsample_method=method, use_graph=not args.no_graph)

    # Log directory of this run (behaviour of train.py:207-217): --run_name when given, otherwise the first
    # "<model_name><k>", k = 0, 1, 2, ..., that is not yet a directory under summaries/train/.
    run_name = args.run_name
    if rank == 0:
        base = os.path.join('.', 'summaries', 'train')
        os.makedirs(base, exist_ok=True)
        if run_name is None:
            k = 0
            while os.path.isdir(os.path.join(base, '%s%d' % (args.model_name, k))):
                k += 1
            run_name = '%s%d' % (args.model_name, k)
        for d in ('./training', './models', './summaries/train/' + run_name):
            if not os.path.exists(d):
                os.makedirs(d)
        log = open('./summaries/train/' + run_name + '/scalars.jsonl', 'a')
        train_writer = tbevents.EventWriter('./summaries/train/' + run_name)      # train.py:216-217

    def save(prefix, full):
        # full: saver = tf.train.Saver() -- all variables incl. Adam slots and global_step (train.py:224)
        ckpt.save_checkpoint(prefix, tr.state_tensors(full=full))

    if args.resume_from:
        step0 = tr.load_state(ckpt.load_checkpoint(args.resume_from))
        if rank == 0:
            print('Resumed from %s at step %d.' % (args.resume_from, step0))

    # Input pipeline (train.py:192-196): TFRecord shards train-* when present
    shards = sorted(glob.glob(os.path.join(args.train_dir, 'train-*'))) if args.train_dir != 'synthetic' else []

    def common_step_count(n_local):
        """Ranks read disjoint shards / files of unequal size: every rank must run the SAME number of steps, or the
        ones with a batch left over block forever in the gradient all-reduce of a peer that already left."""
        if world == 1:
            return n_local
        t = torch.tensor([n_local], device="cuda:%d" % local_rank, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    if shards:
        if len(shards) < world:            # identical on every rank (same directory listing): all leave together, no collective pending
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit("--train_dir holds %d train-* shards for %d ranks: every rank needs at least one" % (len(shards), world))
        cap = common_step_count(datapipe.count_records(shards[rank::world]) * args.n_epochs // args.batch_size)
        batches = datapipe.batcher(shards, args.batch_size, args.preprocess_size, args.n_epochs,
                                   args.num_pipe_buffer, engine=eng, seed=1234, rank=rank, world=world,
                                   max_batches=cap)
    else:
        gen = batcher(args.train_dir, args.batch_size, args.preprocess_size, args.n_epochs, args.num_pipe_buffer, 1234, rank, world)
        if args.train_dir != 'synthetic':   # a directory of image files, dealt files[rank::world]: counts differ by up to one image
            n_files = len([f for f in os.listdir(args.train_dir) if f.lower().endswith((".jpg", ".jpeg", ".png"))])
            n_mine = len(range(rank, n_files, world))
            gen = itertools.islice(gen, common_step_count(n_mine * args.n_epochs // args.batch_size))
        batches = (eng.mem.from_numpy(b) for b in gen)
    if rank == 0:
        print('Starting training...')
    clean_exit = False        # the loop ended the same way on every rank (epochs exhausted / num_steps_break)
    try:
        for batch in batches:
            current_step = tr.global_step
            if rank == 0 and current_step % args.num_steps_ckpt == 0:
                # Save a checkpoint (train.py:256-259), incl. step 0
                save('training/' + args.model_name + '.ckpt-%d' % current_step, full=True)
            losses = tr.step(batch)
            if current_step % args.num_steps_ckpt == 0 or current_step % 10 == 0:
                if world > 1:
                    dist.all_reduce(losses, op=dist.ReduceOp.SUM)     # losses are batch-summed (losses.py:32,63)
                lv = [float(v) for v in eng.mem.to_numpy(losses)]
                if rank == 0:
                    log.write(json.dumps({"step": current_step, "loss": lv[0], "content_loss": lv[1],
                                          "style_loss": lv[2], "tv_loss": lv[3]}) + "\n")
                    log.flush()
                    train_writer.add_scalars(current_step, [("summaries/loss", lv[0]), ("summaries/style_loss", lv[2]),
                                                            ("summaries/content_loss", lv[1]), ("summaries/tv_loss", lv[3])])
                    print(current_step, lv[0])
            if current_step == args.num_steps_break:
                if rank == 0:
                    print('Done training.')
                break
        else:
            if rank == 0:
                print('Done training.')
        clean_exit = True
    finally:
        # Save the model (the image transformation network) for later usage (train.py:283-286)
        try:
            if rank == 0:
                save('models/' + args.model_name + '_final.ckpt', full=False)
                train_writer.close()
        finally:
            if world > 1:
                if clean_exit:
                    # every rank ran the same number of steps: leave together, so that a rank tearing the process group down
                    # does not turn a peer's last collective into a watchdog abort
                    dist.barrier()
                    dist.destroy_process_group()
                else:
                    # THIS rank is leaving through an exception while its peers sit in the gradient all-reduce of a step it
                    # never reached.  A barrier here would pair with that all-reduce (mismatched collectives: undefined
                    # behaviour, at best a watchdog timeout).  Exit at once and non-zero so that torch.distributed.run tears the
                    # peers down.  If THIS rank is rank 0 its final model is on disk (the save above); if a peer failed, rank 0
                    # is killed inside its all-reduce and the newest training/<name>.ckpt-<step> (+ --resume_from) is what
                    # survives.  os._exit skips atexit handlers: close what this rank holds open first.
                    import traceback
                    traceback.print_exc()
                    if rank == 0:
                        try:
                            log.close()
                        except Exception:
                            pass
                    sys.stdout.flush()
                    sys.stderr.flush()
                    os._exit(1)
    return tr          # (the reference's main returns nothing; tests look at the trainer: hipGraph captured, step count)


if __name__ == "__main__":
    parser = setup_parser()
    main(parser.parse_args())
