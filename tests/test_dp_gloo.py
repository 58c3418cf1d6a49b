"""Data-parallel path on CPU: world_size 2, gloo backend, the kernel emulator as the engine.

Checks what the 8-GPU RCCL run relies on: after ONE all-reduce(SUM) of the flat
gradient buffer (a) every rank holds bit-identical parameters, and (b) the reduced gradient equals
the single-process gradient of the concatenated global batch (losses are batch-summed, instance
norm is per sample)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TorchCpuMem(object):
    """torch CPU tensors as engine memory (so torch.distributed/gloo can reduce them)."""

    def empty(self, shape):
        return torch.full(tuple(int(s) for s in shape), float("nan"), dtype=torch.float32)

    def zeros(self, shape):
        return torch.zeros(tuple(int(s) for s in shape), dtype=torch.float32)

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).copy())

    def to_numpy(self, t):
        return t.numpy()

    def ptr(self, t):
        if t is None:
            return None
        assert t.is_contiguous() and t.dtype == torch.float32
        return t.data_ptr()

    def stream(self):
        return 0

    def device_index(self):
        return 0

    def view(self, t, offset, shape):
        n = int(np.prod(shape))
        return t.view(-1)[offset:offset + n].view(*shape)


def _make_trainer(use_dist):
    import ctypes
    from faststyle_amd import _lib, engine, im_transf_net, trainer, vgg16
    from tests import emu_lib
    lib = _lib.bind(ctypes.CDLL(emu_lib.build_emu()))
    eng = engine.Engine(mem=TorchCpuMem(), lib=lib)
    params = eng.flatten_params(im_transf_net.initial_variables(seed=0), scope="")
    style = np.random.default_rng(2).uniform(0, 255, (1, 24, 28, 3)).astype(np.float32)
    cfg = engine.default_loss_cfg()
    return trainer.Trainer(eng, params, vgg16.synthetic_weights(3), style, cfg, dist=dist if use_dist else None)


def _batch():
    return np.random.default_rng(5).uniform(0, 255, (2, 44, 48, 3)).astype(np.float32)


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = _make_trainer(True)
    x = _batch()[rank:rank + 1]
    losses = tr.step(torch.from_numpy(x.copy()))
    np.save(os.path.join(outdir, "params_%d.npy" % rank), tr.params.numpy())
    np.save(os.path.join(outdir, "grads_%d.npy" % rank), tr.grads.numpy())
    np.save(os.path.join(outdir, "loss_%d.npy" % rank), losses.numpy())
    dist.destroy_process_group()


def test_two_rank_sum_allreduce_equals_global_batch(tmp_path):
    from tests import emu_lib
    emu_lib.build_emu()                      # build once, before forking
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = (np.load(str(tmp_path / ("params_%d.npy" % r))) for r in range(2))
    g0, g1 = (np.load(str(tmp_path / ("grads_%d.npy" % r))) for r in range(2))
    assert np.array_equal(p0, p1) and np.array_equal(g0, g1)          # (a) ranks stay in lock-step
    tr = _make_trainer(False)
    losses = tr.step(torch.from_numpy(_batch()))
    g = tr.grads.numpy()
    assert np.abs(g0 - g).max() / np.abs(g).max() < 1e-4                # (b) SUM of shards == global batch
    l0, l1 = (np.load(str(tmp_path / ("loss_%d.npy" % r))) for r in range(2))
    np.testing.assert_allclose(l0 + l1, losses.numpy(), rtol=1e-5)
