"""The data-parallel trainer under a live RCCL process group on the ONE GPU a test box has (world_size 1, backend
"nccl" == RCCL on ROCm): init_process_group with a device id, the parameter broadcast, hipGraph capture of
forward+backward while the process group (and its watchdog thread) exists, the all-reduce(SUM) -> TF-Adam ordering on
the launch stream.  With one rank the reduced gradient is the local gradient, so the run must match a trainer without
torch.distributed bit for bit.  (The 1 -> 8 GPU curve is the driver's to measure; tests/test_dp_gloo.py covers
world_size 2 on CPU.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from faststyle_amd import engine, im_transf_net, trainer, vgg16
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = engine.Engine(engine.TorchMem("cuda:0"))
params = eng.flatten_params(im_transf_net.initial_variables(seed=0), scope="")
style = np.random.default_rng(2).uniform(0, 255, (1, 96, 128, 3)).astype(np.float32)
vw = vgg16.synthetic_weights(seed=3)
a = trainer.Trainer(eng, params, vw, style, dist=dist, use_graph=True)
b = trainer.Trainer(eng, params, None, style, dist=None, use_graph=True)
g = torch.Generator(device="cuda"); g.manual_seed(7)
batches = [torch.rand((2, 128, 128, 3), device="cuda", generator=g) * 255.0 for _ in range(4)]
for x in batches:
    la = a.step(x).clone()
    lb = b.step(x).clone()
    assert torch.equal(la, lb), (la, lb)
torch.cuda.synchronize()
assert a.graph is not None and b.graph is not None, "hipGraph capture fell back to eager launches"
assert a.global_step == 4 and torch.equal(a.grads, b.grads) and torch.equal(a.params, b.params)
assert torch.equal(a.m, b.m) and torch.equal(a.v, b.v)
assert float(a.grads.abs().max()) > 0 and bool(torch.isfinite(a.params).all())
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""


@pytest.mark.gpu
def test_trainer_with_rccl_process_group_world1_matches_plain_trainer_bitwise():
    port = 29600 + (os.getpid() % 1000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "port": port}], cwd=ROOT, capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0 and "RCCL_WORLD1_OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


@pytest.mark.gpu
def test_bench_under_torchrun_one_process():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` is a supported command: the step then
    includes the RCCL all-reduce (world size 1)."""
    port = 29700 + (os.getpid() % 1000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
           "--no-cpu-baseline", "--no-stylize", "--b4-steps", "4", "--profile-steps", "1"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines[:6]     # ONE JSON line: RCCL's version banner goes to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["hip_graph"] is True and "RCCL all-reduce" in d["config"]["collective"]
    # the all-reduce and Adam stay outside the hipGraph: the device idles < 1 % of the step between two steps
    assert 0 <= d["host_gap_frac_of_step"] < 0.01, (d["inter_step_gap_us"], d["ms_per_step"])
    assert d["value"] > 0 and d["train_b4_per_gpu"]["images_per_sec"] > 0
