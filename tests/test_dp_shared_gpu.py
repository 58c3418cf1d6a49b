"""The WHOLE multi-rank path on the one GPU a test box has (north_star: "1, 2, 4 and 8 GPUs").

RCCL refuses two ranks on one device, so these runs switch the process group to gloo (FS_DIST_BACKEND=gloo) and put every
rank on GPU 0 (FS_DIST_SHARE_GPU=1): the ranks time-share the chip, the 1.7 MB gradient is staged through the host.
Everything else is the production path of an N-GPU lease: bench.py's self-launch through torch.distributed.run, both train
legs (batch 32 and batch 4 per rank) with the SUM all-reduce inside the step, the stylize legs, max-over-ranks timing,
per-rank rates, ONE JSON line from rank 0; and train.py's data-parallel loop (parameter broadcast, sharded batches,
all-reduce -> identical TF-Adam on every rank).  The rates these runs print are NOT measurements of anything."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict((k, v) for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"))
    env.update(FS_DIST_BACKEND="gloo", FS_DIST_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


@pytest.mark.gpu
def test_bench_full_path_two_ranks_sharing_the_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--b4-steps", "4", "--profile-steps", "1", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True,
                         timeout=1500, env=_env())
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines[:6]            # rank 0 alone prints, ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["hip_graph"] is True
    assert d["config"]["global_batch"] == 64 and d["config"]["batch_per_gpu"] == 32 and d["config"]["parallelism"] == "dp2"
    assert "gloo all-reduce(SUM)" in d["config"]["collective"] and "TEST MODE" in d["config"]["collective"]
    pr = d["per_rank_images_per_sec"]
    assert len(pr) == 2 and all(r > 0 for r in pr)
    # whole-job value = both ranks' images over the SLOWEST rank's time
    assert abs(d["value"] - 2 * min(pr)) / d["value"] < 0.02, (d["value"], pr)
    assert abs(d["value"] - 64 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-2
    b4 = d["train_b4_per_gpu"]
    assert "configs[3]" in b4["config"] and b4["global_batch"] == 8 and len(b4["per_rank_images_per_sec"]) == 2
    assert d["stylize_720p"]["fps"] > 0 and d["stylize_1080p_b8_bf16"]["fps"] > 0 and d["stylize_1080p_b8_fp32"]["fps"] > 0
    assert "cpu_baseline" not in d                                           # rank 0 at N = 1 only
    assert np.isfinite(d["final_loss"]) and d["roofline"]["frac"] > 0


TRAIN_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import train
args = train.setup_parser().parse_args(
    ["--train_dir", "synthetic", "--model_name", "dp", "--style_img_path", %(style)r, "--style_target_resize", "0.25",
     "--preprocess_size", "96", "96", "--batch_size", "2", "--num_steps_break", "2", "--num_steps_ckpt", "10"])
tr = train.main(args)
rank = int(os.environ["RANK"])
np.save(os.path.join(%(work)r, "params_rank%%d.npy" %% rank), tr.params.cpu().numpy())
np.save(os.path.join(%(work)r, "adam_m_rank%%d.npy" %% rank), tr.m.cpu().numpy())
assert tr.global_step == 3 and tr.use_graph and tr.graph is not None
"""


@pytest.mark.gpu
def test_train_py_world2_three_steps_identical_parameters(tmp_path):
    from faststyle_amd import ckpt, vgg16
    work = tmp_path
    (work / "libs").mkdir()
    np.savez(str(work / "libs" / "vgg16_weights.npz"), **vgg16.synthetic_weights(3))          # train.py reads it from CWD
    script = work / "run_dp.py"
    script.write_text(TRAIN_SCRIPT % {"root": ROOT, "work": str(work),
                                      "style": os.path.join(ROOT, "style_images", "starry_night_crop.jpg")})
    port = 29800 + (os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, cwd=str(work), capture_output=True, text=True, timeout=900, env=_env())
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-4000:])
    p0, p1 = np.load(str(work / "params_rank0.npy")), np.load(str(work / "params_rank1.npy"))
    m0, m1 = np.load(str(work / "adam_m_rank0.npy")), np.load(str(work / "adam_m_rank1.npy"))
    # the ranks saw DIFFERENT batches; one SUM all-reduce per step + identical Adam keeps them bit-identical
    assert np.array_equal(p0, p1) and np.array_equal(m0, m1) and np.isfinite(p0).all() and np.abs(m0).max() > 0
    # rank 0 alone wrote the run's files, and the final model is the parameters both ranks hold
    final = ckpt.load_checkpoint(str(work / "models" / "dp_final.ckpt"))
    assert len(final) == 48
    lines = [l for l in out.stdout.splitlines() if l.strip() and "amdgpu" not in l]
    assert lines.count("Starting training...") == 1 and lines.count("Done training.") == 1
    from faststyle_amd import _lib as L
    assert sum(v.size for v in final.values()) == L.FS_TNET_NPARAMS == p0.size
