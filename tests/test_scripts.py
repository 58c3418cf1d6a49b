"""The drop-in command lines on the GPU: stylize_image.py reproduces the reference's shipped golden
(README.md:5-18); train.py runs the reference's loop (checkpoint names, logging cadence, final model)
and its final checkpoint loads back through stylize_image.py."""
import json
import os
import sys

import numpy as np
import pytest

from tests.imgutil import load_rgb, psnr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_flags_match_reference():
    """Flag names/defaults of the reference's argparse (stylize_image.py:24-42, train.py:27-104)."""
    sys.path.insert(0, ROOT)
    import stylize_image
    import train
    s = vars(stylize_image.setup_parser().parse_args(["--input_img_path", "x.jpg"]))
    assert s == {"input_img_path": "x.jpg", "output_img_path": "./results/styled.jpg",
                 "model_path": "./models/starry_final.ckpt", "content_target_resize": 1.0, "upsample_method": "resize"}
    t = vars(train.setup_parser().parse_args([]))
    assert t == {"train_dir": None, "model_name": None, "style_img_path": "./style_images/starry_night_crop.jpg",
                 "learn_rate": 1e-3, "batch_size": 4, "n_epochs": 2, "preprocess_size": [256, 256], "run_name": None,
                 "loss_content_layers": ["conv3_3"], "loss_style_layers": ["conv1_2", "conv2_2", "conv3_3", "conv4_3"],
                 "content_weights": [1.0], "style_weights": [5.0, 5.0, 5.0, 5.0], "num_steps_ckpt": 1000,
                 "num_pipe_buffer": 4000, "num_steps_break": -1, "resume_from": None, "no_graph": False, "beta": 0.0, "style_target_resize": 1.0,
                 "upsample_method": "resize"}


@pytest.mark.gpu
@pytest.mark.parametrize("style", ["starry", "candy"])
def test_stylize_image_cli_reproduces_golden(tmp_path, style, capsys):
    # run in-process (main(argv)): the GPU boxes refuse a second process on the device while pytest holds it
    sys.path.insert(0, ROOT)
    import stylize_image
    out = str(tmp_path / "styled.jpg")
    stylize_image.main(["--input_img_path", os.path.join(ROOT, "tests", "golden", "ref_assets", "chicago.jpg"),
                        "--output_img_path", out, "--model_path", os.path.join(ROOT, "models", style + "_final.ckpt")])
    lines = [l for l in capsys.readouterr().out.splitlines() if l and "amdgpu" not in l]
    assert lines == ["Loading up model...", "Evaluating...", "Saving image.", "Done."]      # stylize_image.py:72-82
    got = load_rgb(out)
    gold = load_rgb(os.path.join(ROOT, "tests", "golden", "ref_assets", style + "_chicago.jpg"))
    assert got.shape == gold.shape == (476, 712, 3)
    assert psnr(got, gold) >= 63.0 and (got == gold).mean() >= 0.985


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["resize", "deconv"])
def test_train_cli_runs_reference_loop(tmp_path, monkeypatch, capsys, method):
    from faststyle_amd import ckpt, vgg16
    sys.path.insert(0, ROOT)
    import stylize_image
    import train
    work = tmp_path
    (work / "libs").mkdir()
    np.savez(str(work / "libs" / "vgg16_weights.npz"), **vgg16.synthetic_weights(3))     # train.py:148 reads it from CWD
    monkeypatch.chdir(work)
    tr = train.main(train.setup_parser().parse_args(
        ["--train_dir", "synthetic", "--model_name", "t",
         "--style_img_path", os.path.join(ROOT, "style_images", "starry_night_crop.jpg"),
         "--style_target_resize", "0.25", "--preprocess_size", "128", "128", "--batch_size", "2",
         "--num_steps_break", "11", "--num_steps_ckpt", "10", "--upsample_method", method] +
        (["--beta", "1e-4"] if method == "deconv" else [])))
    # the script runs what bench.py times: forward + backward replay ONE captured hipGraph (opt-out: --no_graph)
    assert tr.use_graph and tr.graph is not None and tr.global_step == 12
    out = [l for l in capsys.readouterr().out.splitlines() if l and "amdgpu" not in l]
    assert out[0] == "Precomputing target style layers." and "Starting training..." in out and out[-1] == "Done training."
    steps = [int(l.split()[0]) for l in out if l.split()[0].isdigit()]
    assert steps == [0, 10]                                                    # printed every 10 steps (train.py:265-272)
    assert os.path.exists(str(work / "training" / "t.ckpt-0.index")) and os.path.exists(str(work / "training" / "t.ckpt-10.index"))
    final = ckpt.load_checkpoint(str(work / "models" / "t_final.ckpt"))        # train.py:286
    assert len(final) == 48 and all(k.startswith("img_t_net/") for k in final)
    # deconv2d keeps its filter as [k,k,Cout,Cin] (im_transf_net.py:174)
    assert final["img_t_net/upsample_0/W"].shape == ((3, 3, 32, 64) if method == "deconv" else (3, 3, 64, 32))
    full = ckpt.load_checkpoint(str(work / "training" / "t.ckpt-10"))
    assert int(full["global_step"]) == 10 and "img_t_net/initconv_0/W/Adam" in full
    assert any(f.startswith("events.out.tfevents.") for f in os.listdir(str(work / "summaries" / "train" / "t0")))
    logs = [json.loads(l) for l in open(str(work / "summaries" / "train" / "t0" / "scalars.jsonl"))]
    assert [d["step"] for d in logs] == [0, 10] and np.isfinite(logs[1]["loss"])
    assert logs[1]["loss"] < logs[0]["loss"] or method == "deconv"
    assert (logs[1]["tv_loss"] > 0) == (method == "deconv")                  # --beta only set for deconv
    # continue from the step-10 bundle (weights + Adam slots + global_step): the step counter carries on
    train.main(train.setup_parser().parse_args(
        ["--train_dir", "synthetic", "--model_name", "t", "--run_name", "t0",
         "--style_img_path", os.path.join(ROOT, "style_images", "starry_night_crop.jpg"),
         "--style_target_resize", "0.25", "--preprocess_size", "128", "128", "--batch_size", "2",
         "--num_steps_break", "20", "--num_steps_ckpt", "10", "--upsample_method", method,
         "--resume_from", str(work / "training" / "t.ckpt-10")]))
    out2 = [l for l in capsys.readouterr().out.splitlines() if l and "amdgpu" not in l]
    assert any(l.startswith("Resumed from") and l.endswith("at step 10.") for l in out2)
    assert [int(l.split()[0]) for l in out2 if l.split()[0].isdigit()] == [10, 20]
    assert int(ckpt.load_checkpoint(str(work / "training" / "t.ckpt-20"))["global_step"]) == 20
    # the final model is a valid stylize_image.py --model_path
    outimg = str(work / "o.jpg")
    stylize_image.main(["--input_img_path", os.path.join(ROOT, "tests", "golden", "ref_assets", "chicago.jpg"),
                        "--output_img_path", outimg, "--model_path", str(work / "models" / "t_final.ckpt"),
                        "--content_target_resize", "0.25", "--upsample_method", method])
    assert load_rgb(outimg).shape[2] == 3


@pytest.mark.gpu
def test_autograd_glue_matches_direct_calls():
    """loss.backward() through the torch.autograd wrappers == the direct C-ABI call sequence."""
    import torch
    from faststyle_amd import autograd as fa, engine, im_transf_net, vgg16
    e = engine.Engine()
    e.vgg_load(vgg16.synthetic_weights(3))
    cfg = engine.default_loss_cfg()
    rng = np.random.default_rng(0)
    tg = e.style_targets(e.mem.from_numpy(rng.uniform(0, 255, (1, 64, 80, 3)).astype(np.float32)), cfg)
    X = e.mem.from_numpy(rng.uniform(0, 255, (2, 64, 64, 3)).astype(np.float32))
    flat = e.mem.from_numpy(e.flatten_params(im_transf_net.initial_variables(0), scope=""))
    v = flat.clone().requires_grad_(True)
    loss = fa.PerceptualLoss.apply(fa.TransformNet.apply(v, X, e), X, e, tg, cfg)
    loss.backward()
    y = e.tnet_forward(flat, X, save_for_bwd=True)
    losses, dy = e.perceptual_loss(y, X, tg, cfg)
    g = e.tnet_backward(flat, X, dy)
    assert torch.equal(v.grad, g) and float(loss) == float(losses[0])


@pytest.mark.gpu
def test_resume_is_bit_identical_to_uninterrupted_run(tmp_path):
    """state_tensors -> bundle on disk -> load_state continues exactly where the run stopped."""
    import torch
    from faststyle_amd import ckpt, engine, im_transf_net, trainer, vgg16
    e = engine.Engine()
    rng = np.random.default_rng(0)
    style = rng.uniform(0, 255, (1, 64, 80, 3)).astype(np.float32)
    p0 = e.flatten_params(im_transf_net.initial_variables(0), scope="")
    batches = [e.mem.from_numpy(rng.uniform(0, 255, (2, 64, 64, 3)).astype(np.float32)) for _ in range(3)]
    a = trainer.Trainer(e, p0, vgg16.synthetic_weights(3), style)
    a.step(batches[0])
    a.step(batches[1])
    ckpt.save_checkpoint(str(tmp_path / "run.ckpt-2"), a.state_tensors(full=True))
    a.step(batches[2])
    b = trainer.Trainer(e, p0, vgg16.synthetic_weights(3), style)
    assert b.load_state(ckpt.load_checkpoint(str(tmp_path / "run.ckpt-2"))) == 2
    b.step(batches[2])
    assert b.global_step == 3
    assert torch.equal(a.params, b.params) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v)
