"""Bundle-V2 checkpoint reader/writer against the reference's shipped models
(models/*_final.ckpt.* are data files copied from the reference repo: SURVEY.md §8a-W)."""
import filecmp
import os

import numpy as np
import pytest

from faststyle_amd import ckpt
from oracle import tnet


@pytest.mark.parametrize("style", ["starry", "candy"])
def test_shipped_ckpt_layout_and_roundtrip(repo_root, tmp_path, style):
    prefix = os.path.join(repo_root, "models", style + "_final.ckpt")
    w = ckpt.load_checkpoint(prefix, verify=True)          # verifies block + tensor CRC32C
    shapes = tnet.param_shapes("resize")
    assert list(w) == ["img_t_net/" + k for k in shapes]   # 48 tensors, sorted-key order
    assert all(w["img_t_net/" + k].shape == s for k, s in shapes.items())
    assert all(v.dtype == np.float32 for v in w.values())
    assert sum(v.size for v in w.values()) == 424102
    idx = ckpt.read_index(prefix)
    assert idx["img_t_net/initconv_0/W"]["offset"] == 128
    assert idx["img_t_net/upsample_2/W"]["offset"] == 1680856
    out = str(tmp_path / "rt.ckpt")
    ckpt.save_checkpoint(out, w)
    assert filecmp.cmp(out + ".index", prefix + ".index", shallow=False)
    assert filecmp.cmp(out + ".data-00000-of-00001", prefix + ".data-00000-of-00001", shallow=False)


def test_crc32c_known_answers():
    assert ckpt.crc32c(b"123456789") == 0xE3069283          # standard CRC-32C check value
    assert ckpt.crc32c(b"\x00" * 32) == 0x8A9136AA          # RFC 3720 B.4
    assert ckpt.crc32c(b"\xff" * 32) == 0x62A8AB43


def test_corruption_detected(repo_root, tmp_path):
    prefix = os.path.join(repo_root, "models", "starry_final.ckpt")
    w = ckpt.load_checkpoint(prefix)
    out = str(tmp_path / "bad.ckpt")
    ckpt.save_checkpoint(out, w)
    with open(out + ".data-00000-of-00001", "r+b") as f:
        f.seek(4000)
        f.write(b"\x01")
    with pytest.raises(ValueError):
        ckpt.load_checkpoint(out)


def test_mixed_dtypes_and_scalars(tmp_path):
    t = {"global_step": np.array(7, dtype=np.int64), "a/b": np.arange(6, dtype=np.float32).reshape(2, 3),
         "z": np.zeros((0,), np.float32)}
    out = str(tmp_path / "m.ckpt")
    ckpt.save_checkpoint(out, t)
    r = ckpt.load_checkpoint(out)
    assert list(r) == sorted(t)
    for k in t:
        assert r[k].dtype == t[k].dtype and r[k].shape == t[k].shape
        np.testing.assert_array_equal(r[k], t[k])
