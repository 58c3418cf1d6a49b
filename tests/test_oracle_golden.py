"""Pin the oracle's forward path against the reference's own known-answer data
(SURVEY.md §8c): results/chicago.jpg + models/<style>_final.ckpt -> results/<style>_chicago.jpg
(reference README.md:5-18).  Protocol: PIL decode -> oracle create_net (float32) ->
rint -> clip -> uint8 -> JPEG q95 4:2:0 (what cv2.imwrite does, utils.py:51-52) -> decode
-> compare with the decoded golden."""
import os

import numpy as np
import pytest

from faststyle_amd import ckpt
from oracle import tnet
from tests.imgutil import jpeg_roundtrip, load_rgb, psnr


@pytest.mark.parametrize("style", ["starry", "candy"])
def test_oracle_reproduces_shipped_golden(repo_root, style):
    assets = os.path.join(repo_root, "tests", "golden", "ref_assets")
    x = load_rgb(os.path.join(assets, "chicago.jpg")).astype(np.float32)[None]
    assert x.shape == (1, 474, 712, 3)
    P = tnet.strip_scope(ckpt.load_checkpoint(os.path.join(repo_root, "models", style + "_final.ckpt")))
    y = tnet.create_net(x, P, "resize")
    assert y.shape == (1, 476, 712, 3) == (1,) + tnet.out_shape(474, 712) + (3,)
    dec = jpeg_roundtrip(y[0])
    gold = load_rgb(os.path.join(assets, style + "_chicago.jpg"))
    assert gold.shape == dec.shape
    p = psnr(dec, gold)
    same = (dec == gold).mean()
    assert p >= 63.0, p            # measured 66.7 dB (starry) / 66.5 dB (candy)
    assert same >= 0.985, same     # measured 99.3 %
