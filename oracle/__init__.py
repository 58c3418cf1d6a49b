"""CPU oracle for the faststyle hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement (dtype-generic: float32 or float64) of the reference's algorithm for

* the image-transform net forward (reference im_transf_net.py:14-276), and
* the perceptual-loss training step (reference libs/vgg16.py:36-220, utils.py:66-83,
  losses.py:12-97, train.py:157-204,245-275) incl. hand-derived backward passes and
  TF1-Adam.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / the CPU baseline -- never as a fallback
for the HIP path (``faststyle_amd`` never imports it).

Pinning: the reference cannot be executed in the build container (TF 1.0 /
Python 2 / OpenCV are absent), so the forward path is pinned against the reference's own
shipped known-answer data: ``results/chicago.jpg`` + ``models/{starry,candy}_final.ckpt``
-> ``results/{starry,candy}_chicago.jpg`` (tests/test_oracle_golden.py, >= 63 dB PSNR
after the same round->uint8->JPEG q95 4:2:0 re-encode).  The TRAINING path (VGG features,
Grams, losses, gradients, Adam) has no golden data anywhere in the reference (no VGG
weights, no logged losses): **training-path parity is unpinned by the reference**; it is
cross-checked against an independent torch-CPU float64 autograd restatement instead
(tests/test_oracle_backward.py).
"""
