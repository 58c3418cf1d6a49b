#!/usr/bin/env python
"""Drop-in for the reference's slow_style.py (the Gatys et al. optimisation variant; same flags):
optimises the PIXELS of an image against the VGG16 perceptual loss with TF-style Adam -- the same
HIP kernels as the training path (fs_style_targets, fs_perceptual_loss, fs_adam_tf_step), with the
image itself as the variable.

Reference semantics kept (slow_style.py:117-183): white-noise init ``rand*255`` of the content
image's shape, loss = content + style + beta*tv with beta defaulting to 1e-4, Adam(lr=10), and the
loop ``while current_step < num_steps_break`` that reads the step BEFORE the update -- i.e.
num_steps_break + 1 updates, a loss line at every step divisible by 10.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def setup_parser():
    """The reference flag surface (slow_style.py:17-67), defined in faststyle_amd/cli.py."""
    from faststyle_amd import cli
    return cli.slow_style_parser()


def optimise(eng, vgg_weights, style_img, cont_img, cfg, learn_rate, num_steps_break, seed=None, log=print):
    """style_img/cont_img: float32 [1,H,W,3] RGB 0..255.  Returns the optimised image [1,H,W,3] (numpy)."""
    mem = eng.mem
    eng.vgg_load(vgg_weights)
    log('Precomputing target style layers.')
    target_grams = eng.style_targets(mem.from_numpy(style_img), cfg)
    rng = np.random.RandomState(seed)
    X = mem.from_numpy((rng.rand(*cont_img.shape) * 255.0).astype(np.float32))     # slow_style.py:118-120
    cont = mem.from_numpy(cont_img)
    log('Precomputing target content layers.')      # (the content features ride along in fs_perceptual_loss)
    n = int(np.prod(cont_img.shape))
    m, v = mem.zeros((n,)), mem.zeros((n,))
    current_step = 0
    global_step = 0
    while current_step < num_steps_break:
        current_step = global_step
        losses, dX = eng.perceptual_loss(X, cont, target_grams, cfg)
        global_step += 1
        eng.adam_tf_step(mem.view(X, 0, (n,)), mem.view(dX, 0, (n,)), m, v, global_step, lr=learn_rate)
        if current_step % 10 == 0:
            log('%d %s' % (current_step, float(mem.to_numpy(losses)[0])))
    return mem.to_numpy(X)


def main(args):
    from faststyle_amd import engine, utils, vgg16
    style_img = utils.imread(args.style_img_path)
    style_img = utils.imresize(style_img, args.style_target_resize)
    style_img = style_img[np.newaxis, :].astype(np.float32)
    cont_img = utils.imread(args.cont_img_path)
    cont_img = utils.imresize(cont_img, args.cont_target_resize)
    cont_img = cont_img[np.newaxis, :].astype(np.float32)
    cfg = dict(content_layers=args.loss_content_layers, content_weights=args.content_weights,
               style_layers=args.loss_style_layers, style_weights=args.style_weights, beta=args.beta)
    eng = engine.Engine()
    vgg_w = vgg16.load_weights('libs/vgg16_weights.npz')             # slow_style.py:100 (path relative to CWD)
    img_out = optimise(eng, vgg_w, style_img, cont_img, cfg, args.learn_rate, args.num_steps_break)
    utils.imwrite(args.output_img_path, np.squeeze(img_out))


if __name__ == "__main__":
    parser = setup_parser()
    main(parser.parse_args())
