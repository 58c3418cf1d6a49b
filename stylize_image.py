#!/usr/bin/env python
"""Drop-in for the reference's stylize_image.py (same flags, same messages, same file formats):
loads a TF bundle-V2 checkpoint of the image-transform net and filters one image -- on the
MI355X through libfaststyle_hip.so instead of a TF1 session (reference stylize_image.py:19-82).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def setup_parser():
    """The reference flag surface (stylize_image.py:19-43), defined in faststyle_amd/cli.py."""
    from faststyle_amd import cli
    return cli.stylize_image_parser()


def main(argv=None):
    args = setup_parser().parse_args(argv)
    from faststyle_amd import ckpt, engine, utils
    from faststyle_amd.im_transf_net import create_net

    # Read + preprocess input image (stylize_image.py:57-60).
    img = utils.imread(args.input_img_path)
    img = utils.imresize(img, args.content_target_resize)
    img_4d = img[np.newaxis, :].astype(np.float32)

    eng = engine.Engine()                 # fails loudly without the HIP library / a GPU
    print('Loading up model...')
    variables = eng.mem.from_numpy(eng.flatten_params(ckpt.load_checkpoint(args.model_path),
                                                      upsample_method=args.upsample_method))
    print('Evaluating...')
    Y = create_net(eng.mem.from_numpy(img_4d), args.upsample_method, variables=variables, engine=eng)
    img_out = eng.mem.to_numpy(Y)

    print('Saving image.')
    utils.imwrite(args.output_img_path, np.squeeze(img_out))
    print('Done.')


if __name__ == '__main__':
    main()
