"""Command-line surfaces of the drop-in scripts.  The FLAG NAMES, DEFAULTS, TYPES and CHOICES are the
reference's interface (stylize_image.py:19-43, train.py:23-105, slow_style.py:17-67,
stylize_webcam.py:17-39) and are kept exactly -- tests/test_scripts.py and tests/test_aux_scripts.py
pin them; the help texts are this project's own."""
import argparse

LAYERS_STYLE = ['conv1_2', 'conv2_2', 'conv3_3', 'conv4_3']
UPSAMPLE = dict(choices=['resize', 'deconv'], default='resize',
                help="how the model's two upsampling layers were built: 'resize' (nearest x4 + stride-2 conv, the shipped "
                     "models) or 'deconv' (conv2d_transpose); a mismatch with the checkpoint is an error")

LOSS_FLAGS = [
    ('--loss_content_layers', dict(nargs='*', default=['conv3_3'], help='VGG16 layers of the content loss')),
    ('--loss_style_layers', dict(nargs='*', default=LAYERS_STYLE, help='VGG16 layers whose Gram matrices define the style loss')),
    ('--content_weights', dict(nargs='*', default=[1.0], type=float, help='one weight per content layer')),
    ('--style_weights', dict(nargs='*', default=[5.0, 5.0, 5.0, 5.0], type=float, help='one weight per style layer')),
]


def _build(description, flags):
    parser = argparse.ArgumentParser(description=description)
    for name, kw in flags:
        parser.add_argument(name, **kw)
    return parser


def stylize_image_parser():
    return _build("Filter one image through a trained transform net (HIP engine on MI355X).", [
        ('--input_img_path', dict(help='image to stylize')),
        ('--output_img_path', dict(default='./results/styled.jpg', help='where the result is written')),
        ('--model_path', dict(default='./models/starry_final.ckpt', help='checkpoint prefix of the trained net (bundle V2)')),
        ('--content_target_resize', dict(default=1.0, type=float, help='scale factor applied to the input first')),
        ('--upsample_method', dict(UPSAMPLE)),
    ])


def train_parser():
    return _build("Train a transform net against the VGG16 perceptual loss (HIP engine, optionally data-parallel).", [
        ('--train_dir', dict(help="directory of train-* TFRecord shards (or of image files, or the literal 'synthetic')")),
        ('--model_name', dict(help='name used for checkpoints, the final model and the default log directory')),
        ('--style_img_path', dict(default='./style_images/starry_night_crop.jpg', help='style target image')),
        ('--learn_rate', dict(default=1e-3, type=float, help='Adam step size')),
        ('--batch_size', dict(default=4, type=int, help='images per step (per GPU when launched data-parallel)')),
        ('--n_epochs', dict(default=2, type=int, help='passes over the training set')),
        ('--preprocess_size', dict(default=[256, 256], nargs=2, type=int, help='training images are resized to H W')),
        ('--run_name', dict(default=None, help='log sub-directory under ./summaries/train (default: <model_name><k>)')),
    ] + LOSS_FLAGS + [
        ('--num_steps_ckpt', dict(default=1000, type=int, help='checkpoint period in steps')),
        ('--num_pipe_buffer', dict(default=4000, type=int, help='images held by the shuffle queue (min_after_dequeue)')),
        ('--num_steps_break', dict(default=-1, type=int, help='stop after this step (-1: run all epochs)')),
        ('--resume_from', dict(default=None, help='bundle prefix of a training/<model_name>.ckpt-<step> to continue from '
                                             '(weights, Adam slots, global_step); not in the reference')),
        ('--no_graph', dict(action='store_true', help='launch every kernel of a step eagerly instead of replaying the captured '
                                                      'hipGraph (the default, and what bench.py times); not in the reference')),
        ('--beta', dict(default=0.0, type=float, help='total-variation weight (about 1e-4 helps deconv models)')),
        ('--style_target_resize', dict(default=1.0, type=float, help='scale factor applied to the style image')),
        ('--upsample_method', dict(UPSAMPLE)),
    ])


def slow_style_parser():
    return _build("Optimise the pixels of an image against the perceptual loss (Gatys et al.), VGG16 variant.", [
        ('--style_img_path', dict(help='style image')),
        ('--cont_img_path', dict(help='content image')),
        ('--learn_rate', dict(default=1e1, type=float, help='Adam step size on the 0..255 pixel scale')),
    ] + LOSS_FLAGS + [
        ('--num_steps_break', dict(default=500, type=int, help='optimiser iterations')),
        ('--beta', dict(default=1.e-4, type=float, help='total-variation weight')),
        ('--style_target_resize', dict(default=1.0, type=float, help='scale factor applied to the style image')),
        ('--cont_target_resize', dict(default=1.0, type=float, help='scale factor applied to the content image (= output size)')),
        ('--output_img_path', dict(default='./out.jpg', help='where the result is written')),
    ])


def stylize_webcam_parser():
    return _build("Filter a webcam feed (or a directory of frames) through a trained transform net.", [
        ('--model_path', dict(default='./models/starry_final.ckpt', help='checkpoint prefix of the trained net')),
        ('--upsample_method', dict(UPSAMPLE)),
        ('--resolution', dict(nargs=2, type=int, default=None, help='capture width height (default: the camera default)')),
        ('--frames_dir', dict(default=None, help='(addition) read frames from this directory instead of a camera')),
        ('--output_dir', dict(default='./frames_out', help='(addition) where --frames_dir results go')),
    ])
