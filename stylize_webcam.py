#!/usr/bin/env python
"""Drop-in for the reference's stylize_webcam.py (same flags): filters an OpenCV webcam feed through
a trained model and writes output.avi, with the network on the MI355X (faststyle_amd/stream.py).

OpenCV (camera capture, window, XVID writer) is a host-side dependency of the reference that this
image does not ship; without it the script can still filter a *directory of frames*:
    python stylize_webcam.py --model_path models/starry_final.ckpt --frames_dir in/ --output_dir out/
(frames are read RGB with PIL and converted to the BGR order a cv2 capture would deliver, so the
reference's channel handling -- BGR fed as is, output swapped -- is reproduced bit for bit).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def setup_parser():
    """The reference flag surface (stylize_webcam.py:17-39) plus --frames_dir/--output_dir, in faststyle_amd/cli.py."""
    from faststyle_amd import cli
    return cli.stylize_webcam_parser()


def _load(args):
    from faststyle_amd import ckpt, engine
    eng = engine.Engine()
    variables = eng.mem.from_numpy(eng.flatten_params(ckpt.load_checkpoint(args.model_path),
                                                      upsample_method=args.upsample_method))
    return eng, variables


def run_frames_dir(args):
    from PIL import Image
    from faststyle_amd import stream
    names = sorted(f for f in os.listdir(args.frames_dir) if f.lower().endswith(('.jpg', '.jpeg', '.png')))
    if not names:
        raise SystemExit('no frames under %s' % args.frames_dir)
    eng, variables = _load(args)
    if not os.path.isdir(args.output_dir):
        os.makedirs(args.output_dir)
    # On the GPU the frames of a directory are independent work: two of them in flight on two streams (stream.PipelinedStylizer: 720p 1277 -> 1890 frames/s;
    # FS_FRAMES_IN_FLIGHT=1 for one at a time).  The results -- and their order -- are the same.
    import collections
    depth = int(os.environ.get('FS_FRAMES_IN_FLIGHT', '2')) if hasattr(eng.mem, 'torch') else 1
    st = None
    pend = collections.deque()

    def save(n, img_out):
        # cv2.imshow / VideoWriter interpret that array as BGR; save exactly what they would show
        Image.fromarray(img_out[:, :, ::-1]).save(os.path.join(args.output_dir, os.path.splitext(n)[0] + '.png'))

    for n in names:
        im = Image.open(os.path.join(args.frames_dir, n)).convert('RGB')
        if args.resolution is not None:
            im = im.resize(tuple(args.resolution))
        frame = np.asarray(im, np.uint8)[:, :, ::-1]              # what cap.read() returns: BGR
        if st is None or shape != frame.shape[:2]:
            while pend:
                save(pend.popleft(), st.fetch())
            shape = frame.shape[:2]
            print('Resolution is: {0} by {1}'.format(frame.shape[1], frame.shape[0]))
            if depth > 1:
                st = stream.PipelinedStylizer(eng, variables, frame.shape[0], frame.shape[1], depth=depth, upsample_method=args.upsample_method)
            else:
                st = stream.FrameStylizer(eng, variables, frame.shape[0], frame.shape[1], args.upsample_method)
        if depth > 1:
            if len(pend) >= depth:
                save(pend.popleft(), st.fetch())
            st.submit(np.ascontiguousarray(frame))
            pend.append(n)
        else:
            save(n, st(np.ascontiguousarray(frame)))                 # = cvtColor(astype(uint8)(Y), BGR2RGB)
    while pend:
        save(pend.popleft(), st.fetch())


def run_webcam(args):
    try:
        import cv2
    except ImportError:
        raise SystemExit('stylize_webcam.py needs OpenCV (cv2) for camera capture and display, as the reference does; '
                         'it is not installed here.  Use --frames_dir to filter a directory of frames.')
    from faststyle_amd import stream
    cap = cv2.VideoCapture(0)
    if args.resolution is not None:
        x_length, y_length = args.resolution
        cap.set(cv2.CAP_PROP_FRAME_WIDTH, x_length)      # (the reference passes the raw property ids 3 and 4: stylize_webcam.py:52-53)
        cap.set(cv2.CAP_PROP_FRAME_HEIGHT, y_length)
    x_new, y_new = int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)), int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT))
    print('Resolution is: {0} by {1}'.format(x_new, y_new))
    print('Loading up model...')
    eng, variables = _load(args)
    st = stream.FrameStylizer(eng, variables, y_new, x_new, args.upsample_method)
    fourcc = cv2.VideoWriter_fourcc(*'XVID')
    out = cv2.VideoWriter('output.avi', fourcc, 15.0, (x_new, y_new))
    print('Begin filtering...')
    while True:
        ret, frame = cap.read()
        img_out = st(np.ascontiguousarray(frame))
        out.write(img_out)
        cv2.imshow('frame', img_out)
        if cv2.waitKey(1) & 0xFF == ord('q'):
            break
    cap.release()
    out.release()
    cv2.destroyAllWindows()


if __name__ == '__main__':
    args = setup_parser().parse_args()
    if args.frames_dir is not None:
        run_frames_dir(args)
    else:
        run_webcam(args)
