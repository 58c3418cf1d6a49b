#!/usr/bin/env python
"""Drop-in for the reference's tfrecords_writer.py: converts a flat directory of images into sharded
TFRecord files of tf.train.Example protos (same flags, shard names and feature keys), without
TensorFlow -- framing/checksums come from libfaststyle_hip.so's host-side I/O entry points
(include/faststyle_io.h), image validation/PNG conversion from PIL.

  python tfrecords_writer.py --train_directory /path/to/train2014 --output_directory /path/to/out \\
      --train_shards 126 --num_threads 6

Output: <output_directory>/train-00000-of-00126 ... (tfrecords_writer.py:213-217 of the reference),
each record an Example with image/encoded (JPEG bytes, RGB), image/height, image/width,
image/colorspace 'RGB', image/channels 3, image/format 'JPEG', image/filename (reference :103-110).
"""
from __future__ import print_function

import argparse
import glob
import io
import os
import random
import sys
import threading
from datetime import datetime

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def setup_parser():
    parser = argparse.ArgumentParser(description="Convert an image directory to TFRecords.")
    parser.add_argument('--train_directory', default='/tmp/', help='Training data directory')
    parser.add_argument('--output_directory', default='/tmp/', help='Output data directory')
    parser.add_argument('--train_shards', default=2, type=int, help='Number of shards in training TFRecord files.')
    parser.add_argument('--num_threads', default=2, type=int, help='Number of threads to preprocess the images.')
    return parser


def _is_png(filename):
    return '.png' in filename                                       # reference :148-157


def _process_image(filename):
    """-> (jpeg bytes, height, width); PNGs are re-encoded as RGB JPEG quality 100 (reference :160-190)."""
    from PIL import Image
    with open(filename, 'rb') as f:
        image_data = f.read()
    if _is_png(filename):
        print('Converting PNG to JPEG for %s' % filename)
        buf = io.BytesIO()
        Image.open(io.BytesIO(image_data)).convert('RGB').save(buf, 'JPEG', quality=100)
        image_data = buf.getvalue()
    im = Image.open(io.BytesIO(image_data))
    im.load()                                                       # decode fully: a corrupt file fails here, as in TF
    width, height = im.size
    return image_data, height, width


def _convert_to_example(filename, image_buffer, height, width):
    from faststyle_amd import tfrecord
    return tfrecord.encode_example({
        'image/height': int(height),
        'image/width': int(width),
        'image/colorspace': b'RGB',
        'image/channels': 3,
        'image/format': b'JPEG',
        'image/filename': os.path.basename(filename).encode(),
        'image/encoded': image_buffer})


def _process_image_files_batch(thread_index, ranges, name, filenames, num_shards, output_directory):
    from faststyle_amd import tfrecord
    num_threads = len(ranges)
    assert not num_shards % num_threads                             # reference :204
    num_shards_per_batch = int(num_shards / num_threads)
    shard_ranges = np.linspace(ranges[thread_index][0], ranges[thread_index][1], num_shards_per_batch + 1).astype(int)
    counter = 0
    for s in range(num_shards_per_batch):
        shard = thread_index * num_shards_per_batch + s
        output_file = os.path.join(output_directory, '%s-%.5d-of-%.5d' % (name, shard, num_shards))
        shard_counter = 0
        with tfrecord.RecordWriter(output_file) as writer:
            for i in range(shard_ranges[s], shard_ranges[s + 1]):
                image_buffer, height, width = _process_image(filenames[i])
                writer.write(_convert_to_example(filenames[i], image_buffer, height, width))
                shard_counter += 1
                counter += 1
        print('%s [thread %d]: Wrote %d images to %s' % (datetime.now(), thread_index, shard_counter, output_file))
        sys.stdout.flush()


def _find_image_files(data_dir):
    print('Determining list of input files from %s.' % data_dir)
    filenames = sorted(glob.glob(data_dir + '/*'))
    shuffled_index = list(range(len(filenames)))
    random.seed(12345)                                              # reference :307-310
    random.shuffle(shuffled_index)
    filenames = [filenames[i] for i in shuffled_index]
    print('Found %d JPEG files inside %s.' % (len(filenames), data_dir))
    return filenames


def main(argv=None):
    args = setup_parser().parse_args(argv)
    assert not args.train_shards % args.num_threads, \
        'Please make the FLAGS.num_threads commensurate with FLAGS.train_shards'
    print('Saving results to %s' % args.output_directory)
    if not os.path.isdir(args.output_directory):
        os.makedirs(args.output_directory)
    filenames = _find_image_files(args.train_directory)
    spacing = np.linspace(0, len(filenames), args.num_threads + 1).astype(int)
    ranges = [[spacing[i], spacing[i + 1]] for i in range(len(spacing) - 1)]
    print('Launching %d threads for spacings: %s' % (args.num_threads, ranges))
    threads = []
    for thread_index in range(len(ranges)):
        t = threading.Thread(target=_process_image_files_batch,
                             args=(thread_index, ranges, 'train', filenames, args.train_shards, args.output_directory))
        t.start()
        threads.append(t)
    for t in threads:
        t.join()
    print('%s: Finished writing all %d images in data set.' % (datetime.now(), len(filenames)))


if __name__ == '__main__':
    main()
