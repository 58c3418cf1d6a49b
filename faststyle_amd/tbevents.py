"""TensorBoard event files without TensorFlow: what ``tf.summary.FileWriter.add_summary`` writes for the
four scalars of the reference's train.py (:185-189, :213-217, :260-267).

An event file is a TFRecord file (framing by the native library, faststyle_amd/tfrecord.py) of
``tensorflow.Event`` protos: the first record carries ``file_version = "brain.Event:2"``, every
later one ``{wall_time, step, summary{value{tag, simple_value}...}}``.  Field numbers follow
tensorflow/core/util/event.proto and tensorflow/core/framework/summary.proto.
"""
import os
import socket
import struct
import time

from . import tfrecord


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_event(wall_time, step=None, file_version=None, scalars=None):
    ev = b"\x09" + struct.pack("<d", wall_time)                       # 1: wall_time (double)
    if step is not None:
        ev += b"\x10" + _varint(int(step))                            # 2: step (int64)
    if file_version is not None:
        ev += _ld(3, file_version.encode())                           # 3: file_version
    if scalars:
        summary = b""
        for tag, value in scalars:                                    # Summary.value = 1; Value.tag = 1, simple_value = 2 (float)
            summary += _ld(1, _ld(1, tag.encode()) + b"\x15" + struct.pack("<f", float(value)))
        ev += _ld(5, summary)                                         # 5: summary
    return ev


class EventWriter(object):
    """tf.summary.FileWriter(logdir): events.out.tfevents.<unix time>.<hostname> inside ``logdir``."""

    def __init__(self, logdir):
        if not os.path.isdir(logdir):
            os.makedirs(logdir)
        now = time.time()
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s" % (int(now), socket.gethostname()))
        self._w = tfrecord.RecordWriter(self.path)
        self._w.write(encode_event(now, file_version="brain.Event:2"))
        self._w._f.flush()

    def add_scalars(self, step, scalars):
        """scalars: list of (tag, value) -- one Summary with all of them, like the merged summary op."""
        self._w.write(encode_event(time.time(), step=step, scalars=scalars))
        self._w._f.flush()

    def close(self):
        self._w.close()
