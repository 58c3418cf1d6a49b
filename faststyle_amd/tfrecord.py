"""TFRecord files of tf.train.Example protos -- the training-set format of the reference
(tfrecords_writer.py:90-114 writes them, datapipe.py:38-49 reads them).

Framing, checksums and the Example lookup run in the native library (csrc/fs_io.hip, host C++:
hardware CRC-32C, zero-copy over an mmap of the shard); this module only does the Python plumbing
and the (cold-path) Example *encoder* used by tfrecords_writer.py.
"""
import ctypes
import mmap
import os

from . import _lib as L


def _lib():
    return L.load()


def masked_crc32c(data):
    buf = bytes(data)
    return int(_lib().fs_crc32c_masked(buf, len(buf)))


class RecordFile(object):
    """One shard, memory-mapped: ``len()``, ``payload(i)`` (zero-copy memoryview), iteration."""

    def __init__(self, path, verify_crc=True):
        self.path = path
        self._f = open(path, "rb")
        size = os.fstat(self._f.fileno()).st_size
        if size == 0:
            self._mm, self.n = None, 0
            self.off = self.len = ()
            return
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self._view = memoryview(self._mm)
        # a read-only mmap cannot be handed to ctypes.from_buffer; numpy exposes its address without a copy
        import numpy as np
        self._np = np.frombuffer(self._mm, dtype=np.uint8)
        self._addr = self._np.ctypes.data
        lib = _lib()
        n = lib.fs_tfrecord_scan(self._addr, size, 0, None, None, 0)
        off = ln = None
        rc = n
        if n >= 0:
            off = (ctypes.c_uint64 * n)()
            ln = (ctypes.c_uint64 * n)()
            rc = lib.fs_tfrecord_scan(self._addr, size, 1 if verify_crc else 0, off, ln, n)
        if rc < 0:
            msg = lib.fs_last_error().decode()
            self.close()                       # do not leak the mapping / descriptor of a corrupt shard
            raise L.FaststyleError("%s: %s" % (path, msg))
        self.n, self.off, self.len = int(n), off, ln

    def __len__(self):
        return self.n

    def payload(self, i):
        o = int(self.off[i])
        return self._view[o:o + int(self.len[i])]

    def feature_bytes(self, i, key):
        """bytes_list.value[0] of feature ``key`` of record i (tf.FixedLenFeature([], tf.string))."""
        o, n = int(self.off[i]), int(self.len[i])
        off, ln = ctypes.c_uint64(), ctypes.c_uint64()
        lib = _lib()
        if lib.fs_example_bytes(self._addr + o, n, key.encode(), ctypes.byref(off), ctypes.byref(ln)):
            raise L.FaststyleError("%s record %d: %s" % (self.path, i, lib.fs_last_error().decode()))
        return self._view[o + off.value:o + off.value + ln.value]

    def feature_int64(self, i, key):
        o, n = int(self.off[i]), int(self.len[i])
        v = ctypes.c_longlong()
        lib = _lib()
        if lib.fs_example_int64(self._addr + o, n, key.encode(), ctypes.byref(v)):
            raise L.FaststyleError("%s record %d: %s" % (self.path, i, lib.fs_last_error().decode()))
        return v.value

    def __iter__(self):
        for i in range(self.n):
            yield self.payload(i)

    def close(self):
        if self._mm is not None:
            self._view.release()
            self._np = None
            try:
                self._mm.close()
            except BufferError:
                pass
            self._mm = None
        self._f.close()


class RecordWriter(object):
    """tf.python_io.TFRecordWriter (tfrecords_writer.py:217, 230)."""

    def __init__(self, path):
        self._f = open(path, "wb")

    def write(self, payload):
        payload = bytes(payload)
        out = ctypes.create_string_buffer(len(payload) + 16)
        n = _lib().fs_tfrecord_frame(payload, len(payload), out)
        self._f.write(out.raw[:n])

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ------------------------------------------------------------------ Example encoder (writer side only)
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


def encode_example(features):
    """dict name -> bytes | int | list of those  ->  serialized tf.train.Example
    (Example{features=1}, Features{map feature=1}, Feature{bytes_list=1 | int64_list=3};
    int64 lists are packed, map entries are written in sorted key order like protobuf's
    deterministic serialisation)."""
    entries = b""
    for key in sorted(features):
        v = features[key]
        vals = v if isinstance(v, (list, tuple)) else [v]
        if all(isinstance(x, (bytes, bytearray, memoryview)) for x in vals):
            feat = _ld(1, b"".join(_ld(1, bytes(x)) for x in vals))
        elif all(isinstance(x, int) for x in vals):
            feat = _ld(3, _ld(1, b"".join(_varint(x) for x in vals)))
        else:
            raise TypeError("feature %r: only bytes and int64 features are used by the reference" % key)
        entries += _ld(1, _ld(1, key.encode()) + _ld(2, feat))
    return _ld(1, entries)
