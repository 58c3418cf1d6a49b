"""TensorFlow "bundle V2" checkpoint reader/writer (no TensorFlow needed).

The reference stores and restores the image-transform net with ``tf.train.Saver``
(reference train.py:224-225,286 writes; stylize_image.py:68,73 restores).  The on-disk
format is therefore part of the drop-in contract:

* ``<prefix>.data-00000-of-00001`` -- raw little-endian tensors, back to back, in
  sorted-key order.
* ``<prefix>.index`` -- a LevelDB-style SSTable: one uncompressed data block with
  prefix-compressed entries (restart interval 16), an empty metaindex block, an index
  block with one entry, and a 48-byte footer.  The first entry (empty key) is a
  ``BundleHeaderProto``; every other entry is a ``BundleEntryProto`` carrying dtype,
  shape, offset, size and a masked CRC32C of the tensor bytes.

The writer reproduces the shipped ``models/*_final.ckpt.index`` byte for byte when given
the same tensors (tests/test_ckpt.py), so TF1's ``saver.restore`` accepts its output.
"""
import os
import struct
from collections import OrderedDict

import numpy as np

_MAGIC = bytes.fromhex("57fb808b247547db")
_RESTART_INTERVAL = 16
DT_FLOAT = 1
DT_INT32 = 3
DT_INT64 = 9
_NP_OF_DT = {DT_FLOAT: np.dtype("<f4"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}
_DT_OF_NP = {v: k for k, v in _NP_OF_DT.items()}


# --------------------------------------------------------------------------- crc32c
def _make_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_TAB = _make_table()
_TAB_NP = np.array(_TAB, dtype=np.uint32)


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), the checksum LevelDB/TF bundle files use."""
    c = crc ^ 0xFFFFFFFF
    tab = _TAB
    for b in bytes(data):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# --------------------------------------------------------------------------- varints / protobuf
def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _get_varint(buf, pos):
    shift = 0
    val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _parse_fields(buf):
    """Yield (field_number, wire_type, value) for one protobuf message."""
    pos = 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0}
    for fn, _, v in _parse_fields(buf):
        if fn == 1:
            e["dtype"] = v
        elif fn == 2:
            for fn2, _, v2 in _parse_fields(v):
                if fn2 == 2:  # dim
                    size = 0
                    for fn3, _, v3 in _parse_fields(v2):
                        if fn3 == 1:
                            size = v3
                    e["shape"].append(size)
        elif fn == 3:
            e["shard_id"] = v
        elif fn == 4:
            e["offset"] = v
        elif fn == 5:
            e["size"] = v
        elif fn == 6:
            e["crc32c"] = v
    return e


def _encode_entry(dtype, shape, offset, size, crc):
    shp = b"".join(b"\x12" + _put_varint(len(d)) + d
                   for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08" + _put_varint(dtype) + b"\x12" + _put_varint(len(shp)) + shp
    if offset:  # proto3: zero-valued scalars are omitted
        out += b"\x20" + _put_varint(offset)
    if size:
        out += b"\x28" + _put_varint(size)
    out += b"\x35" + struct.pack("<I", crc)
    return out


_HEADER = b"\x08\x01\x1a\x02\x08\x01"  # num_shards=1, version{producer=1}


# --------------------------------------------------------------------------- SSTable blocks
def _read_block(buf, offset, size, verify):
    body = buf[offset:offset + size]
    trailer = buf[offset + size:offset + size + 5]
    if trailer[0] != 0:
        raise ValueError("compressed SSTable blocks are not supported")
    if verify:
        want = struct.unpack("<I", trailer[1:5])[0]
        if masked_crc32c(body + trailer[:1]) != want:
            raise ValueError("SSTable block checksum mismatch")
    n_restarts = struct.unpack_from("<I", body, len(body) - 4)[0]
    end = len(body) - 4 - 4 * n_restarts
    entries = []
    pos = 0
    key = b""
    while pos < end:
        shared, pos = _get_varint(body, pos)
        non_shared, pos = _get_varint(body, pos)
        vlen, pos = _get_varint(body, pos)
        key = key[:shared] + body[pos:pos + non_shared]
        pos += non_shared
        entries.append((key, body[pos:pos + vlen]))
        pos += vlen
    return entries


def _build_block(entries):
    out = bytearray()
    restarts = []
    prev = b""
    for i, (key, val) in enumerate(entries):
        if i % _RESTART_INTERVAL == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            lim = min(len(prev), len(key))
            while shared < lim and prev[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val))
        out += key[shared:] + val
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _with_trailer(block):
    return block + b"\x00" + struct.pack("<I", masked_crc32c(block + b"\x00"))


def _short_successor(key):
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


# --------------------------------------------------------------------------- public API
def read_index(prefix, verify=True):
    """Return an OrderedDict name -> entry dict (dtype, shape, offset, size, crc32c)."""
    buf = open(prefix + ".index", "rb").read()
    if buf[-8:] != _MAGIC:
        raise ValueError("%s.index: bad SSTable magic" % prefix)
    footer = buf[-48:]
    _, p = _get_varint(footer, 0)          # metaindex offset
    _, p = _get_varint(footer, p)          # metaindex size
    idx_off, p = _get_varint(footer, p)
    idx_size, p = _get_varint(footer, p)
    out = OrderedDict()
    for _, handle in _read_block(buf, idx_off, idx_size, verify):
        off, p = _get_varint(handle, 0)
        size, p = _get_varint(handle, p)
        for key, val in _read_block(buf, off, size, verify):
            if key == b"":
                continue  # BundleHeaderProto
            out[key.decode()] = _parse_entry(val)
    return out


def load_checkpoint(prefix, verify=True):
    """Read every tensor of a bundle-V2 checkpoint -> OrderedDict name -> ndarray.

    Mirrors what ``saver.restore(sess, model_path)`` (reference stylize_image.py:73)
    makes available: ``prefix`` is the path without ``.index``/``.data-…`` suffix.
    """
    index = read_index(prefix, verify)
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    out = OrderedDict()
    for name, e in index.items():
        if e["shard_id"] != 0:
            raise ValueError("multi-shard bundles are not supported")
        raw = data[e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise ValueError("%s: tensor %s is truncated" % (prefix, name))
        if verify and masked_crc32c(raw) != e["crc32c"]:
            raise ValueError("%s: tensor %s fails its CRC32C" % (prefix, name))
        dt = _NP_OF_DT[e["dtype"]]
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()
    return out


def save_checkpoint(prefix, tensors):
    """Write ``tensors`` (name -> ndarray) as ``prefix.index`` + ``prefix.data-00000-of-00001``.

    Same role as ``final_saver.save(sess, 'models/<name>_final.ckpt')`` (reference
    train.py:286).  Keys are stored sorted, tensors contiguous, exactly as TF does.
    """
    d = os.path.dirname(prefix)
    if d and not os.path.isdir(d):
        os.makedirs(d)
    entries = [(b"", _HEADER)]
    blob = bytearray()
    for name in sorted(tensors):
        arr = np.asarray(tensors[name])
        if not arr.flags.c_contiguous:
            arr = arr.copy()
        dt = arr.dtype.newbyteorder("<") if arr.dtype.byteorder == ">" else arr.dtype
        arr = arr.astype(dt, copy=False)
        code = _DT_OF_NP[np.dtype(arr.dtype.str.replace("=", "<").replace("|", "<"))]
        raw = arr.tobytes()
        entries.append((name.encode(),
                        _encode_entry(code, arr.shape, len(blob), len(raw), masked_crc32c(raw))))
        blob += raw
    data_block = _build_block(entries)
    out = bytearray(_with_trailer(data_block))
    meta_off = len(out)
    meta_block = _build_block([])
    out += _with_trailer(meta_block)
    idx_off = len(out)
    handle = _put_varint(0) + _put_varint(len(data_block))
    idx_block = _build_block([(_short_successor(entries[-1][0]), handle)])
    out += _with_trailer(idx_block)
    footer = (_put_varint(meta_off) + _put_varint(len(meta_block)) +
              _put_varint(idx_off) + _put_varint(len(idx_block)))
    footer += b"\x00" * (40 - len(footer)) + _MAGIC
    out += footer
    # both files are written to "<file>.tmp" and fsynced FIRST; only then are they renamed into place back to back, the
    # data file before the index (the file a reader opens first; TF's BundleWriter order), and the directory is fsynced.
    # Re-saving over an existing prefix (`<model>_final.ckpt`, the same `ckpt-N` after a resume) therefore has no long
    # window in which new data sits behind the old index (same offsets, stale CRCs); a crash before the renames leaves
    # the old bundle intact, a crash between them is caught by the per-tensor CRC check of load_checkpoint.
    files = ((".data-00000-of-00001", bytes(blob)), (".index", bytes(out)))
    for suffix, payload in files:
        with open(prefix + suffix + ".tmp", "wb") as f:
            f.write(payload)
            f.flush()
            os.fsync(f.fileno())
    for suffix, _ in files:
        os.replace(prefix + suffix + ".tmp", prefix + suffix)
    try:
        dfd = os.open(os.path.dirname(os.path.abspath(prefix)), os.O_RDONLY)
        try:
            os.fsync(dfd)
        finally:
            os.close(dfd)
    except OSError:          # (a file system without directory fsync)
        pass
