"""TEST INFRASTRUCTURE -- CPU restatement of the reference's input pipeline pieces (datapipe.py).

Only tests/ may import this.  Independent of the product code: its own CRC table, record walker,
protobuf reader and a numpy float32 statement of TF 1.0's bicubic resize.

Sources restated (TensorFlow 1.0.0 is an un-vendored dependency of the reference, README.md:23;
it cannot be installed here, so the resize is "parity unpinned" -- restated from the published
kernel, pinned only by hand-derived known answers in tests/test_datapipe.py):
  * TFRecord framing: tensorflow/core/lib/io/record_writer.cc -- uint64 length, uint32
    masked_crc32c(length), data, uint32 masked_crc32c(data); mask = rotr(crc,15) + 0xa282ead8
  * tf.train.Example wire format: tensorflow/core/example/{example,feature}.proto
  * resize_images(method=2) (reference datapipe.py:24) -> ResizeBicubic, align_corners=False:
    tensorflow/core/kernels/resize_bicubic_op.cc @ r1.0
"""
import struct

import numpy as np


# ------------------------------------------------------------------ crc32c / framing
def _table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_TAB = _table()


def crc32c(data):
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = _TAB[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def read_tfrecord(path):
    """-> list of payload bytes; raises ValueError on a checksum mismatch or truncation."""
    buf = open(path, "rb").read()
    pos, out = 0, []
    while pos < len(buf):
        if len(buf) - pos < 12:
            raise ValueError("truncated header")
        (n,), (c,) = struct.unpack_from("<Q", buf, pos), struct.unpack_from("<I", buf, pos + 8)
        if masked_crc32c(buf[pos:pos + 8]) != c:
            raise ValueError("length crc")
        data = buf[pos + 12:pos + 12 + n]
        if len(data) != n or len(buf) < pos + 16 + n:
            raise ValueError("truncated payload")
        (c,) = struct.unpack_from("<I", buf, pos + 12 + n)
        if masked_crc32c(data) != c:
            raise ValueError("payload crc")
        out.append(data)
        pos += 16 + n
    return out


def frame_record(payload):
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", masked_crc32c(head)) + payload + struct.pack("<I", masked_crc32c(payload))


# ------------------------------------------------------------------ protobuf (reader only)
def _varint(buf, pos):
    v, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _fields(buf):
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("wire type %d" % wt)
        yield f, wt, v


def parse_example(buf):
    """serialized tf.train.Example -> {key: [bytes...] | [int...]} (float lists are not used by the path)."""
    out = {}
    for f, _, feats in _fields(bytes(buf)):
        if f != 1:
            continue
        for f2, _, entry in _fields(feats):
            if f2 != 1:
                continue
            key, feature = None, b""
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = v.decode()
                elif f3 == 2:
                    feature = v
            vals = []
            for kind, _, lst in _fields(feature):
                for f5, wt, v in _fields(lst):
                    if f5 != 1:
                        continue
                    if kind == 1:
                        vals.append(bytes(v))
                    elif kind == 3 and wt == 0:
                        vals.append(v - (1 << 64) if v >> 63 else v)
                    elif kind == 3 and wt == 2:
                        p = 0
                        while p < len(v):
                            x, p = _varint(v, p)
                            vals.append(x - (1 << 64) if x >> 63 else x)
            out[key] = vals
    return out


# ------------------------------------------------------------------ TF 1.0 bicubic resize
K_TABLE = 1024


def _coeffs_table():
    """GetCoeffsTable(): float arithmetic, A = -0.75; entry 2i for |x| <= 1, entry 2i+1 for x+1."""
    f = np.float32
    a = f(-0.75)
    tab = np.zeros((K_TABLE + 1) * 2, dtype=np.float32)
    for i in range(K_TABLE + 1):
        x = f(i * 1.0 / K_TABLE)
        tab[2 * i] = f(f(f(f(f(f(a + f(2)) * x) - f(a + f(3))) * x) * x) + f(1))
        x = f(x + f(1.0))
        tab[2 * i + 1] = f(f(f(f(f(f(f(a * x) - f(f(5) * a)) * x) + f(f(8) * a)) * x)) - f(f(4) * a))
    return tab


_COEFFS = _coeffs_table()


def _weights_indices(scale, out_loc, limit):
    """GetWeightsAndIndices(scale, out_loc, limit) for a vector of out_loc."""
    f = np.float32
    in_f = (f(scale) * out_loc.astype(np.float32)).astype(np.float32)
    in_loc = in_f.astype(np.int64)                                  # const int64 in_loc = scale * out_loc
    delta = (in_f - in_loc.astype(np.float32)).astype(np.float32)
    offset = np.rint((delta * f(K_TABLE)).astype(np.float32)).astype(np.int64)    # lrintf: round half to even
    w = np.stack([_COEFFS[offset * 2 + 1], _COEFFS[offset * 2], _COEFFS[(K_TABLE - offset) * 2],
                  _COEFFS[(K_TABLE - offset) * 2 + 1]], axis=-1).astype(np.float32)
    idx = np.stack([np.clip(in_loc + k, 0, limit - 1) for k in (-1, 0, 1, 2)], axis=-1)
    return w, idx


def _interp(w, v):
    """Interpolate1D: v0*w0 + v1*w1 + v2*w2 + v3*w3, float32, left to right."""
    f = np.float32
    acc = (v[..., 0] * w[..., 0]).astype(f)
    for k in (1, 2, 3):
        acc = (acc + (v[..., k] * w[..., k]).astype(f)).astype(f)
    return acc


def resize_bicubic_tf1(img, Ho, Wo):
    """img uint8/float [H,W,C] -> float32 [Ho,Wo,C]; rows are interpolated in x first, then in y
    (ResizeBicubicOp::Compute); the result is not clipped."""
    img = np.asarray(img).astype(np.float32)
    H, W, C = img.shape
    hs = np.float32(H) / np.float32(Ho)                             # CalculateResizeScale, align_corners=False
    ws = np.float32(W) / np.float32(Wo)
    wy, iy = _weights_indices(hs, np.arange(Ho), H)
    wx, ix = _weights_indices(ws, np.arange(Wo), W)
    rows = img[iy]                                                  # [Ho,4,W,C]
    taps = rows[:, :, ix, :]                                        # [Ho,4,Wo,4,C]
    # x pass: for every (oy, r, ox, c): sum_k taps[oy,r,ox,k,c] * wx[ox,k]
    t = np.moveaxis(taps, 3, -1)                                    # [Ho,4,Wo,C,4]
    col = _interp(wx[None, None, :, None, :], t)                    # [Ho,4,Wo,C]
    # y pass: sum_r col[oy,r,ox,c] * wy[oy,r]
    c2 = np.moveaxis(col, 1, -1)                                    # [Ho,Wo,C,4]
    return _interp(wy[:, None, None, :], c2)
