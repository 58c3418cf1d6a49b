"""Input pipeline (reference datapipe.py, tfrecords_writer.py): native TFRecord /
Example code against published known answers and the independent oracle reader; the bicubic resize
kernel bit-exact against the numpy restatement; the batcher's shuffle_batch semantics end to end."""
import ctypes
import io
import os
import struct
import sys

import numpy as np
import pytest

from faststyle_amd import _lib, datapipe, tfrecord
from oracle import datapipe as odp
from tests.backends import engine_params, get_engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(params=engine_params())
def eng(request):
    return get_engine(request.param)


def lib():
    return _lib.load()


# ------------------------------------------------------------------ checksums / framing / Example (host code, no GPU)
def test_crc32c_known_answers():
    """Check values of the CRC-32C definition (RFC 3720 B.4) -- the same vectors leveldb/TensorFlow's crc32c_test uses."""
    L = lib()
    crc = lambda b: L.fs_crc32c(bytes(b), len(b))
    assert crc(b"\x00" * 32) == 0x8A9136AA
    assert crc(b"\xff" * 32) == 0x62A8AB43
    assert crc(bytes(range(32))) == 0x46DD794E
    assert crc(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert crc(b"123456789") == 0xE3069283
    assert crc(b"") == 0
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 63, 64, 65, 1000, 4099):             # hardware 8-byte steps + byte tail vs the table oracle
        buf = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert crc(buf) == odp.crc32c(buf)
        assert L.fs_crc32c_masked(buf, n) == odp.masked_crc32c(buf)


def test_record_framing_round_trips_with_the_independent_reader(tmp_path):
    rng = np.random.default_rng(1)
    payloads = [b"", b"a", rng.integers(0, 256, 1000, dtype=np.uint8).tobytes(), b"x" * 70000]
    p = str(tmp_path / "w.tfrecord")
    with tfrecord.RecordWriter(p) as w:
        for x in payloads:
            w.write(x)
    assert odp.read_tfrecord(p) == payloads                       # product writer -> oracle reader
    q = str(tmp_path / "o.tfrecord")
    open(q, "wb").write(b"".join(odp.frame_record(x) for x in payloads))
    assert open(p, "rb").read() == open(q, "rb").read()           # byte-identical files
    rf = tfrecord.RecordFile(q)                                    # oracle writer -> product reader
    assert len(rf) == 4 and [bytes(x) for x in rf] == payloads
    rf.close()
    # header layout known answer: length little-endian u64
    raw = open(p, "rb").read()
    assert struct.unpack_from("<Q", raw, 0)[0] == 0 and struct.unpack_from("<Q", raw, 16)[0] == 1
    empty = str(tmp_path / "e.tfrecord")
    open(empty, "wb").close()
    assert len(tfrecord.RecordFile(empty)) == 0


@pytest.mark.parametrize("damage,code", [("payload", -3), ("length", -2), ("truncate", -1), ("truncate_header", -1)])
def test_corrupt_records_are_rejected(tmp_path, damage, code):
    raw = bytearray(odp.frame_record(b"hello world") + odp.frame_record(b"second"))
    if damage == "payload":
        raw[14] ^= 1
    elif damage == "length":
        raw[0] ^= 1
    elif damage == "truncate":
        raw = raw[:-3]
    else:
        raw = raw[:len(odp.frame_record(b"hello world")) + 5]
    buf = bytes(raw)
    L = lib()
    assert L.fs_tfrecord_scan(buf, len(buf), 1, None, None, 0) == code
    assert b"fs_tfrecord_scan" in L.fs_last_error()
    p = str(tmp_path / "bad")
    open(p, "wb").write(buf)
    with pytest.raises(_lib.FaststyleError):
        tfrecord.RecordFile(p)
    with pytest.raises(ValueError):
        odp.read_tfrecord(p)


def test_example_proto_known_answer_and_lookup():
    # Example{features{feature{key:"a" value{int64_list{value:1}}}}} by the protobuf wire format, packed int64
    known = bytes.fromhex("0a0c0a0a0a0161120 51a030a0101".replace(" ", ""))
    assert tfrecord.encode_example({"a": 1}) == known
    assert odp.parse_example(known) == {"a": [1]}
    jpeg = bytes(range(256)) * 3
    ex = tfrecord.encode_example({"image/encoded": jpeg, "image/height": 480, "image/width": 640, "image/channels": 3,
                                  "image/colorspace": b"RGB", "image/format": b"JPEG", "image/filename": b"x.jpg"})
    want = odp.parse_example(ex)
    assert want["image/encoded"] == [jpeg] and want["image/height"] == [480] and want["image/filename"] == [b"x.jpg"]
    L = lib()
    off, ln, v = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_longlong()
    assert L.fs_example_bytes(ex, len(ex), b"image/encoded", ctypes.byref(off), ctypes.byref(ln)) == 0
    assert ex[off.value:off.value + ln.value] == jpeg
    for key, val in (("image/height", 480), ("image/width", 640), ("image/channels", 3)):
        assert L.fs_example_int64(ex, len(ex), key.encode(), ctypes.byref(v)) == 0 and v.value == val
    assert L.fs_example_int64(ex, len(ex), b"image/depth", ctypes.byref(v)) == -2          # absent key
    assert L.fs_example_int64(ex, len(ex), b"image/format", ctypes.byref(v)) == -3         # wrong kind
    assert L.fs_example_bytes(ex, len(ex), b"image/height", ctypes.byref(off), ctypes.byref(ln)) == -3
    assert L.fs_example_bytes(ex[:-5], len(ex) - 5, b"image/encoded", ctypes.byref(off), ctypes.byref(ln)) == -1
    # unpacked int64 (proto2 writers) and a negative value: key "n", value -2 as a 10-byte varint
    neg = b"\x08" + b"\xfe" + b"\xff" * 8 + b"\x01"
    feat = b"\x1a" + bytes([len(neg)]) + neg
    entry = b"\x0a\x01n" + b"\x12" + bytes([len(feat)]) + feat
    ex2 = b"\x0a" + bytes([len(entry) + 2]) + b"\x0a" + bytes([len(entry)]) + entry
    assert odp.parse_example(ex2) == {"n": [-2]}
    assert L.fs_example_int64(ex2, len(ex2), b"n", ctypes.byref(v)) == 0 and v.value == -2


# ------------------------------------------------------------------ TF1 bicubic resize
def test_oracle_bicubic_known_answers():
    """Hand-derived from the Keys kernel with a = -0.75 on TF's legacy grid (in = out * in_size/out_size):
    same size -> identity; x2 -> even outputs copy, odd outputs use (-3/32, 19/32, 19/32, -3/32) with clamped borders."""
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    assert np.array_equal(odp.resize_bicubic_tf1(img, 5, 7), img.astype(np.float32))
    row = rng.integers(0, 256, (1, 6, 1)).astype(np.float32)
    up = odp.resize_bicubic_tf1(row, 1, 12)[0, :, 0]
    x = row[0, :, 0]
    assert np.array_equal(up[0::2], x)
    xp = np.concatenate([x[:1], x, x[-1:], x[-1:]])               # clamp: index -1 -> 0, 6 and 7 -> 5
    want = np.array([(-3 * xp[j] + 19 * xp[j + 1] + 19 * xp[j + 2] - 3 * xp[j + 3]) / 32.0 for j in range(6)])
    np.testing.assert_allclose(up[1::2], want, rtol=0, atol=1e-4)
    flat = np.full((9, 11, 3), 77, np.uint8)
    np.testing.assert_allclose(odp.resize_bicubic_tf1(flat, 5, 17), 77.0, atol=2e-4)      # float32 table weights sum to 1 +- ulps
    # overshoot is kept (TF does not clip): a step edge rings below 0 / above 255
    edge = np.zeros((1, 8, 1), np.float32)
    edge[0, 4:, 0] = 255
    r = odp.resize_bicubic_tf1(edge, 1, 16)
    assert r.min() < -20 and r.max() > 275


@pytest.mark.parametrize("shape,out", [((37, 53), (16, 20)), ((19, 23), (40, 31)), ((64, 48), (64, 48)), ((5, 4), (9, 1))])
def test_resize_kernel_is_bit_exact_against_the_restatement(eng, shape, out):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    dst = eng.mem.empty(out + (3,))
    eng.resize_bicubic_u8(img, dst)
    got = eng.mem.to_numpy(dst)
    want = odp.resize_bicubic_tf1(img, out[0], out[1])
    assert np.array_equal(got, want)


def test_resize_kernel_reads_rgbx_pixels_as_the_decoder_stores_them(eng):
    """fs_resize_bicubic_u8x(pixel_bytes=4): the decode threads hand over PIL's own RGBX storage (no repack under the interpreter lock); the fourth
    byte is never read and the result is the packed form's, bit for bit -- hence the restatement's."""
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    rgbx = np.concatenate([img, rng.integers(0, 256, (37, 53, 1), dtype=np.uint8)], axis=2)     # garbage in the fourth byte
    a, b = eng.mem.empty((24, 40, 3)), eng.mem.empty((24, 40, 3))
    eng.resize_bicubic_u8(img, a)
    eng.resize_bicubic_u8(rgbx, b)
    assert np.array_equal(eng.mem.to_numpy(a), eng.mem.to_numpy(b))
    assert np.array_equal(eng.mem.to_numpy(b), odp.resize_bicubic_tf1(img, 24, 40))


def test_decode_jpeg_zero_copy_view_equals_the_packed_decode():
    from PIL import Image
    rng = np.random.default_rng(5)
    arr = (np.linspace(0, 255, 48)[None, :, None] * np.ones((31, 1, 3)) * rng.uniform(0.4, 1, (1, 1, 3))).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, "JPEG", quality=92)
    packed = datapipe.decode_jpeg(buf.getvalue())
    view = datapipe.decode_jpeg(buf.getvalue(), packed=False)
    assert packed.shape == (31, 48, 3) and view.dtype == np.uint8 and view.shape[:2] == (31, 48)
    assert view.shape[2] == (4 if datapipe._ARROW_OK else 3)
    assert np.array_equal(view[:, :, :3], packed)
    gray = io.BytesIO()
    Image.fromarray(arr[:, :, 0]).save(gray, "JPEG", quality=92)                       # channels=3 of a grayscale file: converted, as tf.image.decode_jpeg does
    assert np.array_equal(datapipe.decode_jpeg(gray.getvalue(), packed=False)[:, :, :3], datapipe.decode_jpeg(gray.getvalue()))


# ------------------------------------------------------------------ batcher
def make_shards(tmp_path, counts, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    files, images = [], []
    for s, n in enumerate(counts):
        p = str(tmp_path / ("train-%05d-of-%05d" % (s, len(counts))))
        with tfrecord.RecordWriter(p) as w:
            for k in range(n):
                h, wd = int(rng.integers(20, 40)), int(rng.integers(20, 40))
                # smooth content so JPEG is near-lossless enough to tell images apart
                arr = (rng.uniform(0, 255, (1, 1, 3)) * np.ones((h, wd, 1)) * np.linspace(0.3, 1, wd)[None, :, None]).astype(np.uint8)
                buf = io.BytesIO()
                Image.fromarray(arr).save(buf, "JPEG", quality=95)
                data = buf.getvalue()
                w.write(tfrecord.encode_example({"image/encoded": data, "image/height": h, "image/width": wd,
                                                 "image/channels": 3, "image/colorspace": b"RGB",
                                                 "image/format": b"JPEG", "image/filename": b"%d_%d.jpg" % (s, k)}))
                images.append(odp.resize_bicubic_tf1(datapipe.decode_jpeg(data), 16, 20))
        files.append(p)
    return files, images


def test_batcher_reproduces_shuffle_batch_semantics(eng, tmp_path):
    files, images = make_shards(tmp_path, [5, 4])
    assert datapipe.count_records(files) == 9
    batches = [eng.mem.to_numpy(b).copy() for b in
               datapipe.batcher(files, 2, (16, 20), num_epochs=2, min_after_dequeue=3, engine=eng, seed=5, num_threads=2)]
    assert len(batches) == 9 and all(b.shape == (2, 16, 20, 3) and b.dtype == np.float32 for b in batches)
    # every image is delivered exactly num_epochs times, bit-identical to decode -> restated resize
    seen = [0] * len(images)
    for b in batches:
        for img in b:
            hits = [i for i, ref in enumerate(images) if np.array_equal(ref, img)]
            assert len(hits) == 1
            seen[hits[0]] += 1
    assert seen == [2] * 9
    # it is a shuffle: the delivery order differs from the storage order, and depends on the seed
    order = [[i for i, ref in enumerate(images) if np.array_equal(ref, img)][0] for b in batches for img in b]
    assert order != sorted(order)
    # a trailing partial batch is dropped (allow_smaller_final_batch=False): 9 images, batch 2, one epoch -> 4 batches
    assert len(list(datapipe.batcher(files, 2, (16, 20), num_epochs=1, min_after_dequeue=3, engine=eng, num_threads=2))) == 4
    # max_batches stops an endless (num_epochs=None) producer
    assert len(list(datapipe.batcher(files, 3, (16, 20), num_epochs=None, min_after_dequeue=4, engine=eng,
                                     num_threads=2, max_batches=7))) == 7


def test_batcher_shards_by_rank_and_refuses_bad_input(eng, tmp_path):
    files, images = make_shards(tmp_path, [3, 2])
    got = []
    for rank in (0, 1):
        bs = list(datapipe.batcher(files, 1, (16, 20), num_epochs=1, min_after_dequeue=1, engine=eng, rank=rank, world=2,
                                   num_threads=1))
        got.append([[i for i, ref in enumerate(images) if np.array_equal(ref, eng.mem.to_numpy(b)[0])][0] for b in bs])
    assert sorted(got[0]) == [0, 1, 2] and sorted(got[1]) == [3, 4]        # rank r reads files[r::world]
    with pytest.raises(_lib.FaststyleError):
        next(datapipe.batcher(files, 1, None, engine=eng))                   # no static shape
    with pytest.raises(_lib.FaststyleError):
        next(datapipe.batcher(files, 1, (16, 20), engine=None))              # no CPU path
    with pytest.raises(_lib.FaststyleError):
        next(datapipe.batcher(files[:1], 1, (16, 20), engine=eng, rank=1, world=2))


def test_tfrecords_writer_cli_matches_reference_layout(tmp_path):
    from PIL import Image
    sys.path.insert(0, ROOT)
    import tfrecords_writer
    src = tmp_path / "imgs"
    src.mkdir()
    rng = np.random.default_rng(4)
    for k in range(5):
        Image.fromarray(rng.integers(0, 256, (24 + k, 30, 3), dtype=np.uint8)).save(str(src / ("im%d.jpg" % k)), quality=90)
    Image.fromarray(rng.integers(0, 256, (21, 17, 3), dtype=np.uint8)).save(str(src / "p.png"))
    out = tmp_path / "rec"
    tfrecords_writer.main(["--train_directory", str(src), "--output_directory", str(out), "--train_shards", "2",
                           "--num_threads", "2"])
    names = sorted(os.listdir(str(out)))
    assert names == ["train-00000-of-00002", "train-00001-of-00002"]         # reference tfrecords_writer.py:214
    recs = [odp.parse_example(r) for n in names for r in odp.read_tfrecord(str(out / n))]
    assert len(recs) == 6
    for r in recs:
        assert sorted(r) == ["image/channels", "image/colorspace", "image/encoded", "image/filename", "image/format",
                             "image/height", "image/width"]
        im = datapipe.decode_jpeg(r["image/encoded"][0])
        assert im.shape == (r["image/height"][0], r["image/width"][0], 3) and r["image/channels"] == [3]
        assert r["image/colorspace"] == [b"RGB"] and r["image/format"] == [b"JPEG"]
    assert sorted(r["image/filename"][0] for r in recs) == [b"im0.jpg", b"im1.jpg", b"im2.jpg", b"im3.jpg", b"im4.jpg", b"p.png"]


@pytest.mark.gpu
def test_train_cli_reads_tfrecord_shards(tmp_path, monkeypatch, capsys):
    """train.py --train_dir <dir of train-* shards>: the reference's own input format, end to end on the GPU."""
    import json
    from faststyle_amd import vgg16
    sys.path.insert(0, ROOT)
    import train
    files, _ = make_shards(tmp_path, [7, 6])
    work = tmp_path / "w"
    (work / "libs").mkdir(parents=True)
    np.savez(str(work / "libs" / "vgg16_weights.npz"), **vgg16.synthetic_weights(3))
    monkeypatch.chdir(work)
    train.main(train.setup_parser().parse_args(
        ["--train_dir", str(tmp_path), "--model_name", "r", "--style_img_path",
         os.path.join(ROOT, "style_images", "starry_night_crop.jpg"), "--style_target_resize", "0.25",
         "--preprocess_size", "64", "64", "--batch_size", "2", "--n_epochs", "2", "--num_pipe_buffer", "5",
         "--num_steps_ckpt", "100"]))
    out = [l for l in capsys.readouterr().out.splitlines() if l and "amdgpu" not in l]
    assert out[-1] == "Done training."
    logs = [json.loads(l) for l in open(str(work / "summaries" / "train" / "r0" / "scalars.jsonl"))]
    assert [d["step"] for d in logs] == [0, 10] and all(np.isfinite(d["loss"]) for d in logs)   # 26 images / 2 = 13 steps


def test_tensorboard_event_file_layout(tmp_path):
    """Event files are TFRecord files of tensorflow.Event protos (event.proto / summary.proto field numbers)."""
    from faststyle_amd import tbevents
    w = tbevents.EventWriter(str(tmp_path / "run"))
    w.add_scalars(0, [("summaries/loss", 3.5), ("summaries/style_loss", 1.25)])
    w.add_scalars(10, [("summaries/loss", 2.0)])
    w.close()
    assert os.path.basename(w.path).startswith("events.out.tfevents.")
    recs = odp.read_tfrecord(w.path)
    assert len(recs) == 3
    # known answer by the wire format: step=10 -> 10 0a ; summary(5){value(1){tag(1) "summaries/loss", simple_value(2)=2.0f}}
    tail = b"\x10\x0a" + b"\x2a\x17" + b"\x0a\x15" + b"\x0a\x0esummaries/loss" + b"\x15" + struct.pack("<f", 2.0)
    assert recs[2][9:] == tail and recs[2][0] == 0x09
    first = dict((f, v) for f, _, v in odp._fields(recs[0]))
    assert first[3] == b"brain.Event:2" and abs(struct.unpack("<d", first[1])[0] - __import__("time").time()) < 600
    ev = dict((f, v) for f, _, v in odp._fields(recs[1]))
    assert 2 not in ev or ev[2] == 0                      # step 0 (varint 0 is written explicitly)
    vals = [dict((f, v) for f, _, v in odp._fields(x)) for f0, _, x in odp._fields(ev[5]) if f0 == 1]
    assert [v[1] for v in vals] == [b"summaries/loss", b"summaries/style_loss"]
    assert [struct.unpack("<f", v[2])[0] for v in vals] == [3.5, 1.25]
