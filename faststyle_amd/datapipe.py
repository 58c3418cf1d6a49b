"""Input pipeline of train.py (drop-in for the reference's datapipe.py): TFRecord shards of
tf.train.Example protos -> JPEG decode -> TF1 bicubic resize -> shuffle queue -> batches.

Where each stage runs:
  * shard reading / record framing / Example lookup: native host code over an mmap (csrc/fs_io.hip);
  * JPEG entropy decode: libjpeg via PIL on ``num_threads`` host threads (PIL drops the GIL), running
    ahead of the training loop through a bounded prefetch window;
  * resize (tf.image.resize_images(method=2), datapipe.py:24): HIP kernel, fed with the u8 pixels
    (a quarter of the fp32 bytes over PCIe), writing straight into
  * the shuffle queue (tf.train.shuffle_batch, datapipe.py:74-77): ONE [capacity,H,W,3] fp32 tensor
    resident in HBM (4000 x 256x256 images = 3.1 GB of the 288 GB); a batch is a device-side gather.

Data-parallel: rank r reads shards ``files[r::world]`` (no exchange).
"""
import io
import os
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib as L
from . import tfrecord

FEATURE_KEYS = ("image/encoded", "image/height", "image/channels", "image/width")     # datapipe.py:42-45


def decode_jpeg(data, packed=True):
    """tf.image.decode_jpeg(contents, channels=3) (datapipe.py:46): uint8 [H,W,3] RGB.
    packed=False (the batcher's decode threads): the decoder's own RGBX storage [H,W,4] as a zero-copy view (Arrow C data interface of
    Pillow >= 11.2 + pyarrow) when both are there -- np.asarray(im) repacks RGBX -> RGB through im.tobytes() under the interpreter lock, 0.3 ms
    per 640 x 480 image and the largest serialised piece of a decode thread's work; fs_resize_bicubic_u8x reads the RGBX pixels as they are."""
    from PIL import Image
    im = Image.open(io.BytesIO(bytes(data)))
    if im.mode != "RGB":
        im = im.convert("RGB")
    if not packed and _ARROW_OK:
        im.load()
        try:
            flat = _pa.array(im).flatten().to_numpy(zero_copy_only=True)     # (keeps the image's memory alive through the Arrow buffer)
            if flat.size == im.height * im.width * 4:
                return flat.reshape(im.height, im.width, 4)
        except Exception:
            pass
    return np.asarray(im, dtype=np.uint8)


try:
    import pyarrow as _pa
    from PIL import Image as _Image
    _ARROW_OK = hasattr(_Image.Image, "__arrow_c_array__") and os.environ.get("FS_DATAPIPE_RGBX", "1") != "0"
except Exception:      # (no pyarrow / an older Pillow: the packed path)
    _pa = None
    _ARROW_OK = False


def count_records(filenames):
    n = 0
    for f in filenames:
        rf = tfrecord.RecordFile(f, verify_crc=False)
        n += len(rf)
        rf.close()
    return n


def _examples(files, num_epochs, rng):
    """tf.train.string_input_producer(files, num_epochs, shuffle=True) + TFRecordReader +
    parse_single_example (datapipe.py:38-46): yields the 'image/encoded' bytes, shard order
    reshuffled every epoch."""
    epoch = 0
    while num_epochs is None or epoch < num_epochs:
        for fi in rng.permutation(len(files)):
            rf = tfrecord.RecordFile(files[fi])
            try:
                for i in range(len(rf)):
                    for key in FEATURE_KEYS[1:]:        # FixedLenFeature: a missing key is an error, as in TF
                        rf.feature_int64(i, key)
                    yield bytes(rf.feature_bytes(i, "image/encoded"))
            finally:
                rf.close()
        epoch += 1


def _prefetch_map(fn, it, num_threads, window):
    """Ordered map over ``it`` on a thread pool, at most ``window`` items in flight."""
    pool = ThreadPoolExecutor(max_workers=num_threads)
    pending = deque()
    try:
        for item in it:
            pending.append(pool.submit(fn, item))
            if len(pending) >= window:
                yield pending.popleft().result()
        while pending:
            yield pending.popleft().result()
    finally:
        pool.shutdown(wait=False, cancel_futures=True)


class ShuffleQueue(object):
    """tf.RandomShuffleQueue as used by tf.train.shuffle_batch: elements live in one HBM tensor;
    dequeue_many draws uniformly without replacement and back-fills the holes from the tail."""

    def __init__(self, engine, capacity, shape, rng):
        self.eng = engine
        self.capacity = int(capacity)
        self.shape = tuple(int(s) for s in shape)
        self.store = engine.mem.empty((self.capacity,) + self.shape)
        self.size = 0
        self.rng = rng

    def enqueue_resized(self, img_u8):
        assert self.size < self.capacity
        slot = self.eng.mem.view(self.store, self.size * int(np.prod(self.shape)), self.shape)
        self.eng.resize_bicubic_u8(img_u8, slot)
        self.size += 1

    def dequeue_many(self, n):
        idx = self.rng.choice(self.size, size=n, replace=False)
        batch = self.eng.mem.gather_rows(self.store, idx)
        # swap-remove, highest index first so the tail elements moved in are never ones being removed
        for i in sorted((int(v) for v in idx), reverse=True):
            last = self.size - 1
            if i != last:
                self.eng.mem.copy_row(self.store, last, i)
            self.size -= 1
        return batch


def batcher(filenames, batch_size, resize_shape=None, num_epochs=None, min_after_dequeue=4000, engine=None,
            seed=0, rank=0, world=1, num_threads=None, max_batches=None):
    """Generator of device tensors [batch_size,H,W,3] float32 (RGB 0..255, TF1-bicubic resized).

    Same arguments as the reference's ``batcher`` (datapipe.py:55-78) plus the engine that owns the
    device, the shard partition and a seed.  Like tf.train.shuffle_batch it fills the queue to
    ``min_after_dequeue`` before the first batch (capacity = min_after_dequeue + 3*batch_size), and when the
    epochs are exhausted it drains the queue and drops the last partial batch.
    """
    if engine is None:
        raise L.FaststyleError("datapipe.batcher needs the Engine that owns the device (no CPU resize path)")
    if resize_shape is None:
        raise L.FaststyleError("batching needs a static image shape: pass resize_shape (train.py --preprocess_size)")
    if max_batches is not None and max_batches <= 0:       # (checked before anything is read: a zero cap yields nothing)
        return
    files = sorted(filenames)[rank::world]
    if not files:
        raise L.FaststyleError("rank %d of %d has no TFRecord shard (%d files)" % (rank, world, len(filenames)))
    H, W = (int(v) for v in resize_shape)
    rng = np.random.default_rng(seed + 7919 * rank)
    capacity = min_after_dequeue + 3 * batch_size                      # datapipe.py:73
    queue = ShuffleQueue(engine, capacity, (H, W, 3), rng)
    threads = num_threads or min(32, max(4, (os.cpu_count() or 8) // max(1, world)))
    decoded = _prefetch_map(lambda d: decode_jpeg(d, packed=False), _examples(files, num_epochs, rng), threads, window=4 * threads)
    produced = 0
    for img in decoded:
        queue.enqueue_resized(img)
        while queue.size - batch_size >= min_after_dequeue:            # RandomShuffleQueue.dequeue_many's condition
            yield queue.dequeue_many(batch_size)
            produced += 1
            if max_batches is not None and produced >= max_batches:
                decoded.close()
                return
    while queue.size >= batch_size:                                    # queue closed: drain, drop the remainder
        yield queue.dequeue_many(batch_size)
        produced += 1
        if max_batches is not None and produced >= max_batches:
            return
