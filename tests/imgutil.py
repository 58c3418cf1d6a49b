import io

import numpy as np
from PIL import Image


def load_rgb(path):
    return np.asarray(Image.open(path).convert("RGB"))


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def jpeg_roundtrip(img_f32):
    """float image -> rint/saturate u8 -> JPEG q95 4:2:0 -> decoded u8 (OpenCV imwrite defaults)."""
    u8 = np.clip(np.rint(img_f32), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(u8).save(buf, format="JPEG", quality=95, subsampling="4:2:0")
    return np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
