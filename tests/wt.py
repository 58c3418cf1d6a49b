"""numpy versions of the filter re-layouts (test helpers)."""
import numpy as np

_R = {(0, 0): (0, 1, 2), (0, 1): (), (1, 0): (0, 1), (1, 1): (2,)}


def upconv_weff(w):
    """[3,3,Ci,Co] -> [2,2,Ci,4*Co]: the four phase filters of NEAREST x4 + conv3x3 stride 2
    (SURVEY.md §8a row a7)."""
    _, _, ci, co = w.shape
    out = np.zeros((2, 2, ci, 4, co), dtype=w.dtype)
    for a in range(2):
        for b in range(2):
            for dy in range(2):
                for dx in range(2):
                    for kh in _R[(a, dy)]:
                        for kw in _R[(b, dx)]:
                            out[dy, dx, :, a * 2 + b, :] += w[kh, kw]
    return out.reshape(2, 2, ci, 4 * co)
