#!/usr/bin/env python
"""EXPERIMENT (round-4 review item 5), CPU part: what would split-bf16 products cost the Winograd-domain GEMMs of the VGG16 convs in ACCURACY?

fp32 MFMA shares the vector pipe on gfx950 (MI355X_MICROARCH.md:41,447-448), which is why the F(4x4,3x3) kernel's input transform cannot hide
beside its matrix instructions.  bf16 MFMA does not share it and runs at 16x the rate: with the transformed operands U (filter) and V (input)
split into bf16 pieces, U = Uh + Ul (+ Um), V = Vh + Vl (+ Vm), one fp32 product becomes 3 bf16 products (Uh Vh + Uh Vl + Ul Vh; the dropped
Ul Vl term and the 16 bits the two pieces keep of a 24-bit mantissa cost ~2^-16 relative per operand) or 6 (three pieces, every term above
2^-24 kept), each accumulated in fp32 by the hardware.  This script restates exactly that arithmetic in numpy on the two layers the review
names -- conv4_2 (512 -> 512, deepest reduction) and conv1_2 (64 -> 64) -- and reports max / rms error of the OUTPUT against the float64 direct
convolution, beside the same figures for the fp32 F(4x4,3x3) arithmetic the shipped kernel performs (transforms in fp32, filter transform in
float64 rounded once, products and sums in fp32).  No GPU involved; tools/mfma_bf16x3.hip measures the instruction-mix side.

usage: python tools/bf16x3_error.py            (about a minute)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nnops  # noqa: E402

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def split(x, pieces):
    out, rest = [], np.asarray(x, np.float32)
    for _ in range(pieces):
        h = bf16(rest)
        out.append(h)
        rest = (rest - h).astype(np.float32)      # exact in fp32
    return out


def gemm32(U, V):
    """sum_k U[k, m] V[k, t] with fp32 products (exact for bf16 pieces, rounded for fp32 operands -- np.float32 multiply) and an fp32 running sum
    in the order of k (the matrix instruction's fmaf chain is at least as accurate)."""
    acc = np.zeros((U.shape[1], V.shape[1]), np.float32)
    for k in range(U.shape[0]):
        acc = (acc + np.outer(U[k], V[k]).astype(np.float32)).astype(np.float32)
    return acc


def run(name, H, W, Cin, Cout, seed):
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((1, H, W, Cin)), 0).astype(np.float32) * 50.0          # post-ReLU activations
    w = (rng.standard_normal((3, 3, Cin, Cout)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    want = nnops.conv2d(x.astype(np.float64), w.astype(np.float64), 1, "SAME")[0]
    ty, tx = H // 4, W // 4
    xp = np.pad(x[0], ((1, 1), (1, 1), (0, 0)))
    # U[pos][ci][co] (float64 transform, rounded once), V[pos][ci][tile] (fp32 transform, as the kernel's)
    U = np.einsum("ik,klcd,jl->ijcd", G, w.astype(np.float64), G).reshape(36, Cin, Cout).astype(np.float32)
    d = np.stack([xp[4 * i:4 * i + 6, 4 * j:4 * j + 6, :] for i in range(ty) for j in range(tx)])           # [tile][6][6][ci]
    t1 = np.einsum("ik,tklc->tilc", BT.astype(np.float32), d).astype(np.float32)
    V = np.einsum("tilc,jl->tijc", t1, BT.astype(np.float32)).astype(np.float32).reshape(ty * tx, 36, Cin).transpose(1, 2, 0)   # [pos][ci][tile]
    res = {}
    for mode in ("fp32", "bf16x3", "bf16x6"):
        M = np.zeros((36, Cout, ty * tx), np.float32)
        for p in range(36):
            if mode == "fp32":
                M[p] = gemm32(U[p], V[p])
            else:
                us, vs = split(U[p], 2 if mode == "bf16x3" else 3), split(V[p], 2 if mode == "bf16x3" else 3)
                terms = [(0, 0), (0, 1), (1, 0)] if mode == "bf16x3" else [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
                acc = np.zeros((Cout, ty * tx), np.float32)
                for (i, j) in terms[::-1]:           # small terms first, as a kernel would issue them
                    acc = (acc + gemm32(us[i], vs[j])).astype(np.float32)
                M[p] = acc
        Mt = M.reshape(6, 6, Cout, ty, tx)
        y = np.einsum("ik,klcyx,jl->yixjc", AT.astype(np.float32), Mt, AT.astype(np.float32)).astype(np.float32).reshape(H, W, Cout)
        e = y.astype(np.float64) - want
        res[mode] = (np.abs(e).max() / np.abs(want).max(), np.sqrt((e ** 2).mean()) / np.sqrt((want ** 2).mean()))
    print("%-28s " % name + " | ".join("%s max %.2e rms %.2e" % (m, res[m][0], res[m][1]) for m in ("fp32", "bf16x3", "bf16x6")), flush=True)
    return res


if __name__ == "__main__":
    print("error of the F(4x4,3x3) output against the float64 direct convolution: max |e| / max |y|, rms(e) / rms(y)")
    run("conv4_2 512->512, 16x16 px", 16, 16, 512, 512, 1)
    run("conv3_2 256->256, 16x16 px", 16, 16, 256, 256, 2)
    run("conv1_2  64->64,  32x32 px", 32, 32, 64, 64, 3)
