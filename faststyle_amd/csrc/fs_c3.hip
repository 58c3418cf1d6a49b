// 3x3 stride-1 SAME convolution onto THREE output channels on the vector ALU: the input gradient of VGG16 conv1_1
// (reference libs/vgg16.py:40-53; dL/d(generated image) closes the backward pass of fs_perceptual_loss).
//
// On the matrix cores this layer multiplies 16-column tiles of which 3 columns are real (0.53 ms per batch of 32 at 38.6 padded
// GFLOP for 7.2 useful ones).  Here a thread owns one output pixel and its 3 channels: the 18x18 input patch of a 16x16-pixel
// tile passes through LDS 32 channels at a time ([pixel][32+4]: a lane's 16-byte read and its 15 neighbours' cover all 64 banks), the 1,728 filter values
// are wave-uniform and arrive through the scalar cache as SGPR operands of v_fma_f32 -- 12 FMAs per 16-byte LDS read, no
// padding work.  The layer reads its 64-channel input once: ~0.54 GB per batch of 32, the HBM floor of ~0.14 ms.
#include "fs_kernels.h"

namespace fs {

namespace {
constexpr int kT = 16;            // output tile side
constexpr int kP = kT + 2;        // patch side
constexpr unsigned kOOB = 0x80000000u;
}  // namespace

template <int C>
__global__ __launch_bounds__(256) void conv3x3_to3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, int H, int W, int tiles_x, int tiles_y) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int CH = C / 2;     // channels per pass: half a patch in LDS (46.7 KB -> three workgroups per CU)
    constexpr int S = CH + 4;     // LDS pitch of a patch pixel
    constexpr int Q = CH / 4;     // float4 per pixel and pass
    const int tid = threadIdx.x;
    const int tiles = tiles_x * tiles_y;
    const int n = (int)blockIdx.x / tiles, tr = (int)blockIdx.x - n * tiles;
    const int tyi = tr / tiles_x, txi = tr - tyi * tiles_x;
    const int oy0 = tyi * kT, ox0 = txi * kT;
    const float* xn = x + (size_t)n * H * W * C;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, (unsigned)(H * W * C) * 4u, 0x00020000);
    // staging: kP*kP pixels x Q float4 per pass, padding / outside the image -> zeros (out-of-range offset).  ALL loads of a
    // pass are issued before the first LDS write (a "load, wait, store" loop pays one memory latency per iteration), and the
    // second pass's loads fly while the first pass computes
    constexpr int NE = (kP * kP * Q + 255) / 256;
    unsigned go[NE];
    int ld[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + i * 256;
        const int pix = e / Q, q = e - pix * Q;
        const int py = pix / kP, px = pix - py * kP;
        const int sy = oy0 - 1 + py, sx = ox0 - 1 + px;
        const bool ok = e < kP * kP * Q && sy >= 0 && sy < H && sx >= 0 && sx < W;
        go[i] = ok ? (unsigned)((sy * W + sx) * C + q * 4) * 4u : kOOB;
        ld[i] = e < kP * kP * Q ? pix * S + q * 4 : -1;
    }
    float4 v[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, go[i], 0, 0));
    const int ty = tid >> 4, tx = tid & 15;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (ld[i] >= 0) *reinterpret_cast<float4*>(smem + ld[i]) = v[i];
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < NE; ++i) v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, go[i], CH * 4, 0));
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* src = smem + ((ty + tap / 3) * kP + tx + tap % 3) * S;
            const float* wt = w + (tap * C + half * CH) * 3;       // [ci][3], wave-uniform: scalar loads
#pragma unroll 4
            for (int q = 0; q < Q; ++q) {
                const float4 u = *reinterpret_cast<const float4*>(src + q * 4);
                const float* wq = wt + q * 12;
                a0 = fmaf(u.x, wq[0], a0);
                a1 = fmaf(u.x, wq[1], a1);
                a2 = fmaf(u.x, wq[2], a2);
                a0 = fmaf(u.y, wq[3], a0);
                a1 = fmaf(u.y, wq[4], a1);
                a2 = fmaf(u.y, wq[5], a2);
                a0 = fmaf(u.z, wq[6], a0);
                a1 = fmaf(u.z, wq[7], a1);
                a2 = fmaf(u.z, wq[8], a2);
                a0 = fmaf(u.w, wq[9], a0);
                a1 = fmaf(u.w, wq[10], a1);
                a2 = fmaf(u.w, wq[11], a2);
            }
        }
        if (half == 0) __syncthreads();   // the patch is overwritten by the second pass
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < H && ox < W) {
        float* d = y + (((size_t)n * H + oy) * W + ox) * 3;
        d[0] = a0;
        d[1] = a1;
        d[2] = a2;
    }
}

bool conv3x3_to3_eligible(const ConvArgs& a) {
    return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad_t == 1 && a.pad_l == 1 && a.Cout == 3 && a.Cin == 64 && a.Ho == a.H &&
           a.Wo == a.W && a.src_mode == SRC_PLAIN && !a.in_a && !a.bias && !a.out_relu && !a.shuffle && !a.stats && !a.add_src &&
           !a.mask_src && a.w_nstride == 0 && (a.dil_x <= 1) && tune_int("FS_C3_VALU", 1) != 0;
}

int conv3x3_to3_launch(const ConvArgs& a, hipStream_t s) {
    const int tx = cdiv(a.Wo, kT), ty = cdiv(a.Ho, kT);
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(conv3x3_to3_kernel<64>));
    Profiler* prof = Profiler::current();
    if (prof) prof->begin(PF_C3, 2.0 * a.N * a.Ho * a.Wo * 9.0 * a.Cin * a.Cout, s);   // (reported with the narrow-output family)
    hipLaunchKernelGGL(conv3x3_to3_kernel<64>, dim3((unsigned)(a.N * tx * ty)), dim3(256), (size_t)(kP * kP * (32 + 4) * 4), s, a.x, a.w, a.y, a.H,
                       a.W, tx, ty);
    if (prof) prof->end(s);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
