// kw-folding of the 9x9, 16 -> 3 output layer (reference im_transf_net.py:69-70).
//
// With only 3 output channels an MFMA tile (N = 16) would be 3/16 full.  Writing kw = 5b + v
// (b in {0,1}, v in 0..4) the conv becomes
//     Z[q,(v,co)] = sum_{kh,b,ci} X[q + (kh, 5b) - pad, ci] * W[kh, 5b+v, ci, co]     (N = 15 of 16)
//     Y[p,co]     = sum_v Z[p + (0,v), (v,co)]
// i.e. a 9x2-tap conv with horizontal tap spacing 5 and 16 "virtual" output channels over a 4-column
// wider image, followed by a 5-term shifted sum: 4.5x fewer matrix instructions, identical maths up
// to summation order.  The filter gradient uses the adjoint: dY is unfolded to [q][(v,co)] and a
// 9x2-tap wgrad produces dW in the folded layout.
#include "fs_kernels.h"

namespace fs {

// every launcher reports a failed launch (bad configuration, missing code object) instead of returning success
static inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : -3; }


// wf[kh][b][ci][16] from w[9][9][Ci][3]
__global__ __launch_bounds__(256) void wt_fold5_fwd_kernel(const float* w, float* wf, int Ci) {
    const int total = 9 * 2 * Ci * 16;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int j = i & 15;
        int r = i >> 4;
        const int ci = r % Ci;
        r /= Ci;
        const int b = r & 1, kh = r >> 1;
        const int v = j / 3, co = j - v * 3, kw = 5 * b + v;
        wf[i] = (j < 15 && kw < 9) ? w[((kh * 9 + kw) * Ci + ci) * 3 + co] : 0.f;
    }
}

// dw[9][9][Ci][3] from dwf[kh][b][ci][16]
__global__ __launch_bounds__(256) void wt_fold5_back_kernel(const float* dwf, float* dw, int Ci) {
    const int total = 81 * Ci * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int co = i % 3;
        int r = i / 3;
        const int ci = r % Ci;
        r /= Ci;
        const int kw = r % 9, kh = r / 9;
        const int b = kw / 5, v = kw - 5 * b;
        dw[i] = dwf[(((kh * 2 + b) * Ci + ci) << 4) + v * 3 + co];
    }
}

// z[n,oy,ox,co] = sum_v Z[n,oy,ox+v,(v,co)]; per-block {mean, M2, count} partials per channel for
// the instance norm (same format as the conv epilogue's: [N][T][3][3], T = ceil(Ho*Wo/256)).
__device__ __forceinline__ float fold_ld(float v) { return v; }
__device__ __forceinline__ float fold_ld(unsigned short v) { return __builtin_bit_cast(float, (unsigned)v << 16); }  // bf16
// The block's 256 pixels read Z columns ox .. ox+4 of their rows: a contiguous range of at most 260 + 4 x (rows crossed) Z
// pixels.  Each is fetched ONCE as whole 16-channel rows (two / four 16-byte loads per thread) and laid out channel-major in
// LDS, so the 15 reads of a pixel are conflict-free LDS reads instead of 15 two-byte global loads (the kernel was bound by
// the number of global-load instructions: 1080p batch 8, 0.30 ms for 0.73 GB).  Blocks whose range exceeds the LDS array
// (maps narrower than ~9 pixels) read global memory directly, as before; the sums run in the same order either way.
constexpr int kFoldCap = 384;
template <typename TZ>
__device__ __forceinline__ void fold_ld16(const TZ* src, float* o);
template <>
__device__ __forceinline__ void fold_ld16<float>(const float* src, float* o) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 t = s4[k];
        o[4 * k] = t.x;
        o[4 * k + 1] = t.y;
        o[4 * k + 2] = t.z;
        o[4 * k + 3] = t.w;
    }
}
template <>
__device__ __forceinline__ void fold_ld16<unsigned short>(const unsigned short* src, float* o) {
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint4 t = s4[k];
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[8 * k + 2 * e] = __builtin_bit_cast(float, w[e] << 16);
            o[8 * k + 2 * e + 1] = __builtin_bit_cast(float, w[e] & 0xFFFF0000u);
        }
    }
}
template <typename TZ>
__global__ __launch_bounds__(256) void fold5_fwd_kernel(const TZ* Z, float* z, float* stats, int HW, int Wo) {
    __shared__ float sh[4];
    __shared__ float zl[15 * kFoldCap];
    const int n = blockIdx.y, T = gridDim.x;
    const int p0 = blockIdx.x * 256, p = p0 + threadIdx.x;
    const bool ok = p < HW;
    const int Wz = Wo + 4;
    const int oy0 = p0 / Wo, p1 = min(p0 + 255, HW - 1), oy1 = p1 / Wo;
    const int zq0 = oy0 * Wz + (p0 - oy0 * Wo), count = oy1 * Wz + (p1 - oy1 * Wo) + 4 - zq0 + 1;   // (block-uniform)
    const TZ* Zn = Z + (size_t)n * (HW / Wo) * Wz * 16;
    const bool staged = count <= kFoldCap;
    if (staged) {
        for (int zi = threadIdx.x; zi < count; zi += 256) {
            float o[16];
            fold_ld16<TZ>(Zn + (size_t)(zq0 + zi) * 16, o);
#pragma unroll
            for (int j = 0; j < 15; ++j) zl[j * kFoldCap + zi] = o[j];
        }
        __syncthreads();
    }
    float y[3] = {0.f, 0.f, 0.f};
    if (ok) {
        const int oy = p / Wo, ox = p - oy * Wo;
        if (staged) {
            const int zi = oy * Wz + ox - zq0;
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                y[0] += zl[(v * 3 + 0) * kFoldCap + zi + v];
                y[1] += zl[(v * 3 + 1) * kFoldCap + zi + v];
                y[2] += zl[(v * 3 + 2) * kFoldCap + zi + v];
            }
        } else {
            const TZ* src = Zn + ((size_t)oy * Wz + ox) * 16;
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                y[0] += fold_ld(src[v * 16 + v * 3 + 0]);
                y[1] += fold_ld(src[v * 16 + v * 3 + 1]);
                y[2] += fold_ld(src[v * 16 + v * 3 + 2]);
            }
        }
        float* dst = z + ((size_t)n * HW + p) * 3;
        dst[0] = y[0];
        dst[1] = y[1];
        dst[2] = y[2];
    }
    const int cnt = min(256, HW - (int)blockIdx.x * 256);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = ok ? y[c] : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
        __syncthreads();
        const float mean = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)cnt;
        const float d = ok ? y[c] - mean : 0.f;
        float q = d * d;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) {
            float* st = stats + (((size_t)n * T + blockIdx.x) * 3 + c) * 3;
            st[0] = mean;
            st[1] = sh[0] + sh[1] + sh[2] + sh[3];
            st[2] = (float)cnt;
        }
    }
}

// dYs[n,oy,q,(v,co)] = dz[n,oy,q-v,co] (0 outside [0,Wo)), q in [0, Wo+4); column 15 is zero
__global__ __launch_bounds__(256) void unfold5_kernel(const float* dz, float* dys, int Ho, int Wo, size_t total_q) {
    const int Wz = Wo + 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total_q; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % Wz);
        const size_t row = i / Wz;  // n*Ho + oy
        const float* src = dz + row * Wo * 3;
        float o[16];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const int x = q - v;
            const bool in = x >= 0 && x < Wo;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[v * 3 + c] = in ? src[x * 3 + c] : 0.f;
        }
        o[15] = 0.f;
        float4* dst = reinterpret_cast<float4*>(dys + i * 16);
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        dst[1] = make_float4(o[4], o[5], o[6], o[7]);
        dst[2] = make_float4(o[8], o[9], o[10], o[11]);
        dst[3] = make_float4(o[12], o[13], o[14], o[15]);
    }
    (void)Ho;
}

int wt_fold5_fwd(const float* w, float* wf, int Ci, hipStream_t s) {
    hipLaunchKernelGGL(wt_fold5_fwd_kernel, dim3(cdiv(18 * Ci * 16, 256)), dim3(256), 0, s, w, wf, Ci);
    return launch_status();
}
int wt_fold5_back(const float* dwf, float* dw, int Ci, hipStream_t s) {
    hipLaunchKernelGGL(wt_fold5_back_kernel, dim3(cdiv(81 * Ci * 3, 256)), dim3(256), 0, s, dwf, dw, Ci);
    return launch_status();
}
int fold5_fwd(const float* Z, float* z, float* stats, int N, int Ho, int Wo, hipStream_t s) {
    hipLaunchKernelGGL(fold5_fwd_kernel<float>, dim3(cdiv(Ho * Wo, 256), N), dim3(256), 0, s, Z, z, stats, Ho * Wo, Wo);
    return launch_status();
}
int fold5_fwd_bf16(const unsigned short* Z, float* z, float* stats, int N, int Ho, int Wo, hipStream_t s) {
    hipLaunchKernelGGL(fold5_fwd_kernel<unsigned short>, dim3(cdiv(Ho * Wo, 256), N), dim3(256), 0, s, Z, z, stats, Ho * Wo, Wo);
    return launch_status();
}
int unfold5(const float* dz, float* dys, int N, int Ho, int Wo, hipStream_t s) {
    const size_t total = (size_t)N * Ho * (Wo + 4);
    hipLaunchKernelGGL(unfold5_kernel, dim3((unsigned)min((size_t)8192, (total + 255) / 256)), dim3(256), 0, s, dz, dys, Ho,
                       Wo, total);
    return launch_status();
}

}  // namespace fs
