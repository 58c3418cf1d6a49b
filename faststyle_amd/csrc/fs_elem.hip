// HBM-bound elementwise / reduction kernels of the hot path (all NHWC fp32, channel innermost so a
// wave reads whole 256-byte pixel rows):
//   instance-norm statistics finalize (Chan merge of the conv epilogue's per-tile partials),
//   residual add, scaled tanh, instance-norm backward, max-pool, ReLU/pool gradient routing,
//   loss reductions, slab reduction, TF-style Adam, filter re-layouts.
#include "fs_kernels.h"

#include <cstdlib>

namespace fs {

// every launcher reports a failed launch (bad configuration, missing code object) instead of returning success
static inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : -3; }


__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum over the 256 threads of a block; result valid in every thread.  `sh` needs 4 floats.
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---------------------------------------------------------------- instance-norm finalize
// stats: [N][T][Cv][3] = {mean, M2, count} per conv tile, Cv = groups*C (groups=4 for the
// pixel-shuffled resize-conv whose 4 phases hold the same real channel).
// LPC lanes per channel (256/LPC channels per block): the lanes stride over the tile list with 4 loads in
// flight, merge with Chan's update in fp64, then a fixed-shape tree folds the LPC lane results.  LPC = 64 (one wave
// per channel) when there are many (sample, channel) pairs; LPC = 256 (a whole workgroup per channel) when there
// are few, so that a batch-1 frame still spreads its tile list over enough lanes.
template <int LPC>
__global__ __launch_bounds__(256) void in_finalize_kernel(const float* stats, int T, int C, int groups, const float* gamma,
                                                          const float* beta, float eps, float* mean, float* rstd,
                                                          float* oa, float* ob) {
    __shared__ double sc[256], sm[256], sq[256];
    constexpr int CPB = 256 / LPC;
    const int n = blockIdx.x, lane = threadIdx.x % LPC, w = threadIdx.x / LPC;
    const int c = blockIdx.y * CPB + w;
    const int Cv = C * groups;
    double cnt = 0, mu = 0, m2 = 0;
    if (c < C) {
        const int total = T * groups;
        for (int i0 = lane; i0 < total; i0 += 4 * LPC) {
            float vm[4], vq[4], vc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * LPC;
                vc[u] = 0.f;
                vm[u] = vq[u] = 0.f;
                if (i < total) {
                    const int t = i / groups, q = i - t * groups;
                    const float* st = stats + (((size_t)n * T + t) * Cv + q * C + c) * 3;
                    vm[u] = st[0];
                    vq[u] = st[1];
                    vc[u] = st[2];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double cb = vc[u], mb = vm[u], qb = vq[u];
                if (cb > 0) {
                    const double nn = cnt + cb, d = mb - mu, r = cb / nn;
                    mu += d * r;
                    m2 += qb + d * d * cnt * r;
                    cnt = nn;
                }
            }
        }
    }
    // fixed-shape tree over the LPC lanes of the channel (one fp64 divide per merge)
    for (int stride = LPC / 2; stride > 0; stride >>= 1) {
        sc[threadIdx.x] = cnt;
        sm[threadIdx.x] = mu;
        sq[threadIdx.x] = m2;
        __syncthreads();
        if (lane < stride) {
            const double cb = sc[threadIdx.x + stride], mb = sm[threadIdx.x + stride], qb = sq[threadIdx.x + stride];
            if (cb > 0) {
                const double nn = cnt + cb, d = mb - mu, r = cb / nn;
                mu += d * r;
                m2 += qb + d * d * cnt * r;
                cnt = nn;
            }
        }
        __syncthreads();
    }
    if (lane == 0 && c < C) {
        const float var = (float)(m2 / cnt);
        const float r = 1.0f / sqrtf(var + eps);
        const float fm = (float)mu;
        const float a = gamma[c] * r;
        mean[n * C + c] = fm;
        rstd[n * C + c] = r;
        oa[n * C + c] = a;
        ob[n * C + c] = beta[c] - fm * a;
    }
}

// Large images have thousands of tiles per sample (1080p: ~9000) and the finalize kernel only has N*C waves to
// walk them, each lane with a strided 12-byte read.  Pre-reduction: grid (S, N), a workgroup merges a contiguous
// range of tiles for ALL Cv channels (thread = (tile lane, channel): the Cv*3 floats of a tile are contiguous, so
// the reads coalesce), fixed merge order, output in the same {mean, M2, count} record format with T := S.
__global__ __launch_bounds__(256) void in_prereduce_kernel(const float* __restrict__ stats, int T, int Cv, int S,
                                                           float* __restrict__ out) {
    __shared__ double sc[256], sm[256], sq[256];
    const int n = blockIdx.y, sp = blockIdx.x;
    const int TL = 256 / Cv;  // tile lanes (Cv <= 256)
    const int cv = threadIdx.x % Cv, tl = threadIdx.x / Cv;
    const int t0 = (int)((long long)sp * T / S), t1 = (int)((long long)(sp + 1) * T / S);
    double cnt = 0, mu = 0, m2 = 0;
    if (tl < TL)
        for (int t = t0 + tl; t < t1; t += TL) {
            const float* st = stats + (((size_t)n * T + t) * Cv + cv) * 3;
            const double cb = st[2], mb = st[0], qb = st[1];
            if (cb > 0) {
                const double nn = cnt + cb, d = mb - mu, r = cb / nn;
                mu += d * r;
                m2 += qb + d * d * cnt * r;
                cnt = nn;
            }
        }
    sc[threadIdx.x] = cnt;
    sm[threadIdx.x] = mu;
    sq[threadIdx.x] = m2;
    __syncthreads();
    if (tl == 0) {
        for (int k = 1; k < TL; ++k) {
            const double cb = sc[k * Cv + cv], mb = sm[k * Cv + cv], qb = sq[k * Cv + cv];
            if (cb > 0) {
                const double nn = cnt + cb, d = mb - mu, r = cb / nn;
                mu += d * r;
                m2 += qb + d * d * cnt * r;
                cnt = nn;
            }
        }
        float* o = out + (((size_t)n * S + sp) * Cv + cv) * 3;
        o[0] = (float)mu;
        o[1] = (float)m2;
        o[2] = (float)cnt;
    }
}

int in_finalize(const float* stats, int N, int T, int C, int groups, const float* gamma, const float* beta, float eps,
                float* mean, float* rstd, float* a, float* b, hipStream_t s, float* scratch) {
    const int Cv = C * groups;
    // (tests lower FS_FINALIZE_MIN_T so small images take the two-level path)
    const int min_t = tune_int("FS_FINALIZE_MIN_T", 16 * kFinalizeSplit);  // 1024 tiles: 720p and up; training sizes stay single-level
    if (scratch && T * groups > min_t && T > kFinalizeSplit && Cv <= 256) {  // scratch: N * kFinalizeSplit * Cv * 3 floats
        hipLaunchKernelGGL(in_prereduce_kernel, dim3(kFinalizeSplit, N), dim3(256), 0, s, stats, T, Cv, kFinalizeSplit, scratch);
        stats = scratch;
        T = kFinalizeSplit;
    }
    if (N * C <= 512 && T * groups >= 256)
        hipLaunchKernelGGL(in_finalize_kernel<256>, dim3(N, C), dim3(256), 0, s, stats, T, C, groups, gamma, beta, eps, mean,
                           rstd, a, b);
    else
        hipLaunchKernelGGL(in_finalize_kernel<64>, dim3(N, cdiv(C, 4)), dim3(256), 0, s, stats, T, C, groups, gamma, beta,
                           eps, mean, rstd, a, b);
    return launch_status();
}

// ---------------------------------------------------------------- residual add / tanh
// out[n,y,x,c] = z*a+b + T(skip[n,y+2,x+2,c]),  T = optional affine+ReLU (block 0 reads the raw
// initconv_2 output).  reference im_transf_net.py:268-274
// grid (blocks over one row, H, N); a thread owns 4 channels of one pixel (C = 64)
__global__ __launch_bounds__(256) void apply_res_kernel(const float* __restrict__ z, const float* __restrict__ a,
                                                        const float* __restrict__ b, const float* __restrict__ skip,
                                                        const float* __restrict__ sa, const float* __restrict__ sb, int skip_relu,
                                                        float* __restrict__ out, int H, int W, int C) {
    const int c4n = C >> 2;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= W * c4n) return;
    const int x = j / c4n, c = (j - x * c4n) * 4;
    const int y = blockIdx.y, n = blockIdx.z;
    const size_t i = (((size_t)n * H + y) * W + x) * C + c;
    const float4 zz = *reinterpret_cast<const float4*>(z + i);
    const float4 s4 = *reinterpret_cast<const float4*>(skip + (((size_t)n * (H + 4) + y + 2) * (W + 4) + x + 2) * C + c);
    const float4 av = *reinterpret_cast<const float4*>(a + n * C + c), bv = *reinterpret_cast<const float4*>(b + n * C + c);
    float sk[4] = {s4.x, s4.y, s4.z, s4.w};
    if (sa) {
        const float4 sav = *reinterpret_cast<const float4*>(sa + n * C + c), sbv = *reinterpret_cast<const float4*>(sb + n * C + c);
        sk[0] = fmaf(sk[0], sav.x, sbv.x);
        sk[1] = fmaf(sk[1], sav.y, sbv.y);
        sk[2] = fmaf(sk[2], sav.z, sbv.z);
        sk[3] = fmaf(sk[3], sav.w, sbv.w);
    }
    if (skip_relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sk[q] = fmaxf(sk[q], 0.f);
    }
    *reinterpret_cast<float4*>(out + i) = make_float4(fmaf(zz.x, av.x, bv.x) + sk[0], fmaf(zz.y, av.y, bv.y) + sk[1],
                                                      fmaf(zz.z, av.z, bv.z) + sk[2], fmaf(zz.w, av.w, bv.w) + sk[3]);
}

int apply_res(const float* z, const float* a, const float* b, const float* skip, const float* sa, const float* sb,
              int skip_relu, float* out, int N, int H, int W, int C, hipStream_t s) {
    if (C % 4) return -1;
    hipLaunchKernelGGL(apply_res_kernel, dim3(cdiv(W * (C / 4), 256), H, N), dim3(256), 0, s, z, a, b, skip, sa, sb, skip_relu,
                       out, H, W, C);
    return launch_status();
}

// y = (255*tanh(a z + b) + 255)/2   reference im_transf_net.py:202-215
// A thread owns 4 consecutive floats (16-byte load and store: the scalar form ran at 3.2 TB/s, 123 us per 1080p batch of 8); sample and channel
// of the quad's first element by one division, the rest by comparison (a quad may straddle two samples when HW * C is not a multiple of 4).
__global__ __launch_bounds__(256) void apply_tanh4_kernel(const float* __restrict__ z, const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ y, int per, int C, size_t total4) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total4; q += (size_t)gridDim.x * 256) {
        const size_t i0 = q * 4;
        const int n = (int)(i0 / (size_t)per);
        const int r = (int)(i0 - (size_t)n * per);
        int c = r % C;
        const float4 zz = *reinterpret_cast<const float4*>(z + i0);
        const float zv[4] = {zz.x, zz.y, zz.z, zz.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int nk = r + k >= per ? n + 1 : n;      // (r + k >= per: the next sample starts at channel 0 + ...)
            const int ck = r + k >= per ? (r + k - per) % C : c;
            const float v = fmaf(zv[k], a[nk * C + ck], b[nk * C + ck]);
            o[k] = (255.0f * tanhf(v) + 255.0f) / 2.0f;
            c = c + 1 == C ? 0 : c + 1;
        }
        *reinterpret_cast<float4*>(y + i0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
__global__ __launch_bounds__(256) void apply_tanh_kernel(const float* z, const float* a, const float* b, float* y, int HW,
                                                         int C, size_t first, size_t total) {
    for (size_t i = first + (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int n = (int)(i / ((size_t)HW * C));
        const float v = fmaf(z[i], a[n * C + c], b[n * C + c]);
        y[i] = (255.0f * tanhf(v) + 255.0f) / 2.0f;
    }
}

int apply_tanh(const float* z, const float* a, const float* b, float* y, int N, int HW, int C, hipStream_t s) {
    const size_t total = (size_t)N * HW * C;
    const bool vec = (size_t)HW * C < ((size_t)1 << 31) && !(((uintptr_t)z | (uintptr_t)y) & 15) && total >= 4;
    const size_t total4 = vec ? total / 4 : 0;
    if (total4)
        hipLaunchKernelGGL(apply_tanh4_kernel, dim3((unsigned)min((size_t)8192, (total4 + 255) / 256)), dim3(256), 0, s, z, a, b, y, HW * C, C, total4);
    if (total4 * 4 < total)   // the tail (< 4 elements), or everything when the quads are not aligned
        hipLaunchKernelGGL(apply_tanh_kernel, dim3((unsigned)min((size_t)2048, (total - total4 * 4 + 255) / 256)), dim3(256), 0, s, z, a, b, y, HW, C,
                           total4 * 4, total);
    return launch_status();
}

// out = act(a[n,c] z + b[n,c]), act = identity | ReLU: the instance-norm output MATERIALISED (im_transf_net.py:246 + :98 / :150) -- the fused paths never
// form it (the consumer conv applies the affine while staging); the named export fs_instnorm_apply does, for callers that compose units themselves
__global__ __launch_bounds__(256) void apply_affine_kernel(const float* __restrict__ z, const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ y, int HW, int C, int relu, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int n = (int)(i / ((size_t)HW * C));
        const float v = fmaf(z[i], a[n * C + c], b[n * C + c]);
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}
int apply_affine(const float* z, const float* a, const float* b, float* y, int N, int HW, int C, int relu, hipStream_t s) {
    const size_t total = (size_t)N * HW * C;
    hipLaunchKernelGGL(apply_affine_kernel, dim3((unsigned)min((size_t)8192, (total + 255) / 256)), dim3(256), 0, s, z, a, b, y, HW, C, relu, total);
    return launch_status();
}
__global__ __launch_bounds__(256) void zero_fill_kernel(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}
int zero_fill(float* p, size_t n, hipStream_t s) {   // (a kernel, not hipMemsetAsync: memset nodes misbehave in single-stream graph replays, see zero_words)
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)min((size_t)4096, (n + 255) / 256)), dim3(256), 0, s, p, n);
    return launch_status();
}

// ---------------------------------------------------------------- slab reduction
// out[g][i] = scale * sum_w slabs[g][w][i].  The partial-slab sets are small (<= a few MB) but deep (up to ~500
// slabs), so the reduction is latency-bound: a workgroup covers EL consecutive elements x SG slab groups
// (EL*SG = 256); every thread sums its strided share of the slabs with 8 loads in flight and the SG partials are
// combined through LDS in a fixed order (deterministic; no atomics).
template <int EL>
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int n_wg, size_t count, float scale,
                                                           float* __restrict__ out) {
    constexpr int SG = 256 / EL;
    __shared__ float sh[256];
    const size_t g = blockIdx.y;
    const int el = threadIdx.x % EL, sg = threadIdx.x / EL;
    const size_t i = (size_t)blockIdx.x * EL + el;
    float acc = 0.f;
    if (i < count) {
        const float* p = slabs + g * n_wg * count + i;
        int w = sg;
        for (; w + 7 * SG < n_wg; w += 8 * SG) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(w + u * SG) * count];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; w < n_wg; w += SG) acc += p[(size_t)w * count];
    }
    if (SG == 1) {
        if (i < count) out[g * count + i] = acc * scale;
        return;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (sg == 0 && i < count) {
        float t = sh[el];
#pragma unroll
        for (int k = 1; k < SG; ++k) t += sh[k * EL + el];
        out[g * count + i] = t * scale;
    }
}

// count % 4 == 0 and many slabs: a thread owns 4 consecutive elements (one float4 per slab), 8 threads cover a 128-byte
// line, the 32 slab groups of a workgroup keep all of a thread's loads in flight (<= 8 per pass)
__global__ __launch_bounds__(256) void reduce_slabs4_kernel(const float* __restrict__ slabs, int n_wg, size_t count, float scale,
                                                            float* __restrict__ out) {
    constexpr int EL4 = 8, SG = 32;
    __shared__ float4 sh[256];
    const size_t g = blockIdx.y;
    const int el = threadIdx.x & (EL4 - 1), sg = threadIdx.x >> 3;
    const size_t i = ((size_t)blockIdx.x * EL4 + el) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < count) {
        const float* p = slabs + g * n_wg * count + i;
        int w = sg;
        for (; w + 7 * SG < n_wg; w += 8 * SG) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(w + u * SG) * count);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc.x += v[u].x;
                acc.y += v[u].y;
                acc.z += v[u].z;
                acc.w += v[u].w;
            }
        }
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {   // tail: clamped slab index, contribution masked (keeps the loads unconditional)
            const int ww = w + u * SG;
            v[u] = *reinterpret_cast<const float4*>(p + (size_t)(ww < n_wg ? ww : sg) * count);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float m = (w + u * SG) < n_wg ? 1.f : 0.f;
            acc.x = fmaf(v[u].x, m, acc.x);
            acc.y = fmaf(v[u].y, m, acc.y);
            acc.z = fmaf(v[u].z, m, acc.z);
            acc.w = fmaf(v[u].w, m, acc.w);
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (sg == 0 && i < count) {
        float4 t = sh[el];
#pragma unroll
        for (int k = 1; k < SG; ++k) {
            const float4 q = sh[k * EL4 + el];
            t.x += q.x;
            t.y += q.y;
            t.z += q.z;
            t.w += q.w;
        }
        *reinterpret_cast<float4*>(out + g * count + i) = make_float4(t.x * scale, t.y * scale, t.z * scale, t.w * scale);
    }
}

int reduce_slabs(const float* slabs, int groups, int n_wg, size_t count, float scale, float* out, hipStream_t s) {
    const int deep = n_wg / 8;  // slab groups only while each still owns >= 8 slabs
    if (count % 4 == 0 && n_wg >= 64) {
        hipLaunchKernelGGL(reduce_slabs4_kernel, dim3((unsigned)((count / 4 + 7) / 8), groups), dim3(256), 0, s, slabs, n_wg, count,
                           scale, out);
    } else if (deep >= 16) {
        hipLaunchKernelGGL(reduce_slabs_kernel<16>, dim3((unsigned)((count + 15) / 16), groups), dim3(256), 0, s, slabs, n_wg, count,
                           scale, out);
    } else if (deep >= 4) {
        hipLaunchKernelGGL(reduce_slabs_kernel<64>, dim3((unsigned)((count + 63) / 64), groups), dim3(256), 0, s, slabs, n_wg, count,
                           scale, out);
    } else {
        hipLaunchKernelGGL(reduce_slabs_kernel<256>, dim3((unsigned)((count + 255) / 256), groups), dim3(256), 0, s, slabs, n_wg,
                           count, scale, out);
    }
    return launch_status();
}

// ---------------------------------------------------------------- instance-norm backward
__device__ __forceinline__ float in_bwd_g(float gin, float z, float a, float b, int mode) {
    const float v = fmaf(z, a, b);
    if (mode == 1) return v > 0.f ? gin : 0.f;
    if (mode == 2) {
        const float t = tanhf(v);
        return gin * 127.5f * (1.f - t * t);
    }
    return gin;
}

// partial[n][chunk][C][2] = sums over the chunk's pixels of {g, g*xhat}
__global__ __launch_bounds__(256) void in_bwd_partial_kernel(const float* gin, const float* z, const float* mean,
                                                             const float* rstd, const float* a, const float* b, int mode,
                                                             float* partial, int HW, int C, int chunk_px) {
    __shared__ float sh[512];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int rows = 256 / C > 0 ? 256 / C : 1;
    const int c = threadIdx.x % C, row = threadIdx.x / C;
    float s1 = 0.f, s2 = 0.f;
    if (row < rows && C <= 256) {
        const float mu = mean[n * C + c], r = rstd[n * C + c], ca = a[n * C + c], cb = b[n * C + c];
        const int p0 = chunk * chunk_px, p1 = min(HW, p0 + chunk_px);
        for (int p = p0 + row; p < p1; p += rows) {
            const size_t i = ((size_t)n * HW + p) * C + c;
            const float zz = z[i];
            const float g = in_bwd_g(gin[i], zz, ca, cb, mode);
            s1 += g;
            s2 += g * ((zz - mu) * r);
        }
    }
    sh[threadIdx.x] = s1;
    sh[256 + threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < C) {
        float t1 = 0.f, t2 = 0.f;
        for (int r2 = 0; r2 < rows; ++r2) {
            t1 += sh[r2 * C + threadIdx.x];
            t2 += sh[256 + r2 * C + threadIdx.x];
        }
        float* o = partial + (((size_t)n * gridDim.x + chunk) * C + threadIdx.x) * 2;
        o[0] = t1;
        o[1] = t2;
    }
}

// S[n][c][2] = sum over chunks; dgamma[c] = sum_n S2, dbeta[c] = sum_n S1.
// One block per 16 channels: 16 lanes per channel stride over the chunk list, fixed-order combine.
__global__ __launch_bounds__(256) void in_bwd_final_kernel(const float* partial, int N, int chunks, int C, float* S,
                                                           float* dgamma, float* dbeta) {
    __shared__ float sh[512];
    const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int n = blockIdx.y;  // one block per (16 channels, sample); dgamma/dbeta are summed by the apply kernel
    (void)N;
    (void)dgamma;
    (void)dbeta;
    float t1 = 0.f, t2 = 0.f;
    if (c < C)
        for (int k = tl; k < chunks; k += 16) {
            const float* p = partial + (((size_t)n * chunks + k) * C + c) * 2;
            t1 += p[0];
            t2 += p[1];
        }
    sh[threadIdx.x] = t1;
    sh[256 + threadIdx.x] = t2;
    __syncthreads();
    if (tl == 0 && c < C) {
        float u1 = 0.f, u2 = 0.f;
        for (int j = 0; j < 16; ++j) {
            u1 += sh[j * 16 + cl];
            u2 += sh[256 + j * 16 + cl];
        }
        S[(n * C + c) * 2] = u1;
        S[(n * C + c) * 2 + 1] = u2;
    }
}

// C % 4 == 0: a thread owns 4 channels; the C/4 threads of a pixel read it as consecutive float4, a workgroup covers
// 1024/C pixels per pass and keeps UNR passes of loads in flight (the scalar kernel above waits for every pixel).
template <int UNR>
__global__ __launch_bounds__(256) void in_bwd_partial4_kernel(const float* __restrict__ gin, const float* __restrict__ z,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ a, const float* __restrict__ b, int mode,
                                                              float* __restrict__ partial, int HW, int C, int chunk_px) {
    __shared__ float sh[2 * 256 * 4];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int c4n = C >> 2;            // threads per pixel (<= 64)
    const int rows = 256 / c4n;        // pixels per pass
    const int row = threadIdx.x / c4n, c = (threadIdx.x - row * c4n) * 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
        const float4 mu = *reinterpret_cast<const float4*>(mean + n * C + c), r = *reinterpret_cast<const float4*>(rstd + n * C + c);
        const float4 ca = *reinterpret_cast<const float4*>(a + n * C + c), cb = *reinterpret_cast<const float4*>(b + n * C + c);
        const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rv[4] = {r.x, r.y, r.z, r.w};
        const float av[4] = {ca.x, ca.y, ca.z, ca.w}, bv[4] = {cb.x, cb.y, cb.z, cb.w};
        const int p0 = chunk * chunk_px, p1 = min(HW, p0 + chunk_px);
        const float* zb = z + (size_t)n * HW * C + c;
        const float* gb = gin + (size_t)n * HW * C + c;
        for (int pb = p0 + row; pb < p1; pb += UNR * rows) {
            float4 zz[UNR], gg[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int p = pb + u * rows;
                const int pc = p < p1 ? p : p0;   // clamped: a valid address, the value is discarded below
                zz[u] = *reinterpret_cast<const float4*>(zb + pc * C);
                gg[u] = *reinterpret_cast<const float4*>(gb + pc * C);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (pb + u * rows >= p1) continue;
                const float zv[4] = {zz[u].x, zz[u].y, zz[u].z, zz[u].w}, gv[4] = {gg[u].x, gg[u].y, gg[u].z, gg[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float g = in_bwd_g(gv[q], zv[q], av[q], bv[q], mode);
                    s1[q] += g;
                    s2[q] += g * ((zv[q] - muv[q]) * rv[q]);
                }
            }
        }
    }
    // sh[which][row][C]
    if (row < rows) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sh[row * C + c + q] = s1[q];
            sh[1024 + row * C + c + q] = s2[q];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float t1 = 0.f, t2 = 0.f;
        for (int r2 = 0; r2 < rows; ++r2) {
            t1 += sh[r2 * C + threadIdx.x];
            t2 += sh[1024 + r2 * C + threadIdx.x];
        }
        *reinterpret_cast<float2*>(partial + (((size_t)n * gridDim.x + chunk) * C + threadIdx.x) * 2) = make_float2(t1, t2);
    }
}

// S[n][c][2] = sum over chunks for EVERY sample, and dbeta[c] = sum_n S1, dgamma[c] = sum_n S2, in one launch:
// one block per 16 channels; the 16 lanes of a channel are dealt to the samples (16/N lanes each, N | 16), each lane
// strides over its sample's chunk list with 8 loads in flight; fixed-order combines through LDS (deterministic).
__global__ __launch_bounds__(256) void in_bwd_final_all_kernel(const float* __restrict__ partial, int N, int chunks, int C,
                                                               float* __restrict__ S, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
    __shared__ float sh[512];
    const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int L = 16 / N;            // lanes per sample
    const int n = tl / L, j = tl - n * L;
    float t1 = 0.f, t2 = 0.f;
    if (c < C) {
        const float* pn = partial + ((size_t)n * chunks * C + c) * 2;
        int k = j;
        for (; k + 7 * L < chunks; k += 8 * L) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float2*>(pn + (size_t)(k + u * L) * C * 2);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                t1 += v[u].x;
                t2 += v[u].y;
            }
        }
        for (; k < chunks; k += L) {
            const float2 v = *reinterpret_cast<const float2*>(pn + (size_t)k * C * 2);
            t1 += v.x;
            t2 += v.y;
        }
    }
    sh[threadIdx.x] = t1;
    sh[256 + threadIdx.x] = t2;
    __syncthreads();
    if (tl < N && c < C) {           // thread (tl = sample, cl): combine the sample's L lanes
        float u1 = 0.f, u2 = 0.f;
        for (int q = 0; q < L; ++q) {
            u1 += sh[(tl * L + q) * 16 + cl];
            u2 += sh[256 + (tl * L + q) * 16 + cl];
        }
        *reinterpret_cast<float2*>(S + (tl * C + c) * 2) = make_float2(u1, u2);
        sh[(tl * L) * 16 + cl] = u1;   // (own slot: no other thread reads it before the barrier)
        sh[256 + (tl * L) * 16 + cl] = u2;
    }
    __syncthreads();
    if (tl == 0 && c < C) {
        float g1 = 0.f, g2 = 0.f;
        for (int m = 0; m < N; ++m) {
            g1 += sh[(m * L) * 16 + cl];
            g2 += sh[256 + (m * L) * 16 + cl];
        }
        dbeta[c] = g1;
        dgamma[c] = g2;
    }
}

// dz for C % 4 == 0 (see in_bwd_apply_kernel): every per-channel parameter comes in as one float4, the channel index by a
// mask when C is a power of two (cmask = C - 1, else -1).  dgamma / dbeta are written by in_bwd_final_all_kernel, or here (nparams).
__global__ __launch_bounds__(256) void in_bwd_apply4_kernel(const float* __restrict__ gin, const float* __restrict__ z,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ a, const float* __restrict__ b, int mode,
                                                            const float* __restrict__ S, float* __restrict__ dz, int HW, int C,
                                                            int cmask, int nparams, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
    const float inv = 1.0f / (float)HW;
    const int n = blockIdx.y;
    const int per = HW * C;
    // nparams = N > 0: dbeta[c] = sum_n S1, dgamma[c] = sum_n S2 (fixed order) ride along on the LAST block of sample 0 (the
    // batch sizes in_bwd_final_all_kernel does not take; a launch of their own cost 9 us per layer for 2 KB of work)
    if (nparams > 0 && n == 0 && blockIdx.x == gridDim.x - 1)
        for (int c = threadIdx.x; c < C; c += 256) {
            float g1 = 0.f, g2 = 0.f;
            for (int m = 0; m < nparams; ++m) {
                const float2 v = *reinterpret_cast<const float2*>(S + (m * C + c) * 2);
                g1 += v.x;
                g2 += v.y;
            }
            dbeta[c] = g1;
            dgamma[c] = g2;
        }
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= per) return;
    const int c = cmask >= 0 ? (j & cmask) : j % C;
    const int k = n * C + c;
    const size_t base = (size_t)n * per;
    const float4 zz = *reinterpret_cast<const float4*>(z + base + j);
    const float4 gg = *reinterpret_cast<const float4*>(gin + base + j);
    const float4 a4 = *reinterpret_cast<const float4*>(a + k), b4 = *reinterpret_cast<const float4*>(b + k);
    const float4 m4 = *reinterpret_cast<const float4*>(mean + k), r4 = *reinterpret_cast<const float4*>(rstd + k);
    const float4 sA = *reinterpret_cast<const float4*>(S + 2 * k), sB = *reinterpret_cast<const float4*>(S + 2 * k + 4);
    const float zv[4] = {zz.x, zz.y, zz.z, zz.w}, gv[4] = {gg.x, gg.y, gg.z, gg.w};
    const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
    const float mv[4] = {m4.x, m4.y, m4.z, m4.w}, rv[4] = {r4.x, r4.y, r4.z, r4.w};
    const float s1v[4] = {sA.x, sA.z, sB.x, sB.z}, s2v[4] = {sA.y, sA.w, sB.y, sB.w};
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float g = in_bwd_g(gv[q], zv[q], av[q], bv[q], mode);
        const float xh = (zv[q] - mv[q]) * rv[q];
        o[q] = av[q] * (g - s1v[q] * inv - xh * s2v[q] * inv);
    }
    *reinterpret_cast<float4*>(dz + base + j) = make_float4(o[0], o[1], o[2], o[3]);
}

// dz = a * (g - S1/HW - xhat * S2/HW).  Grid (blocks over one sample, N): the sample comes from blockIdx.y and the
// index inside it is 32-bit, so the per-element cost is one 32-bit remainder instead of two 64-bit divisions; when
// C % 4 == 0 a thread handles 4 channels of one pixel with float4 loads.
template <bool VEC>
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const float* gin, const float* z, const float* mean,
                                                           const float* rstd, const float* a, const float* b, int mode,
                                                           const float* S, float* dz, int HW, int C, int N, float* dgamma,
                                                           float* dbeta) {
    const float inv = 1.0f / (float)HW;
    const int n = blockIdx.y;
    if (blockIdx.x == 0 && n == 0)  // dbeta[c] = sum_n S1, dgamma[c] = sum_n S2 (fixed order)
        for (int c = threadIdx.x; c < C; c += 256) {
            float g1 = 0.f, g2 = 0.f;
            for (int m = 0; m < N; ++m) {
                g1 += S[(m * C + c) * 2];
                g2 += S[(m * C + c) * 2 + 1];
            }
            dbeta[c] = g1;
            dgamma[c] = g2;
        }
    const size_t base = (size_t)n * HW * C;
    const int per = HW * C;
    if (VEC) {
        const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (j >= per) return;
        const int c = j % C;
        const int k = n * C + c;
        const float4 zz = *reinterpret_cast<const float4*>(z + base + j);
        const float4 gg = *reinterpret_cast<const float4*>(gin + base + j);
        const float zv[4] = {zz.x, zz.y, zz.z, zz.w}, gv[4] = {gg.x, gg.y, gg.z, gg.w};
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float g = in_bwd_g(gv[q], zv[q], a[k + q], b[k + q], mode);
            const float xh = (zv[q] - mean[k + q]) * rstd[k + q];
            o[q] = a[k + q] * (g - S[2 * (k + q)] * inv - xh * S[2 * (k + q) + 1] * inv);
        }
        *reinterpret_cast<float4*>(dz + base + j) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        for (int j = blockIdx.x * 256 + threadIdx.x; j < per; j += gridDim.x * 256) {
            const int k = n * C + j % C;
            const float zz = z[base + j];
            const float g = in_bwd_g(gin[base + j], zz, a[k], b[k], mode);
            const float xh = (zz - mean[k]) * rstd[k];
            dz[base + j] = a[k] * (g - S[2 * k] * inv - xh * S[2 * k + 1] * inv);
        }
    }
}

// ---- round 5: the per-sample sums arrive as RECORDS, the final reduction is the apply kernel's prologue.
// rec [N][T][C][2] = {sum g, sum g * xhat} per (sample, pixel block, channel): written by the epilogue of the kernel that PRODUCED gin
// (fs_wino4t_kernel.h, EPI 5 / 6: the residual input gradients -- the producer holds g in registers and reads z once, so the
// in_bwd_partial4 pass over g AND z disappears) or by in_bwd_partial4_kernel itself (same format, T = chunks).  A workgroup owns
// 256 UNR consecutive float4 of ONE sample, UNR (4 or 8) per thread: it ISSUES its 2 UNR element loads first, then sums the sample's T records in
// a fixed order while they travel (thread = (record lane, float4 column of a record row), RL = 256 / (C/2) lanes stride over the rows,
// fixed-order combine through LDS: deterministic and independent of the batch size) -- one memory latency per workgroup, as in the
// kernel without a prologue.  Block 0 of a sample leaves S[n][C][2] for in_bwd_params_kernel (dgamma / dbeta of all units, ONE launch per step).
// `span4` consecutive float4 of ONE sample per workgroup, walked in batches of 4 x 256 with the NEXT batch's loads in flight while the current one
// is computed and stored (round 5, second version: with one batch per workgroup and all workgroups resident at once the launch ran in lockstep --
// every CU loading, then every CU idle in the prologue, then every CU storing: 2.8 TB/s against 4.5 for the kernel without a prologue).
__global__ __launch_bounds__(256) void in_bwd_apply_rec_kernel(const float* __restrict__ gin, const float* __restrict__ z,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ a, const float* __restrict__ b, int mode,
                                                               const float* __restrict__ rec, int T, float* __restrict__ S_out,
                                                               float* __restrict__ dz, int HW, int C, int span4) {
    __shared__ float4 red[256];
    __shared__ float Ssh[512];   // [C][2]
    const int n = blockIdx.y, tid = threadIdx.x;
    constexpr int UNR = 4;
    const int per4 = (HW * C) >> 2;
    const int j0 = blockIdx.x * span4, j1 = min(per4, j0 + span4);
    const size_t base = (size_t)n * HW * C;
    float4 zz[UNR], gg[UNR];
    auto issue = [&](int jb, float4 (&zd)[UNR], float4 (&gd)[UNR]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = jb + u * 256;
            const int jc = j < j1 ? j : j0;   // clamped: a valid address, the value is discarded below
            zd[u] = *reinterpret_cast<const float4*>(z + base + (size_t)jc * 4);
            gd[u] = *reinterpret_cast<const float4*>(gin + base + (size_t)jc * 4);
        }
    };
    issue(j0 + tid, zz, gg);
    const bool fixed_c = (1024 % C) == 0;   // the thread's channel quad is the same for all of its elements (stride 256 float4 = 1024 floats)
    float av[4], bv[4], mv[4], rv[4];
    auto params = [&](int c) {
        const int k = n * C + c;
        const float4 a4 = *reinterpret_cast<const float4*>(a + k), b4 = *reinterpret_cast<const float4*>(b + k);
        const float4 m4 = *reinterpret_cast<const float4*>(mean + k), r4 = *reinterpret_cast<const float4*>(rstd + k);
        av[0] = a4.x, av[1] = a4.y, av[2] = a4.z, av[3] = a4.w;
        bv[0] = b4.x, bv[1] = b4.y, bv[2] = b4.z, bv[3] = b4.w;
        mv[0] = m4.x, mv[1] = m4.y, mv[2] = m4.z, mv[3] = m4.w;
        rv[0] = r4.x, rv[1] = r4.y, rv[2] = r4.z, rv[3] = r4.w;
    };
    const int c_fixed = ((j0 + tid) * 4) % C;
    if (fixed_c) params(c_fixed);
    {
        const int row4 = C >> 1;            // float4 per record row (C % 4 == 0, C <= 256)
        const int RL = 256 / row4;          // record lanes
        const int r = tid / row4, col = tid - r * row4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < RL) {
            const float4* rp = reinterpret_cast<const float4*>(rec) + (size_t)n * T * row4 + col;
            int t = r;
            for (; t + 3 * RL < T; t += 4 * RL) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = rp[(size_t)(t + u * RL) * row4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc.x += v[u].x;
                    acc.y += v[u].y;
                    acc.z += v[u].z;
                    acc.w += v[u].w;
                }
            }
            for (; t < T; t += RL) {
                const float4 v = rp[(size_t)t * row4];
                acc.x += v.x;
                acc.y += v.y;
                acc.z += v.z;
                acc.w += v.w;
            }
        }
        red[tid] = acc;
        __syncthreads();
        if (tid < row4) {
            float4 tsum = red[tid];
            for (int k = 1; k < RL; ++k) {
                const float4 q = red[k * row4 + tid];
                tsum.x += q.x;
                tsum.y += q.y;
                tsum.z += q.z;
                tsum.w += q.w;
            }
            *reinterpret_cast<float4*>(Ssh + 4 * tid) = tsum;
            if (blockIdx.x == 0) *reinterpret_cast<float4*>(S_out + (size_t)n * C * 2 + 4 * tid) = tsum;
        }
        __syncthreads();
    }
    const float inv = 1.0f / (float)HW;
    float s1v[4], s2v[4];
    auto sums = [&](int c) {
        const float4 sA = *reinterpret_cast<const float4*>(Ssh + 2 * c), sB = *reinterpret_cast<const float4*>(Ssh + 2 * c + 4);
        s1v[0] = sA.x * inv, s1v[1] = sA.z * inv, s1v[2] = sB.x * inv, s1v[3] = sB.z * inv;
        s2v[0] = sA.y * inv, s2v[1] = sA.w * inv, s2v[2] = sB.y * inv, s2v[3] = sB.w * inv;
    };
    if (fixed_c) sums(c_fixed);
    for (int jb = j0 + tid; jb < j1; jb += UNR * 256) {
        float4 zn[UNR], gn[UNR];
        const bool more = jb + UNR * 256 < j1;   // (uniform up to the block's last batch: a thread without a next batch loads clamped addresses)
        if (jb - tid + UNR * 256 < j1) issue(jb + UNR * 256, zn, gn);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int j = jb + u * 256;
            if (j >= j1) continue;
            if (!fixed_c) {
                params((j * 4) % C);
                sums((j * 4) % C);
            }
            const float zv[4] = {zz[u].x, zz[u].y, zz[u].z, zz[u].w}, gv[4] = {gg[u].x, gg[u].y, gg[u].z, gg[u].w};
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float g = in_bwd_g(gv[q], zv[q], av[q], bv[q], mode);
                const float xh = (zv[q] - mv[q]) * rv[q];
                o[q] = av[q] * (g - s1v[q] - xh * s2v[q]);
            }
            *reinterpret_cast<float4*>(dz + base + (size_t)j * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
        (void)more;
        if (jb - tid + UNR * 256 < j1) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                zz[u] = zn[u];
                gg[u] = gn[u];
            }
        }
    }
}

// dbeta[c] = sum_n S1, dgamma[c] = sum_n S2 (fixed order) of every unit whose in_bwd ran on records: one launch per step, one block per unit
__global__ __launch_bounds__(256) void in_bwd_params_kernel(InbParams p) {
    // thread = (sample lane, channel): 256 / C lanes stride over the samples with independent loads (a serial walk over N = 32 samples per
    // channel was 32 dependent L2 round trips: 22 us at batch 32), then a fixed-order combine through LDS
    __shared__ float2 sh[256];
    const InbParams::U& u = p.u[blockIdx.x];
    const int C = u.C;
    const int lanes = C >= 256 ? 1 : 256 / C;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + (int)threadIdx.x % (C < 256 ? C : 256), ln = C < 256 ? (int)threadIdx.x / C : 0;
        float g1 = 0.f, g2 = 0.f;
        if (ln < lanes && c < C)
            for (int m = ln; m < p.N; m += lanes) {
                const float2 v = *reinterpret_cast<const float2*>(u.S + ((size_t)m * C + c) * 2);
                g1 += v.x;
                g2 += v.y;
            }
        __syncthreads();
        sh[threadIdx.x] = make_float2(g1, g2);
        __syncthreads();
        if (ln == 0 && c < C) {
            for (int k = 1; k < lanes; ++k) {
                const float2 q = sh[k * C + (int)threadIdx.x];
                g1 += q.x;
                g2 += q.y;
            }
            u.dbeta[c] = g1;
            u.dgamma[c] = g2;
        }
    }
}

int in_bwd_params(const InbParams& p, hipStream_t s) {
    if (p.n <= 0) return 0;
    if (p.n > 16) return -1;
    hipLaunchKernelGGL(in_bwd_params_kernel, dim3(p.n), dim3(256), 0, s, p);
    return launch_status();
}

// how many pixels one partial-sum block of in_bwd covers
static int in_bwd_chunk_px(int N, int HW) {
    // enough blocks to cover the HBM latency: ~2k blocks of >= 128 pixels (64 measures 0.6 % slower on the step:
    // twice the partial records for the final reduction)
    int chunk_px = cdiv(HW * N, 2048);
    if (chunk_px < 128) chunk_px = 128;
    const int v = tune_int("FS_INBWD_CHUNK", 0);   // tuning aid: pixels per partial-sum block (multiples of 64 only: the scratch is sized for 64)
    if (v >= 64) chunk_px = v;
    return chunk_px;
}

// The records path: rec == nullptr -> in_bwd_partial4_kernel writes them into `scratch` first (T = chunks, when there are few enough for
// a prologue: FS_INBWD_REC_MAXT per sample); returns 1 when the shape is not taken (the caller falls back to in_bwd).  dgamma / dbeta
// come from in_bwd_params over S_out.
int in_bwd_rec(const float* gin, const float* z, const float* mean, const float* rstd, const float* a, const float* b, int mode,
               float* dz, const float* rec, int T, float* S_out, float* scratch, int N, int HW, int C, hipStream_t s) {
    if (C > 256 || C % 4 || (size_t)HW * C >= ((size_t)1 << 31) || !tune_int("FS_INBWD_REC", 1)) return 1;
    if (!rec) {
        // a coarser chunking than in_bwd's where needed (the prologue of every apply workgroup reads ALL of its sample's records): <= max_t
        // chunks per sample
        const int max_t = tune_int("FS_INBWD_REC_MAXT", 192);   // (96 / 192 / 384 measured: 225 / 217 / 217 us on the largest unit at batch 32, tools/micro_inbwd.py)
        int chunk_px = in_bwd_chunk_px(N, HW);
        if (cdiv(HW, chunk_px) > max_t) chunk_px = cdiv(cdiv(HW, max_t), 64) * 64;
        const int chunks = cdiv(HW, chunk_px);
        if (tune_int("FS_INBWD_PUNR", 8) >= 8)   // (16 loads of 16 bytes in flight per thread: the pass is latency-bound per workgroup)
            hipLaunchKernelGGL(in_bwd_partial4_kernel<8>, dim3(chunks, N), dim3(256), 0, s, gin, z, mean, rstd, a, b, mode, scratch, HW, C, chunk_px);
        else
            hipLaunchKernelGGL(in_bwd_partial4_kernel<4>, dim3(chunks, N), dim3(256), 0, s, gin, z, mean, rstd, a, b, mode, scratch, HW, C, chunk_px);
        rec = scratch;
        T = chunks;
    }
    const int per4 = (HW * C) >> 2;
    // ~1024 workgroups per launch (four per CU, all resident): each walks per4 / (1024 / N) float4 of its sample in pipelined batches of 1024
    int bps = tune_int("FS_INBWD_APPLY_WGS", 1024) / N;
    if (bps < 1) bps = 1;
    int span4 = cdiv(cdiv(per4, bps), 1024) * 1024;
    if (span4 < 1024) span4 = 1024;
    hipLaunchKernelGGL(in_bwd_apply_rec_kernel, dim3(cdiv(per4, span4), N), dim3(256), 0, s, gin, z, mean, rstd, a, b, mode, rec, T, S_out, dz, HW, C, span4);
    return launch_status();
}

// scratch: N*chunks*C*2 + N*C*2 floats
int in_bwd(const float* gin, const float* z, const float* mean, const float* rstd, const float* a, const float* b, int mode,
           float* dz, float* dgamma, float* dbeta, float* scratch, int N, int HW, int C, hipStream_t s) {
    if (C > 256) return -1;
    const int chunk_px = in_bwd_chunk_px(N, HW);
    const int chunks = cdiv(HW, chunk_px);
    float* partial = scratch;
    float* S = scratch + (size_t)N * chunks * C * 2;
    if (C % 4 == 0) {
        int nparams = 0;
        hipLaunchKernelGGL(in_bwd_partial4_kernel<4>, dim3(chunks, N), dim3(256), 0, s, gin, z, mean, rstd, a, b, mode, partial, HW,
                           C, chunk_px);
        if (N <= 16 && 16 % N == 0) {
            hipLaunchKernelGGL(in_bwd_final_all_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, partial, N, chunks, C, S, dgamma, dbeta);
        } else {
            hipLaunchKernelGGL(in_bwd_final_kernel, dim3(cdiv(C, 16), N), dim3(256), 0, s, partial, N, chunks, C, S, dgamma, dbeta);
            nparams = N;
        }
        const int per = HW * C;
        hipLaunchKernelGGL(in_bwd_apply4_kernel, dim3(cdiv(per / 4, 256), N), dim3(256), 0, s, gin, z, mean, rstd, a, b, mode,
                           S, dz, HW, C, (C & (C - 1)) == 0 ? C - 1 : -1, nparams, dgamma, dbeta);
        return launch_status();
    }
    hipLaunchKernelGGL(in_bwd_partial_kernel, dim3(chunks, N), dim3(256), 0, s, gin, z, mean, rstd, a, b, mode, partial, HW,
                       C, chunk_px);
    hipLaunchKernelGGL(in_bwd_final_kernel, dim3(cdiv(C, 16), N), dim3(256), 0, s, partial, N, chunks, C, S, dgamma, dbeta);
    const int per = HW * C;
    hipLaunchKernelGGL(in_bwd_apply_kernel<false>, dim3(min(2048, cdiv(per, 256)), N), dim3(256), 0, s, gin, z, mean, rstd,
                       a, b, mode, S, dz, HW, C, N, dgamma, dbeta);
    return launch_status();
}
size_t in_bwd_scratch_floats(int N, int HW, int C) { return (size_t)N * cdiv(HW, 64) * C * 2 + (size_t)N * C * 2; }

// ---------------------------------------------------------------- VGG: max-pool + gradient routing
// tf.nn.max_pool 2x2/2 SAME (reference libs/vgg16.py:63-67): out = ceil(in/2), padded cells never win.
// grid (blocks over one output row, Ho, N); a thread owns 4 channels of one output pixel (C % 4 == 0 in VGG16)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C,
                                                      int Ho, int Wo) {
    const int c4n = C >> 2;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Wo * c4n) return;
    const int ox = j / c4n, c = (j - ox * c4n) * 4;
    const int oy = blockIdx.y, n = blockIdx.z;
    const int y0 = 2 * oy, x0 = 2 * ox;
    const float* base = x + (((size_t)n * H + y0) * W + x0) * C + c;
    float4 m = *reinterpret_cast<const float4*>(base);
    auto mx = [&](const float* p) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
    };
    if (x0 + 1 < W) mx(base + C);
    if (y0 + 1 < H) {
        mx(base + (size_t)W * C);
        if (x0 + 1 < W) mx(base + (size_t)W * C + C);
    }
    *reinterpret_cast<float4*>(y + (((size_t)n * Ho + oy) * Wo + ox) * C + c) = m;
}

int maxpool(const float* x, float* y, int N, int H, int W, int C, hipStream_t s) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    if (C % 4) return -1;
    hipLaunchKernelGGL(maxpool_kernel, dim3(cdiv(Wo * (C / 4), 256), Ho, N), dim3(256), 0, s, x, y, H, W, C, Ho, Wo);
    return launch_status();
}

// d_pre[n,y,x,c] = (route(d_above) + d_tap) * (out > 0)
//   pooled=0: d_above has the shape of `out`;   pooled=1: d_above is the gradient of max_pool(out)
//   and goes to the FIRST maximum of each window (TF MaxPoolGrad).  d_above / d_tap may be null.
// grid (blocks over one row, H, N); a thread owns 4 channels of one pixel.
__global__ __launch_bounds__(256) void vgg_bwd_route_kernel(const float* __restrict__ out, const float* __restrict__ d_above,
                                                            const float* __restrict__ d_tap, int pooled, float* __restrict__ d_pre,
                                                            int H, int W, int C) {
    const int c4n = C >> 2;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= W * c4n) return;
    const int x = j / c4n, c = (j - x * c4n) * 4;
    const int y = blockIdx.y, n = blockIdx.z;
    const size_t i = (((size_t)n * H + y) * W + x) * C + c;
    const float4 v4 = *reinterpret_cast<const float4*>(out + i);
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (d_tap) {
        const float4 t = *reinterpret_cast<const float4*>(d_tap + i);
        g[0] = t.x;
        g[1] = t.y;
        g[2] = t.z;
        g[3] = t.w;
    }
    if (d_above) {
        if (!pooled) {
            const float4 t = *reinterpret_cast<const float4*>(d_above + i);
            g[0] += t.x;
            g[1] += t.y;
            g[2] += t.z;
            g[3] += t.w;
        } else {
            const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
            const int y0 = y & ~1, x0 = x & ~1;
            const int me = (y - y0) * 2 + (x - x0);
            bool win[4] = {true, true, true, true};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int yy = y0 + (e >> 1), xx = x0 + (e & 1);
                if (e == me || yy >= H || xx >= W) continue;
                const float4 o4 = *reinterpret_cast<const float4*>(out + (((size_t)n * H + yy) * W + xx) * C + c);
                const float o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (e < me ? o[q] >= v[q] : o[q] > v[q]) win[q] = false;
            }
            const float4 da = *reinterpret_cast<const float4*>(d_above + (((size_t)n * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c);
            const float dav[4] = {da.x, da.y, da.z, da.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (win[q]) g[q] += dav[q];
        }
    }
    *reinterpret_cast<float4*>(d_pre + i) =
        make_float4(v[0] > 0.f ? g[0] : 0.f, v[1] > 0.f ? g[1] : 0.f, v[2] > 0.f ? g[2] : 0.f, v[3] > 0.f ? g[3] : 0.f);
}

int vgg_bwd_route(const float* out, const float* d_above, const float* d_tap, int pooled, float* d_pre, int N, int H, int W,
                  int C, hipStream_t s) {
    if (C % 4) return -1;
    hipLaunchKernelGGL(vgg_bwd_route_kernel, dim3(cdiv(W * (C / 4), 256), H, N), dim3(256), 0, s, out, d_above, d_tap, pooled,
                       d_pre, H, W, C);
    return launch_status();
}

// ---------------------------------------------------------------- losses
// partial[b] = sum (x - t)^2 over the block's elements; grad = gscale*(x - t) (optional).
// t index = i % t_period (style targets broadcast over the batch; reference losses.py:61-64).
// grid (blocks per period, periods): the target index is the offset inside the period -- no 64-bit modulo per element
__global__ __launch_bounds__(256) void sqdiff_kernel(const float* __restrict__ x, const float* __restrict__ t, size_t t_period,
                                                     float gscale, float* __restrict__ grad, float* __restrict__ partial) {
    __shared__ float sh[4];
    float acc = 0.f;
    const size_t base = (size_t)blockIdx.y * t_period;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < t_period; j += (size_t)gridDim.x * 256) {
        const float d = x[base + j] - t[j];
        acc = fmaf(d, d, acc);
        if (grad) grad[base + j] = gscale * d;
    }
    const float tot = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

// out[0] (+)= scale * sum(partial[0..n))  -- single block, fixed order
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* partial, int n, float scale, float* out,
                                                           int accumulate) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    const float tot = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + scale * tot;
}

// loss_out (+)= lscale*sum((x-t)^2);  grad = gscale*(x-t).  scratch: >= 1024 floats.
int sqdiff_loss(const float* x, const float* t, size_t t_period, size_t total, float lscale, float gscale, float* grad,
                float* loss_out, int accumulate, float* scratch, hipStream_t s) {
    const int periods = (int)(total / t_period);  // (total is a whole number of periods: the batch)
    if (periods < 1 || periods > 1024 || (size_t)periods * t_period != total) return -1;  // scratch holds 1024 partials
    int bx = (int)min((size_t)(1024 / periods > 0 ? 1024 / periods : 1), (t_period + 255) / 256);
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(sqdiff_kernel, dim3(bx, periods), dim3(256), 0, s, x, t, t_period, gscale, grad, scratch);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, scratch, bx * periods, lscale, loss_out, accumulate);
    return launch_status();
}

// The same element pass without the final sum: partial[0 .. *n_partial) are left for loss_finish (fs_perceptual_loss: every loss term of a
// step is summed by ONE launch at the end instead of a sum_partials launch per term).  partial: room for 1024 floats.
int sqdiff_partials(const float* x, const float* t, size_t t_period, size_t total, float gscale, float* grad, float* partial, int* n_partial,
                    hipStream_t s) {
    const int periods = (int)(total / t_period);
    if (periods < 1 || periods > 1024 || (size_t)periods * t_period != total) return -1;
    int bx = (int)min((size_t)(1024 / periods > 0 ? 1024 / periods : 1), (t_period + 255) / 256);
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(sqdiff_kernel, dim3(bx, periods), dim3(256), 0, s, x, t, t_period, gscale, grad, partial);
    *n_partial = bx * periods;
    return launch_status();
}

// losses = {total, content, style, beta * tv} (reference train.py:184) from the partial sums of every term: job j adds
// scale_j * sum(partial_j[0 .. n_j)) to losses[slot_j]; one workgroup, fixed order (deterministic); all four scalars are WRITTEN (no
// clear beforehand, no read-modify-write across launches).
__global__ __launch_bounds__(256) void loss_finish_kernel(LossFinish f) {
    __shared__ float sh[4];
    float tot[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < f.n; ++j) {
        float acc = 0.f;
        for (int i = threadIdx.x; i < f.job[j].n; i += 256) acc += f.job[j].partial[i];
        const float v = block_sum(acc, sh) * f.job[j].scale;
        const int slot = f.job[j].slot;
        tot[1] += slot == 1 ? v : 0.f;
        tot[2] += slot == 2 ? v : 0.f;
        tot[3] += slot == 3 ? v : 0.f;
    }
    if (threadIdx.x == 0) {
        f.losses[1] = tot[1];
        f.losses[2] = tot[2];
        f.losses[3] = tot[3];
        f.losses[0] = tot[1] + tot[2] + tot[3];
    }
}

int loss_finish(const LossFinish& f, hipStream_t s) {
    if (f.n < 0 || f.n > LossFinish::kMax) return -1;
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, s, f);
    return launch_status();
}

// TV loss (reference losses.py:70-97): sum of squared forward differences along H and W, and
// its gradient scaled by gscale, ACCUMULATED into grad (grad += gscale * dTV/dx) when grad != null.
__global__ __launch_bounds__(256) void tv_kernel(const float* x, int H, int W, int C, size_t total, float gscale,
                                                 float* grad, float* partial) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t pix = i / C;
        const int xx = (int)(pix % W);
        pix /= W;
        const int yy = (int)(pix % H);
        const float v = x[i];
        float g = 0.f;
        if (yy + 1 < H) {
            const float d = v - x[i + (size_t)W * C];
            acc = fmaf(d, d, acc);
            g += 2.f * d;
        }
        if (yy > 0) g -= 2.f * (x[i - (size_t)W * C] - v);
        if (xx + 1 < W) {
            const float d = v - x[i + C];
            acc = fmaf(d, d, acc);
            g += 2.f * d;
        }
        if (xx > 0) g -= 2.f * (x[i - C] - v);
        if (grad) grad[i] += gscale * g;
    }
    const float tot = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

int tv_partials(const float* x, int N, int H, int W, int C, float gscale, float* grad, float* partial, int* n_partial, hipStream_t s) {
    const size_t total = (size_t)N * H * W * C;
    const int blocks = (int)min((size_t)1024, (total + 255) / 256);
    hipLaunchKernelGGL(tv_kernel, dim3(blocks), dim3(256), 0, s, x, H, W, C, total, gscale, grad, partial);
    *n_partial = blocks;
    return launch_status();
}

int tv_loss(const float* x, int N, int H, int W, int C, float lscale, float gscale, float* grad, float* loss_out,
            float* scratch, hipStream_t s) {
    const size_t total = (size_t)N * H * W * C;
    const int blocks = (int)min((size_t)1024, (total + 255) / 256);
    hipLaunchKernelGGL(tv_kernel, dim3(blocks), dim3(256), 0, s, x, H, W, C, total, gscale, grad, scratch);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, scratch, blocks, lscale, loss_out, 0);
    return launch_status();
}

// out = alpha * (x - t[i % period])     (style: S = coef*(G - Gt), the filter of the Gram backward)
__global__ __launch_bounds__(256) void axpby_kernel(const float* x, const float* y, float a, float b, float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

int axpby(const float* x, const float* y, float a, float b, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)min((size_t)4096, (n + 255) / 256)), dim3(256), 0, s, x, y, a, b, out, n);
    return launch_status();
}

// ---------------------------------------------------------------- TF-style Adam
// tf.train.AdamOptimizer (reference train.py:203): theta -= lr_t * m / (sqrt(v) + eps), with
// lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the host -- epsilon outside the bias correction.
__global__ __launch_bounds__(256) void adam_tf_kernel(float* p, const float* g, float* m, float* v, size_t n, float lr_t,
                                                      float b1, float b2, float eps) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

int adam_tf(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps, hipStream_t s) {
    hipLaunchKernelGGL(adam_tf_kernel, dim3((unsigned)min((size_t)2048, (n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr_t,
                       b1, b2, eps);
    return launch_status();
}

// ---------------------------------------------------------------- filter re-layouts
// dgrad filter of a stride-1/2 conv: out[kh,kw,co,ci] = w[KH-1-kh, KW-1-kw, ci, co]
__global__ __launch_bounds__(256) void wt_flip_transpose_kernel(const float* w, float* out, int KH, int KW, int Ci, int Co) {
    const int total = KH * KW * Ci * Co;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int ci = i % Ci;
        int r = i / Ci;
        const int co = r % Co;
        r /= Co;
        const int kw = r % KW, kh = r / KW;
        out[i] = w[(((KH - 1 - kh) * KW + (KW - 1 - kw)) * Ci + ci) * Co + co];
    }
}

int wt_flip_transpose(const float* w, float* out, int KH, int KW, int Ci, int Co, hipStream_t s) {
    hipLaunchKernelGGL(wt_flip_transpose_kernel, dim3(cdiv(KH * KW * Ci * Co, 256)), dim3(256), 0, s, w, out, KH, KW, Ci, Co);
    return launch_status();
}

// Phase-collapsed resize-conv (reference im_transf_net.py:122-155: NEAREST x4 then 3x3 stride-2
// SAME == per output parity (a,b) a 2x2-tap conv on the low-res input):
//   weff[dy,dx,ci,(a*2+b)*Co+co] = sum_{kh in R(a,dy), kw in R(b,dx)} w[kh,kw,ci,co]
//   R(0,0)={0,1,2}  R(0,1)={}  R(1,0)={0,1}  R(1,1)={2}
__device__ __forceinline__ bool up_in_R(int a, int d, int k) {
    if (a == 0) return d == 0;
    return d == 0 ? k < 2 : k == 2;
}
__global__ __launch_bounds__(256) void wt_upconv_fwd_kernel(const float* w, float* weff, int Ci, int Co) {
    const int total = 4 * Ci * 4 * Co;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int j = i % (4 * Co);
        int r = i / (4 * Co);
        const int ci = r % Ci;
        const int tap = r / Ci;
        const int dy = tap >> 1, dx = tap & 1;
        const int q = j / Co, co = j % Co, a = q >> 1, b = q & 1;
        float acc = 0.f;
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw)
                if (up_in_R(a, dy, kh) && up_in_R(b, dx, kw)) acc += w[((kh * 3 + kw) * Ci + ci) * Co + co];
        weff[i] = acc;
    }
}

// its dgrad as a 3x3 stride-2 conv over dY (pad 1 before):  v[t,s,co,ci], t,s in {0,1,2} <-> {-1,0,1}
//   row sets: t=0 -> {2}, t=1 -> {0,1,2}, t=2 -> {0,1}
__device__ __forceinline__ bool up_in_V(int t, int k) { return t == 0 ? k == 2 : (t == 1 ? true : k < 2); }
__global__ __launch_bounds__(256) void wt_upconv_dgrad_kernel(const float* w, float* v, int Ci, int Co) {
    const int total = 9 * Co * Ci;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int ci = i % Ci;
        int r = i / Ci;
        const int co = r % Co;
        const int tap = r / Co;
        const int t = tap / 3, s2 = tap % 3;
        float acc = 0.f;
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw)
                if (up_in_V(t, kh) && up_in_V(s2, kw)) acc += w[((kh * 3 + kw) * Ci + ci) * Co + co];
        v[i] = acc;
    }
}

// fold the collapsed filter gradient back:  dw[kh,kw,ci,co] = sum over (a,dy)∋kh,(b,dx)∋kw of dweff
__global__ __launch_bounds__(256) void wt_upconv_wgrad_fold_kernel(const float* dweff, float* dw, int Ci, int Co) {
    const int total = 9 * Ci * Co;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int co = i % Co;
        int r = i / Co;
        const int ci = r % Ci;
        const int tap = r / Ci;
        const int kh = tap / 3, kw = tap % 3;
        float acc = 0.f;
        for (int a = 0; a < 2; ++a)
            for (int dy = 0; dy < 2; ++dy)
                for (int b = 0; b < 2; ++b)
                    for (int dx = 0; dx < 2; ++dx)
                        if (up_in_R(a, dy, kh) && up_in_R(b, dx, kw))
                            acc += dweff[(((dy * 2 + dx) * Ci + ci) * 4 + (a * 2 + b)) * Co + co];
        dw[i] = acc;
    }
}

// one launch for a list of re-layout jobs; element formulas identical to the single-job kernels above
// (WT_FOLD5FWD: wf[kh][b][ci][16] from w[9][9][Ci][3], see fs_fold.hip)
__global__ __launch_bounds__(256) void wt_batch_kernel(WtBatch b) {
    const WtJob& q = b.j[blockIdx.y];
    const float* __restrict__ w = q.src;
    const int Ci = q.Ci, Co = q.Co;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < q.total; i += gridDim.x * 256) {
        float acc = 0.f;
        if (q.kind == WT_FLIPT) {
            const int ci = i % Ci;
            int r = i / Ci;
            const int co = r % Co;
            r /= Co;
            const int kw = r % q.KW, kh = r / q.KW;
            acc = w[(((q.KH - 1 - kh) * q.KW + (q.KW - 1 - kw)) * Ci + ci) * Co + co];
        } else if (q.kind == WT_UPFWD) {
            const int j = i % (4 * Co);
            const int r = i / (4 * Co);
            const int ci = r % Ci, tap = r / Ci;
            const int dy = tap >> 1, dx = tap & 1;
            const int qq = j / Co, co = j % Co, a = qq >> 1, bb = qq & 1;
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw)
                    if (up_in_R(a, dy, kh) && up_in_R(bb, dx, kw)) acc += w[((kh * 3 + kw) * Ci + ci) * Co + co];
        } else if (q.kind == WT_UPDGRAD) {
            const int ci = i % Ci;
            const int r = i / Ci;
            const int co = r % Co, tap = r / Co;
            const int t = tap / 3, s2 = tap % 3;
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw)
                    if (up_in_V(t, kh) && up_in_V(s2, kw)) acc += w[((kh * 3 + kw) * Ci + ci) * Co + co];
        } else if (q.kind == WT_S2DGRAD) {
            // input gradient of a 3x3 stride-2 SAME conv, phase-decomposed: a 2x2-tap conv over dz whose 4*Ci output
            // channels are the 4 parities (a,b) of the gradient pixel (2j+a, 2i+b), pixel-shuffle stored.
            //   dx[2j+a] = sum_k dz[j + (a+p-k)/2] w[k]   for a+p-k even   (p = top/left SAME pad of the forward, 0 or 1)
            // out[tap=(dy,dx)][co][(a*2+b)*Ci + ci];  job fields KH/KW carry pad_t/pad_l
            const int j = i % (4 * Ci);
            const int r = i / (4 * Ci);
            const int co = r % Co, tap = r / Co;
            const int dy = tap >> 1, dx = tap & 1;
            const int qq = j / Ci, ci = j - qq * Ci, pa = qq >> 1, pb = qq & 1;
            auto ksel = [](int p, int d, int par) {  // which filter row feeds tap d for parity par (-1: none)
                if (p == 0) return d == 0 ? (par == 0 ? 2 : -1) : (par == 0 ? 0 : 1);
                return d == 0 ? (par == 0 ? 1 : 2) : (par == 0 ? -1 : 0);
            };
            const int kh = ksel(q.KH, dy, pa), kw = ksel(q.KW, dx, pb);
            acc = (kh >= 0 && kw >= 0) ? w[((kh * 3 + kw) * Ci + ci) * Co + co] : 0.f;
        } else {  // WT_FOLD5FWD
            const int j = i & 15;
            int r = i >> 4;
            const int ci = r % Ci;
            r /= Ci;
            const int bb = r & 1, kh = r >> 1;
            const int v = j / 3, co = j - v * 3, kw = 5 * bb + v;
            acc = (j < 15 && kw < 9) ? w[((kh * 9 + kw) * Ci + ci) * 3 + co] : 0.f;
        }
        q.dst[i] = acc;
    }
}

int wt_batch(const WtBatch& b, hipStream_t s) {
    if (b.n <= 0) return 0;
    int mx = 0;
    for (int k = 0; k < b.n; ++k)
        if (b.j[k].total > mx) mx = b.j[k].total;
    hipLaunchKernelGGL(wt_batch_kernel, dim3(cdiv(mx, 256), b.n), dim3(256), 0, s, b);
    return launch_status();
}

int wt_upconv_fwd(const float* w, float* weff, int Ci, int Co, hipStream_t s) {
    hipLaunchKernelGGL(wt_upconv_fwd_kernel, dim3(cdiv(16 * Ci * Co, 256)), dim3(256), 0, s, w, weff, Ci, Co);
    return launch_status();
}
int wt_upconv_dgrad(const float* w, float* v, int Ci, int Co, hipStream_t s) {
    hipLaunchKernelGGL(wt_upconv_dgrad_kernel, dim3(cdiv(9 * Ci * Co, 256)), dim3(256), 0, s, w, v, Ci, Co);
    return launch_status();
}
int wt_upconv_wgrad_fold(const float* dweff, float* dw, int Ci, int Co, hipStream_t s) {
    hipLaunchKernelGGL(wt_upconv_wgrad_fold_kernel, dim3(cdiv(9 * Ci * Co, 256)), dim3(256), 0, s, dweff, dw, Ci, Co);
    return launch_status();
}

}  // namespace fs

namespace fs {
// on-load affine of the VGG input: images - [123.68,116.779,103.939] (reference libs/vgg16.py:41-42)
__global__ void vgg_consts_kernel(float* ab) {
    if (threadIdx.x == 0) {
        ab[0] = ab[1] = ab[2] = 1.0f;
        ab[3] = 0.f;
        ab[4] = -123.68f;
        ab[5] = -116.779f;
        ab[6] = -103.939f;
        ab[7] = 0.f;
    }
}
int vgg_consts(float* ab, hipStream_t s) {
    hipLaunchKernelGGL(vgg_consts_kernel, dim3(1), dim3(64), 0, s, ab);
    return launch_status();
}
// n <= 64 words = 0 (instead of hipMemsetAsync: see fs_perceptual_loss)
__global__ void zero_words_kernel(unsigned* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0u;
}
int zero_words(void* p, int n, hipStream_t s) {
    if (n < 0 || n > 64) return -1;
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned*>(p), n);
    return launch_status();
}
}  // namespace fs
