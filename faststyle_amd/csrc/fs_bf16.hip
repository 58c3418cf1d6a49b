// bf16 mixed-precision INFERENCE path of the transform net (BASELINE config 5: 1080p, batch 8 per GPU).
//
// Activations are stored in HBM as bf16 (raw conv outputs z, residual sums h), weights are packed to bf16 once
// per call, every contraction runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, and the instance-norm
// statistics are taken from the fp32 accumulators before rounding.  At 16x the fp32 MFMA rate the convs stop
// being matrix-core bound: the design goal of this kernel is bytes, not FLOPs -- 2-byte activations in HBM and
// LDS, 16-byte LDS fragment reads, producer instance-norm + ReLU folded into the staging load as in the fp32 path.
//
// The image-facing ends stay fp32: the input image [N,H,W,3] is read as fp32 (reflect-40 fused, packed to
// 4-channel bf16 pixels in LDS); the kw-folded output layer writes its 16 virtual channels as bf16, and the 5-term
// fold, the last instance norm and the tanh run in fp32 as in the fp32 path (fs_fold.hip, fs_elem.hip).
//
// Reference: im_transf_net.py:14-75 (create_net), same layer semantics as fs_conv.hip / fs_tnet.hip.
#include "fs_bf16.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace fs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(unsigned short, (__bf16)f);  // v_cvt_pk_bf16_f32: round to nearest even
#else
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
#endif
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }

// x / d for 0 <= x < 2^22 through a float reciprocal (exact: (x + 0.5)/d stays >= 0.5/d away from an integer while
// the float error is < x * 2^-23 / d); an integer division costs ~35 instructions, and the persistent tile loop
// would otherwise spend more time on index arithmetic than on its few dozen MFMAs
__device__ __forceinline__ int fdiv(int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); }

__device__ __forceinline__ bool bsrc_coord(int mode, int refl, int v, int n_src, int& s) {
    if (mode == SRC_REFLECT) {
        if (v < 0 || v >= n_src + 2 * refl) return false;
        s = v - refl;
        if (s < 0) s = -s;
        if (s >= n_src) s = 2 * (n_src - 1) - s;
        return true;
    }
    s = v;
    return v >= 0 && v < n_src;
}

// ---------------------------------------------------------------------------------------------- conv kernel
// 4 waves; workgroup tile = (4*WM*32 pixels) x (WN*32 channels); per input-channel chunk CC (16 or 32) the patch
// with halo is staged as [pixel][CC+8] bf16 and the filter chunk as [tap][co][CC+8] bf16 (the +8 pad makes the
// 16-byte fragment reads of 16 consecutive lanes hit disjoint banks).  The next chunk's global loads are issued
// before the MFMA sweep of the current one (register prefetch), two barriers per chunk, one LDS stage, so two
// workgroups fit a CU.  C4 = the 3-channel image layer: pixels are 4-channel bf16 (8 bytes), K runs over
// (12 taps of a kernel row) x 4, i.e. 3 MFMA k-steps per kernel row, the whole 9x9 filter in one chunk.
// MFMA sweep over (tap, 16-channel k-step) of one staged chunk, software-pipelined through two register sets: the
// 16-byte LDS fragment reads of step i+1 are issued before the MFMAs of step i.  (Left to the compiler the loop
// becomes "ds_read, s_waitcnt, mfma" per step -- and a bf16 MFMA is only 32 cycles, a fraction of the LDS latency.)
template <int WM, int WN>
__device__ __forceinline__ void sweep_bf16(f32x16 (&acc)[WM][WN], const unsigned short* patch, const unsigned short* wl,
                                           const int (&laneA)[WM], int laneB, int KH, int KW, int dil_x, int PW, int PP, int WP,
                                           int BN, int nks) {
    const int nsteps = KH * KW * nks;
    int kh = 0, kw = 0, ks = 0, g = 0;
    auto load = [&](bf16x8 (&af)[WM], bf16x8 (&bfr)[WN]) {
        const int toff = (kh * PW + kw * dil_x) * PP + ks * 16;
        const int woff = g * BN * WP + ks * 16 + laneB;
#pragma unroll
        for (int m = 0; m < WM; ++m) af[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(patch + laneA[m] + toff));
#pragma unroll
        for (int nn = 0; nn < WN; ++nn) bfr[nn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wl + woff + nn * 32 * WP));
        if (++ks == nks) {
            ks = 0;
            ++g;
            if (++kw == KW) {
                kw = 0;
                ++kh;
            }
        }
    };
    auto mma = [&](const bf16x8 (&af)[WM], const bf16x8 (&bfr)[WN]) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int nn = 0; nn < WN; ++nn) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[nn], acc[m][nn], 0, 0, 0);
    };
    bf16x8 a0[WM], b0[WN], a1[WM], b1[WN];
    if (nsteps > 0) load(a0, b0);
    int q = 0;
    for (; q + 2 <= nsteps; q += 2) {
        load(a1, b1);
        mma(a0, b0);
        if (q + 2 < nsteps) load(a0, b0);
        mma(a1, b1);
    }
    if (q < nsteps) mma(a0, b0);
}

// Per-tile instance-norm partial {mean, M2, count} per channel, from the fp32 accumulators, in ONE pass: every wave
// sums d = z - pivot and d^2 over its rows (pivot = the wave's first row of that channel: the shifted-data form
// keeps sum(d^2) - sum(d)^2/n free of cancellation), the four wave results are merged with Chan's update by one
// thread per channel.  One barrier; interior tiles skip the per-row validity test.  wst: [4][BN][4] floats of LDS.
template <int WM, int WN>
__device__ __forceinline__ void tile_stats_write(const ConvBArgs& a, const f32x16 (&acc)[WM][WN], float* wst, int th_valid,
                                                 int tw_valid) {
    constexpr int BN = WN * 32;
    const ConvBPlan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 31, kq = lane >> 5;
    const int tile_px = p.TH * p.TW;
    const bool full = th_valid == p.TH && tw_valid == p.TW && tile_px == 4 * WM * 32;
    const float inv_tw = 1.0f / (float)p.TW;
    unsigned okmask[WM];  // bit r: accumulator row r of tile m is a real pixel
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        okmask[m] = 0xFFFFu;
        if (!full) {
            okmask[m] = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = (wave * WM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                const int py = (int)(((float)t + 0.5f) * inv_tw), px = t - py * p.TW;
                if (t < tile_px && py < th_valid && px < tw_valid) okmask[m] |= 1u << r;
            }
        }
    }
#pragma unroll
    for (int nn = 0; nn < WN; ++nn) {
        const float piv = __shfl(acc[0][nn][0], lm);
        float sd = 0.f, sq = 0.f, cn = 0.f;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[m][nn][r] - piv;
                if (full || (okmask[m] >> r) & 1u) {
                    sd += d;
                    sq = fmaf(d, d, sq);
                    cn += 1.f;
                }
            }
        sd += __shfl_xor(sd, 32);
        sq += __shfl_xor(sq, 32);
        cn += __shfl_xor(cn, 32);
        if (lane < 32) {
            float* o = wst + ((wave * BN + nn * 32 + lane) << 2);
            o[0] = sd;
            o[1] = sq;
            o[2] = cn;
            o[3] = piv;
        }
    }
}
// second half, after a workgroup barrier: one thread per channel merges the four wave results
template <int WN>
__device__ __forceinline__ void tile_stats_finish(const ConvBArgs& a, const float* wst, size_t tile, int co0) {
    constexpr int BN = WN * 32;
    const int tid = threadIdx.x;
    if (tid < BN && co0 + tid < a.Cout) {
        float cnt = 0.f, mu = 0.f, m2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* o = wst + ((w * BN + tid) << 2);
            const float cb = o[2];
            if (cb > 0.f) {
                const float mb = o[3] + o[0] / cb, qb = fmaxf(o[1] - o[0] * o[0] / cb, 0.f);
                const float nn2 = cnt + cb, dlt = mb - mu, rr = cb / nn2;
                mu += dlt * rr;
                m2 += qb + dlt * dlt * cnt * rr;
                cnt = nn2;
            }
        }
        float* st = a.stats + (tile * a.Cout + co0 + tid) * 3;
        st[0] = mu;
        st[1] = m2;
        st[2] = cnt;
    }
}

// Tile store through LDS: the accumulator layout gives a lane ONE channel of 16 pixel rows, i.e. 2-byte scattered
// global stores; staging the tile as [pixel][BN] lets every thread write 16 contiguous bytes (8 bf16 channels or 4
// fp32) of one pixel -- whole 64/128-byte pixel rows per wavefront.  `stg` must not alias live LDS data.
template <int WM, int WN>
__device__ __forceinline__ void store_tile_write(const ConvBArgs& a, const f32x16 (&acc)[WM][WN], void* stg) {
    constexpr int BN = WN * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 31, kq = lane >> 5;
    unsigned short* sb = static_cast<unsigned short*>(stg);
    float* sf = static_cast<float*>(stg);
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = (wave * WM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
#pragma unroll
            for (int nn = 0; nn < WN; ++nn) {
                if (a.y_f32)
                    sf[t * BN + nn * 32 + lm] = acc[m][nn][r];
                else
                    sb[t * BN + nn * 32 + lm] = f2bf(acc[m][nn][r]);
            }
        }
}
// second half, after a workgroup barrier: 16-byte cooperative stores
template <int WN>
__device__ __forceinline__ void store_tile_finish(const ConvBArgs& a, const void* stg, int n, int ty0, int tx0, int co0,
                                                  int th_valid, int tw_valid) {
    constexpr int BN = WN * 32;
    const ConvBPlan& p = a.p;
    const int tid = threadIdx.x;
    const unsigned short* sb = static_cast<const unsigned short*>(stg);
    const float* sf = static_cast<const float*>(stg);
    const int gsz = a.y_f32 ? 4 : 8;  // channels per 16-byte granule
    const int gpp = BN / gsz;         // granules per staged pixel (a power of two)
    const int gsh = gpp == 4 ? 2 : (gpp == 8 ? 3 : 4);
    const float inv_tw = 1.0f / (float)p.TW;
    const int Cr = a.shuffle ? a.Cout >> 2 : a.Cout;
    const size_t img = (size_t)n * a.Ho * a.Wo * a.Cout;
    const int tile_px = p.TH * p.TW;
    for (int e = tid; e < tile_px * gpp; e += 256) {
        const int pix = e >> gsh, g = e & (gpp - 1);
        const int py = fdiv(pix, inv_tw), px = pix - py * p.TW;
        const int co = co0 + g * gsz;
        if (py >= th_valid || px >= tw_valid || co >= a.Cout) continue;
        const int oy = ty0 + py, ox = tx0 + px;
        size_t o;
        if (a.shuffle) {
            const int q = fdiv(co, 1.0f / (float)Cr), cof = co - q * Cr;
            o = ((size_t)(2 * oy + (q >> 1)) * (2 * a.Wo) + 2 * ox + (q & 1)) * Cr + cof;
        } else {
            o = ((size_t)oy * a.Wo + ox) * a.Cout + co;
        }
        if (a.y_f32)
            *reinterpret_cast<uint4*>(static_cast<float*>(a.y) + img + o) = *reinterpret_cast<const uint4*>(sf + pix * BN + g * 4);
        else
            *reinterpret_cast<uint4*>(static_cast<unsigned short*>(a.y) + img + o) =
                *reinterpret_cast<const uint4*>(sb + pix * BN + g * 8);
    }
}

// (1) chunked kernel: layers whose input channels span several chunks (Cin = 32/64).  One tile per workgroup.
template <int WM, int WN, bool C4>
__global__ __launch_bounds__(256) void conv_bf16_chunked_kernel(ConvBArgs a) {
    constexpr int BN = WN * 32;
    HIP_DYNAMIC_SHARED(float, smem_f)
    unsigned short* smem = reinterpret_cast<unsigned short*>(smem_f);
    const ConvBPlan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = p.tiles_y * p.tiles_x;
    const int n = blockIdx.x / tiles, tr = blockIdx.x % tiles;
    const int ty0 = (tr / p.tiles_x) * p.TH, tx0 = (tr % p.tiles_x) * p.TW;
    const int co0 = blockIdx.y * BN;
    const int PW = p.PW, PH = p.PH, CC = p.CC, PP = p.PP;
    const int lm = lane & 31, kq = lane >> 5;
    const int tile_px = p.TH * p.TW;
    const int G = C4 ? a.KH : a.KH * a.KW;
    const int WP = C4 ? 56 : PP;                   // filter row pitch (elements)
    const int patch_elems = (PH * PW * (C4 ? 4 : PP) + 15) & ~7;
    unsigned short* patch = smem;
    unsigned short* wl = smem + patch_elems;
    float* abl = reinterpret_cast<float*>(smem + patch_elems + G * BN * WP);  // [2][Cin] on-load affine

    int laneA[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        int t = (wave * WM + m) * 32 + lm;
        if (t >= tile_px) t = 0;
        const int py = t / p.TW, px = t - py * p.TW;
        laneA[m] = C4 ? (py * PW + px) * 4 + kq * 8 : (py * a.stride * PW + px * a.stride) * PP + kq * 8;
    }
    const int laneB = lm * WP + kq * 8;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int nn = 0; nn < WN; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;

    const int vy0 = ty0 * a.stride - a.pad_t, vx0 = tx0 * a.stride - a.pad_l;

    if constexpr (C4) {
        // ---- image layer: stage the fp32 RGB patch as 4-channel bf16 pixels, the whole packed filter, one sweep
        const float* xn = static_cast<const float*>(a.x) + (size_t)n * a.H * a.W * 3;
        for (int e = tid; e < PH * PW; e += 256) {
            const int py = e / PW, px = e - py * PW;
            int sy, sx;
            const bool ok = bsrc_coord(a.src_mode, a.refl, vy0 + py, a.H, sy) && bsrc_coord(a.src_mode, a.refl, vx0 + px, a.W, sx);
            uint2 v = make_uint2(0u, 0u);
            if (ok) {
                const float* s = xn + ((size_t)sy * a.W + sx) * 3;
                v.x = pack2(s[0], s[1]);
                v.y = pack2(s[2], 0.f);
            }
            *reinterpret_cast<uint2*>(patch + e * 4) = v;
        }
        const int cpad = a.p.cout_pad;
        for (int e = tid; e < G * BN * 6; e += 256) {  // 48 elements = 6 x 16 bytes per (kh, co)
            const int row = e / 6, g8 = e - row * 6;
            const int kh = row / BN, col = row - kh * BN;
            const uint4 v = *reinterpret_cast<const uint4*>(a.w + ((size_t)kh * cpad + co0 + col) * 48 + g8 * 8);
            *reinterpret_cast<uint4*>(wl + row * WP + g8 * 8) = v;
        }
        __syncthreads();
        for (int kh = 0; kh < G; ++kh)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                bf16x8 af[WM], bfr[WN];
#pragma unroll
                for (int m = 0; m < WM; ++m) {
                    const unsigned short* src = patch + laneA[m] + kh * PW * 4 + ks * 16;  // 8-byte aligned
                    uint4 u;
                    const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 4);
                    u.x = lo.x;
                    u.y = lo.y;
                    u.z = hi.x;
                    u.w = hi.y;
                    af[m] = __builtin_bit_cast(bf16x8, u);
                }
#pragma unroll
                for (int nn = 0; nn < WN; ++nn)
                    bfr[nn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wl + (kh * BN + nn * 32) * WP + laneB + ks * 16));
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int nn = 0; nn < WN; ++nn)
                        acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[nn], acc[m][nn], 0, 0, 0);
            }
        __syncthreads();
    } else {
        const unsigned short* xn = static_cast<const unsigned short*>(a.x) + (size_t)n * a.H * a.W * a.Cin;
        const bool has_ab = a.in_a != nullptr;
        if (has_ab)
            for (int c = tid; c < a.Cin; c += 256) {
                abl[c] = a.in_a[(size_t)n * a.in_nstride + c];
                abl[a.Cin + c] = a.in_b[(size_t)n * a.in_nstride + c];
            }
        constexpr int PMAX = 8, WMAX = 12;  // 16-byte granules per thread and chunk (the plan guarantees the bounds)
        const int g8n = CC >> 3;            // granules per pixel / per filter row
        const int g8sh = g8n == 2 ? 1 : 2;
        const int ne_p = PH * PW * g8n, ne_w = G * BN * g8n;
        const int cpad = a.p.cout_pad;
        // Branch-free staging, as in fs_conv.hip: global reads go through buffer resources, and a granule that is
        // zero padding or not this thread's gets the offset kOOB -- the hardware range check returns zeros for it.
        // Granules the thread does not own are written (as zeros) to the eight slack elements behind the patch.
        constexpr unsigned kOOB = 0x80000000u;
        const int slack = patch_elems - 8;
        unsigned gvo[PMAX];  // byte offset of the granule in the image (chunk 0), or kOOB
        int pdst[PMAX];      // LDS element offset
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int e = tid + i * 256;
            gvo[i] = kOOB;
            pdst[i] = slack;
            if (e < ne_p) {
                const int pix = e >> g8sh, g8 = e & (g8n - 1);
                const int py = pix / PW, px = pix - py * PW;
                int sy, sx;
                const bool ok = bsrc_coord(a.src_mode, a.refl, vy0 + py, a.H, sy) && bsrc_coord(a.src_mode, a.refl, vx0 + px, a.W, sx);
                if (ok) gvo[i] = (unsigned)((sy * a.W + sx) * a.Cin + g8 * 8) * 2u;
                pdst[i] = pix * PP + g8 * 8;
            }
        }
        unsigned wvo[WMAX];
        int wdst[WMAX];
#pragma unroll
        for (int i = 0; i < WMAX; ++i) {
            const int e = tid + i * 256;
            wvo[i] = kOOB;
            wdst[i] = slack - (int)(wl - patch);  // relative to the filter area
            if (e < ne_w) {
                const int row = e >> g8sh, g8 = e & (g8n - 1);
                const int g = row / BN, col = row - g * BN;
                wvo[i] = (unsigned)((g * cpad + co0 + col) * a.Cin + g8 * 8) * 2u;
                wdst[i] = row * WP + g8 * 8;
            }
        }
        const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * a.Cin) * 2u);
        const unsigned w_bytes = __builtin_amdgcn_readfirstlane((unsigned)(G * cpad * a.Cin) * 2u);
        auto uniform_ptr = [](const unsigned short* ptr) {
            const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return reinterpret_cast<unsigned short*>(((unsigned long long)hi << 32) | lo);
        };
        uint4 pv[PMAX], wv[WMAX];
        auto issue = [&](int c0) {
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(xn), 0, x_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.w), 0, w_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < PMAX; ++i)
                pv[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, gvo[i], c0 * 2, 0));
#pragma unroll
            for (int i = 0; i < WMAX; ++i)
                wv[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr, wvo[i], c0 * 2, 0));
        };
        auto commit_as = [&](auto AB, int c0) {
#pragma unroll
            for (int i = 0; i < PMAX; ++i) {
                uint4 v = pv[i];
                if (decltype(AB)::value) {
                    const int g8 = (tid + i * 256) & (g8n - 1);
                    const float* pa = abl + c0 + g8 * 8;
                    const float* pb = pa + a.Cin;
                    // padding arrives as 0 and must stay 0: clear the shift with a bit mask (fma(0, a, 0) = 0)
                    const unsigned okm = gvo[i] != kOOB ? 0xFFFFFFFFu : 0u;
                    const float4 sa0 = *reinterpret_cast<const float4*>(pa), sa1 = *reinterpret_cast<const float4*>(pa + 4);
                    const uint4 sb0 = *reinterpret_cast<const uint4*>(pb), sb1 = *reinterpret_cast<const uint4*>(pb + 4);
                    const float sc[8] = {sa0.x, sa0.y, sa0.z, sa0.w, sa1.x, sa1.y, sa1.z, sa1.w};
                    const unsigned sh[8] = {sb0.x, sb0.y, sb0.z, sb0.w, sb1.x, sb1.y, sb1.z, sb1.w};
                    unsigned* w32 = reinterpret_cast<unsigned*>(&v);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float lo = fmaf(bf2f((unsigned short)(w32[k] & 0xFFFFu)), sc[2 * k], __uint_as_float(sh[2 * k] & okm));
                        float hi = fmaf(bf2f((unsigned short)(w32[k] >> 16)), sc[2 * k + 1], __uint_as_float(sh[2 * k + 1] & okm));
                        if (a.in_relu) {
                            lo = fmaxf(lo, 0.f);
                            hi = fmaxf(hi, 0.f);
                        }
                        w32[k] = pack2(lo, hi);
                    }
                }
                *reinterpret_cast<uint4*>(patch + pdst[i]) = v;
            }
#pragma unroll
            for (int i = 0; i < WMAX; ++i) *reinterpret_cast<uint4*>(wl + wdst[i]) = wv[i];
        };
        auto commit = [&](int c0) {
            if (has_ab)
                commit_as(std::true_type{}, c0);
            else
                commit_as(std::false_type{}, c0);
        };
        const int nks = CC >> 4;
        issue(0);
        __syncthreads();  // abl visible
        for (int c0 = 0; c0 < a.Cin; c0 += CC) {
            commit(c0);
            __syncthreads();
            if (c0 + CC < a.Cin) issue(c0 + CC);
            sweep_bf16<WM, WN>(acc, patch, wl, laneA, laneB, a.KH, a.KW, a.dil_x, PW, PP, WP, BN, nks);
            __syncthreads();
        }
    }

    // ---- epilogue (same structure as fs_conv.hip): per-tile instance-norm partials from the fp32 accumulators, store
    const int th_valid = min(p.TH, a.Ho - ty0), tw_valid = min(p.TW, a.Wo - tx0);
    const float inv_tw = 1.0f / (float)p.TW;
    auto row_of = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * kq; };
    // statistics partials and the staged tile go to LDS together: one barrier for both
    float* wst = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(smem_f) + p.wst_off);
    if (a.stats) tile_stats_write<WM, WN>(a, acc, wst, th_valid, tw_valid);
    store_tile_write<WM, WN>(a, acc, smem);
    __syncthreads();
    if (a.stats) tile_stats_finish<WN>(a, wst, blockIdx.x, co0);
    store_tile_finish<WN>(a, smem, n, ty0, tx0, co0, th_valid, tw_valid);
}


// (2) resident kernel: single-chunk layers (the image layer, the Cin = 16 layers incl. the kw-folded output layer).
template <int WM, int WN, bool C4>
__global__ __launch_bounds__(256, 2) void conv_bf16_resident_kernel(ConvBArgs a) {
    constexpr int BN = WN * 32;
    HIP_DYNAMIC_SHARED(float, smem_f)
    unsigned short* smem = reinterpret_cast<unsigned short*>(smem_f);
    const ConvBPlan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = p.tiles_y * p.tiles_x;
    const int total_tiles = a.N * tiles;
    const int co0 = blockIdx.y * BN;
    const int PW = p.PW, PH = p.PH, CC = p.CC, PP = p.PP;
    const int lm = lane & 31, kq = lane >> 5;
    const int tile_px = p.TH * p.TW;
    const int G = C4 ? a.KH : a.KH * a.KW;
    const int WP = C4 ? 56 : PP;  // filter row pitch (elements)
    const int patch_elems = (PH * PW * (C4 ? 4 : PP) + 15) & ~7;
    unsigned short* patch = smem;
    unsigned short* wl = smem + patch_elems;
    float* abl = reinterpret_cast<float*>(smem + patch_elems + G * BN * WP);  // [2][Cin] on-load affine
    float* sred = abl + 2 * a.Cin;                                             // [4][BN][4] statistics scratch
    float* stage = sred + 16 * BN + 3;                                          // tile staging for the coalesced store
    stage = reinterpret_cast<float*>(reinterpret_cast<size_t>(stage) & ~(size_t)15);  // 16-byte aligned
    const int cpad = p.cout_pad;
    // The whole filter stays resident in LDS and the workgroup walks a strided list of tiles (persistent
    // workgroups): the filter is staged once per workgroup instead of once per tile, and the next tile's patch is
    // in flight (registers) while the current one is multiplied.

    int laneA[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        int t = (wave * WM + m) * 32 + lm;
        if (t >= tile_px) t = 0;
        const int py = t / p.TW, px = t - py * p.TW;
        laneA[m] = C4 ? (py * PW + px) * 4 + kq * 8 : (py * a.stride * PW + px * a.stride) * PP + kq * 8;
    }
    const int laneB = lm * WP + kq * 8;
    const float inv_tw = 1.0f / (float)p.TW;
    auto row_of = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * kq; };

    // ---- staging machinery ----
    constexpr int PMAX = C4 ? 3 : 5;             // patch granules per thread and tile (the plan guarantees the bound)
    constexpr int DEPTH = 3;                     // tiles in flight: a tile's MFMA work is far shorter than the HBM latency
    const int g8n = C4 ? 1 : CC >> 3;            // 16-byte granules per pixel / per filter row
    const int g8sh = g8n == 1 ? 0 : (g8n == 2 ? 1 : 2);
    const int ne_p = PH * PW * g8n;
    const int ne_w = C4 ? G * BN * 6 : G * BN * g8n;
    uint4 pvr[DEPTH][PMAX];
    unsigned vmaskr[DEPTH];  // bit i: pvr[.][i] holds real pixels (not padding)
    const float inv_tiles = 1.0f / (float)tiles, inv_tx = 1.0f / (float)p.tiles_x;
    int pyx[PMAX];  // tile-independent patch coordinates of this thread's granules: py << 16 | px (-1: none)
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        const int e = tid + i * 256;
        pyx[i] = -1;
        if (e < ne_p) {
            const int pix = e >> g8sh;
            const int py = pix / PW;
            pyx[i] = (py << 16) | (pix - py * PW);
        }
    }
    auto issue_patch = [&](auto SET, int tile) {
        uint4(&pv)[PMAX] = pvr[decltype(SET)::value];
        unsigned& vmask = vmaskr[decltype(SET)::value];
        const int c0 = 0;
        const int n = fdiv(tile, inv_tiles), tr = tile - n * tiles;
        const int tyi = fdiv(tr, inv_tx), txi = tr - tyi * p.tiles_x;
        const int vy0 = tyi * p.TH * a.stride - a.pad_t, vx0 = txi * p.TW * a.stride - a.pad_l;
        vmask = 0;
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int e = tid + i * 256;
            pv[i] = make_uint4(0u, 0u, 0u, 0u);
            if (pyx[i] >= 0) {
                const int g8 = e & (g8n - 1);
                const int py = pyx[i] >> 16, px = pyx[i] & 0xFFFF;
                int sy, sx;
                if (bsrc_coord(a.src_mode, a.refl, vy0 + py, a.H, sy) && bsrc_coord(a.src_mode, a.refl, vx0 + px, a.W, sx)) {
                    vmask |= 1u << i;
                    if constexpr (C4) {
                        const float* s = static_cast<const float*>(a.x) + (((size_t)n * a.H + sy) * a.W + sx) * 3;
                        pv[i].x = __builtin_bit_cast(unsigned, s[0]);
                        pv[i].y = __builtin_bit_cast(unsigned, s[1]);
                        pv[i].z = __builtin_bit_cast(unsigned, s[2]);
                    } else {
                        pv[i] = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(a.x) +
                                                                (((size_t)n * a.H + sy) * a.W + sx) * a.Cin + c0 + g8 * 8);
                    }
                }
            }
        }
    };
    const bool has_ab = !C4 && a.in_a != nullptr;
    auto commit_patch = [&](auto SET) {
        const uint4(&pv)[PMAX] = pvr[decltype(SET)::value];
        const unsigned vmask = vmaskr[decltype(SET)::value];
        const int c0 = 0;
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int e = tid + i * 256;
            if (e >= ne_p) continue;
            if constexpr (C4) {
                uint2 v = make_uint2(0u, 0u);
                if (vmask & (1u << i)) {
                    v.x = pack2(__builtin_bit_cast(float, pv[i].x), __builtin_bit_cast(float, pv[i].y));
                    v.y = pack2(__builtin_bit_cast(float, pv[i].z), 0.f);
                }
                *reinterpret_cast<uint2*>(patch + e * 4) = v;
            } else {
                const int pix = e >> g8sh, g8 = e & (g8n - 1);
                uint4 v = pv[i];
                if ((vmask & (1u << i)) && has_ab) {
                    const float* pa = abl + c0 + g8 * 8;
                    const float* pb = pa + a.Cin;
                    unsigned* w32 = reinterpret_cast<unsigned*>(&v);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float lo = fmaf(bf2f((unsigned short)(w32[k] & 0xFFFFu)), pa[2 * k], pb[2 * k]);
                        float hi = fmaf(bf2f((unsigned short)(w32[k] >> 16)), pa[2 * k + 1], pb[2 * k + 1]);
                        if (a.in_relu) {
                            lo = fmaxf(lo, 0.f);
                            hi = fmaxf(hi, 0.f);
                        }
                        w32[k] = pack2(lo, hi);
                    }
                }
                *reinterpret_cast<uint4*>(patch + pix * PP + g8 * 8) = v;
            }
        }
    };
    f32x16 acc[WM][WN];
    auto sweep = [&]() {
        if constexpr (C4) {
            for (int kh = 0; kh < G; ++kh)
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    bf16x8 af[WM], bfr[WN];
#pragma unroll
                    for (int m = 0; m < WM; ++m) {
                        const unsigned short* src = patch + laneA[m] + kh * PW * 4 + ks * 16;  // 8-byte aligned
                        const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 4);
                        af[m] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                    }
#pragma unroll
                    for (int nn = 0; nn < WN; ++nn)
                        bfr[nn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wl + (kh * BN + nn * 32) * WP + laneB + ks * 16));
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int nn = 0; nn < WN; ++nn)
                            acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[nn], acc[m][nn], 0, 0, 0);
                }
        } else {
            sweep_bf16<WM, WN>(acc, patch, wl, laneA, laneB, a.KH, a.KW, a.dil_x, PW, PP, WP, BN, CC >> 4);
        }
    };

    // per-image on-load affine table; persistent workgroups reload it when they cross an image boundary
    int abl_n = -1;
    auto load_abl = [&](int n) {
        if (has_ab && n != abl_n) {
            for (int c = tid; c < a.Cin; c += 256) {
                abl[c] = a.in_a[(size_t)n * a.in_nstride + c];
                abl[a.Cin + c] = a.in_b[(size_t)n * a.in_nstride + c];
            }
        }
        abl_n = n;
    };

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    for (int e = tid; e < ne_w; e += 256) {  // the whole filter, once
        if constexpr (C4) {  // 48 elements = 6 granules per (kh, co)
            const int row = e / 6, g8 = e - row * 6;
            const int kh = row / BN, col = row - kh * BN;
            *reinterpret_cast<uint4*>(wl + row * WP + g8 * 8) =
                *reinterpret_cast<const uint4*>(a.w + ((size_t)kh * cpad + co0 + col) * 48 + g8 * 8);
        } else {
            const int row = e >> g8sh, g8 = e & (g8n - 1);
            const int g = row / BN, col = row - g * BN;
            *reinterpret_cast<uint4*>(wl + row * WP + g8 * 8) =
                *reinterpret_cast<const uint4*>(a.w + ((size_t)g * cpad + co0 + col) * a.Cin + g8 * 8);
        }
    }
    const int gstep = (int)gridDim.x;
    issue_patch(std::integral_constant<int, 0>{}, tile);
    if (tile + gstep < total_tiles) issue_patch(std::integral_constant<int, 1>{}, tile + gstep);
    if (tile + 2 * gstep < total_tiles) issue_patch(std::integral_constant<int, 2>{}, tile + 2 * gstep);
    auto body = [&](auto SET) -> bool {  // one tile; its patch sits in register set SET
        const int n = fdiv(tile, inv_tiles), tr = tile - n * tiles;
        const int tyi = fdiv(tr, inv_tx);
        const int ty0 = tyi * p.TH, tx0 = (tr - tyi * p.tiles_x) * p.TW;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int nn = 0; nn < WN; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;
        if (has_ab && n != abl_n) {
            __syncthreads();  // nobody may still be reading the previous image's table
            load_abl(n);
            __syncthreads();
        }
        commit_patch(SET);
        __syncthreads();
        if (tile + DEPTH * gstep < total_tiles) issue_patch(SET, tile + DEPTH * gstep);
        sweep();
        __syncthreads();
        // ---- epilogue: per-tile instance-norm partials from the fp32 accumulators, then the coalesced store
        const int th_valid = min(p.TH, a.Ho - ty0), tw_valid = min(p.TW, a.Wo - tx0);
        if (a.stats) tile_stats_write<WM, WN>(a, acc, sred, th_valid, tw_valid);
        store_tile_write<WM, WN>(a, acc, stage);
        __syncthreads();  // one barrier for the statistics partials and the staged tile
        if (a.stats) tile_stats_finish<WN>(a, sred, tile, co0);
        store_tile_finish<WN>(a, stage, n, ty0, tx0, co0, th_valid, tw_valid);
        tile += gstep;
        return tile < total_tiles;
    };
    for (;;) {
        if (!body(std::integral_constant<int, 0>{})) break;
        if (!body(std::integral_constant<int, 1>{})) break;
        if (!body(std::integral_constant<int, 2>{})) break;
    }
}

// ---------------------------------------------------------------------------------------------- plan / launch
static inline int roundup(int v, int m) { return (v + m - 1) / m * m; }

ConvBPlan conv_bf16_plan(const ConvBArgs& a) {
    ConvBPlan p{};
    if (bstream_instance(a)) {   // every layer behind the image layer: persistent streaming kernel, filter in registers
        bstream_plan(a, &p);
        return p;
    }
    p.c4 = a.Cin == 3;
    p.BN = a.Cout > 32 ? 64 : 32;
    p.cout_pad = roundup(a.Cout, p.BN);
    const int dil = a.dil_x > 0 ? a.dil_x : 1;
    const int kw_span = p.c4 ? 12 : (a.KW - 1) * dil + 1;
    const int G = p.c4 ? a.KH : a.KH * a.KW;
    // widest pixel tile / deepest channel chunk whose staging fits the per-thread register budget; the narrower
    // tile when the launch could not fill the chip
    const int wm_env = tune_int("FS_BF16_WM", 0);  // tuning aid: cap the pixel tile (1: 128 pixels)
    // the resident kernel with 64 output channels and the 256-pixel tile does not fit the register file (it spills
    // ~1 KB per lane and measures 1.5x slower than the 128-pixel tile): start it at WM = 1
    const bool single_chunk = p.c4 || a.Cin <= 32;
    const int wm0 = wm_env > 0 ? wm_env : (single_chunk && p.BN == 64 ? 1 : 2);
    for (p.WM = wm0; p.WM >= 1; --p.WM) {
        const int max_px = 4 * p.WM * 32;
        plan_tile(a.Ho, a.Wo, a.KH, p.c4 ? 12 : a.KW, a.stride, max_px, &p.TH, &p.TW);
        p.tiles_y = cdiv(a.Ho, p.TH);
        p.tiles_x = cdiv(a.Wo, p.TW);
        p.PH = (p.TH - 1) * a.stride + a.KH;
        p.PW = (p.TW - 1) * a.stride + kw_span;
        const long wgs = (long)a.N * p.tiles_y * p.tiles_x * (p.cout_pad / p.BN);
        bool fits = false;
        for (p.CC = p.c4 ? 4 : (a.Cin % 32 == 0 ? 32 : 16); p.CC >= (p.c4 ? 4 : 16); p.CC >>= 1) {
            p.PP = p.CC + 8;
            const int patch_elems = (p.PH * p.PW * (p.c4 ? 4 : p.PP) + 15) & ~7;
            const bool resident = p.c4 || a.Cin == p.CC;
            const int stage_bytes = max_px * p.BN * (a.y_f32 ? 4 : 2);
            const int main_bytes = 2 * (patch_elems + G * p.BN * (p.c4 ? 56 : p.PP)) + 8 * a.Cin;
            if (resident) {  // dedicated staging: the resident filter must survive the epilogue
                p.wst_off = main_bytes;
                p.lds_bytes = main_bytes + 4 * 16 * p.BN + 32 + stage_bytes;
            } else {         // tile staged over the (dead) patch + filter area, statistics partials behind it
                p.wst_off = (main_bytes > stage_bytes ? main_bytes : stage_bytes);
                p.lds_bytes = p.wst_off + 4 * 16 * p.BN + 32;
            }
            // staging registers: chunked kernel 8 patch + 12 filter granules per thread, resident kernel 5 patch
            fits = p.lds_bytes <= 80 * 1024 &&
                   (p.c4 || (resident ? p.PH * p.PW * (p.CC / 8) <= 5 * 256
                                      : (p.PH * p.PW * (p.CC / 8) <= 8 * 256 && G * p.BN * (p.CC / 8) <= 12 * 256)));
            if (fits || p.c4) break;
        }
        if (p.CC < (p.c4 ? 4 : 16)) p.CC = p.c4 ? 4 : 16;
        if (p.WM == 1 || (fits && wgs >= 512)) break;
    }
    return p;
}

int conv_bf16_launch(const ConvBArgs& a_in, hipStream_t s) {
    if (a_in.p.bs) return bstream_launch(a_in, s);
    ConvBArgs a = a_in;
    if (a.dil_x < 1) a.dil_x = 1;
    const ConvBPlan& p = a.p;
    if (!p.c4 && (a.Cin % p.CC)) return -1;
    if (a.Cout % 8 || (a.shuffle && (a.Cout >> 2) % 8)) return -1;  // 16-byte output granules
    if (p.c4 && (a.stride != 1 || a.KW > 12 || !a.x_f32)) return -1;
    if (p.lds_bytes > 160 * 1024) return -2;
    const int G = p.c4 ? a.KH : a.KH * a.KW;
    const bool resident_k = p.c4 || a.Cin == p.CC;
    if (!p.c4 && resident_k && p.PH * p.PW * (p.CC / 8) > 5 * 256) return -2;
    if (!resident_k && (p.PH * p.PW * (p.CC / 8) > 8 * 256 || G * p.BN * (p.CC / 8) > 12 * 256)) return -2;
    // resident kernels walk a strided list of tiles: cap the grid at a few waves of workgroups per CU so that the
    // resident filter is amortised over many tiles
    const long total_tiles = (long)a.N * p.tiles_y * p.tiles_x;
    const int cap = tune_int("FS_BF16_GRID", 2048);  // (tests shrink it to force the multi-tile walk on small images)
    const bool resident = p.c4 || a.Cin == p.CC;
    dim3 grid((unsigned)(resident && total_tiles > cap ? cap : total_tiles), (unsigned)(p.cout_pad / p.BN));
#define FS_BLAUNCH_K(KERNEL_)                                                                                             \
    do {                                                                                                                  \
        static BigLds lds_attr;                                                                                           \
        lds_attr.ensure(reinterpret_cast<const void*>(KERNEL_));                                                          \
        hipLaunchKernelGGL(KERNEL_, grid, dim3(256), (size_t)p.lds_bytes, s, a);                                          \
    } while (0)
#define FS_BLAUNCH(WM_, WN_, C4_)                                        \
    do {                                                                 \
        if (resident)                                                    \
            FS_BLAUNCH_K((conv_bf16_resident_kernel<WM_, WN_, C4_>));    \
        else                                                             \
            FS_BLAUNCH_K((conv_bf16_chunked_kernel<WM_, WN_, false>));   \
    } while (0)
    if (p.c4) {
        if (p.BN != 32) return -4;
        if (p.WM == 2)
            FS_BLAUNCH(2, 1, true);
        else
            FS_BLAUNCH(1, 1, true);
    } else if (p.BN == 64) {
        if (p.WM == 2)
            FS_BLAUNCH(2, 2, false);
        else
            FS_BLAUNCH(1, 2, false);
    } else {
        if (p.WM == 2)
            FS_BLAUNCH(2, 1, false);
        else
            FS_BLAUNCH(1, 1, false);
    }
#undef FS_BLAUNCH
#undef FS_BLAUNCH_K
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---------------------------------------------------------------------------------------------- weight packing
// One launch converts every filter of the net to the packed bf16 layouts the kernel reads:
//   PK_CONV  [G][cout_pad][Cin]        from HWIO fp32
//   PK_UP    same, G = 4, from the phase-collapsed resize-conv filter (im_transf_net.py:122-155; see fs_elem.hip)
//   PK_FOLD  [18][32][16]              kw-folded output layer (fs_fold.hip): tap = kh*2+b, column j = v*3+co
//   PK_C4    [9][32][48]               image layer: k = kw*4 + ci, zero for kw >= 9, ci == 3, co >= Cout
struct PackJob {
    int kind, G, Cin, Cout, cout_pad, total;
    const float* src;
    unsigned short* dst;
};
struct PackBatch {
    int n;
    PackJob j[16];
};
__device__ __forceinline__ bool bup_in_R(int a, int d, int k) {
    if (a == 0) return d == 0;
    return d == 0 ? k < 2 : k == 2;
}
__global__ __launch_bounds__(256) void pack_bf16_kernel(PackBatch b) {
    const PackJob& q = b.j[blockIdx.y];
    const float* __restrict__ w = q.src;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < q.total; i += gridDim.x * 256) {
        float v = 0.f;
        if (q.kind == PK_CONV) {
            const int ci = i % q.Cin;
            const int r = i / q.Cin;
            const int co = r % q.cout_pad, g = r / q.cout_pad;
            if (co < q.Cout) v = w[((size_t)g * q.Cin + ci) * q.Cout + co];
        } else if (q.kind == PK_UP) {  // q.Cout = 4 * Co virtual channels
            const int Co = q.Cout >> 2;
            const int ci = i % q.Cin;
            const int r = i / q.Cin;
            const int j = r % q.cout_pad, tap = r / q.cout_pad;
            if (j < q.Cout) {
                const int dy = tap >> 1, dx = tap & 1;
                const int qq = j / Co, co = j - qq * Co, aa = qq >> 1, bb = qq & 1;
                for (int kh = 0; kh < 3; ++kh)
                    for (int kw = 0; kw < 3; ++kw)
                        if (bup_in_R(aa, dy, kh) && bup_in_R(bb, dx, kw)) v += w[((kh * 3 + kw) * q.Cin + ci) * Co + co];
            }
        } else if (q.kind == PK_FOLD) {
            const int ci = i % q.Cin;
            const int r = i / q.Cin;
            const int j = r % q.cout_pad, tap = r / q.cout_pad;
            const int kh = tap >> 1, bb = tap & 1;
            const int vv = j / 3, co = j - vv * 3, kw = 5 * bb + vv;
            if (j < 15 && kw < 9) v = w[((kh * 9 + kw) * q.Cin + ci) * 3 + co];
        } else {  // PK_C4
            const int k = i % 48;
            const int r = i / 48;
            const int co = r % q.cout_pad, kh = r / q.cout_pad;
            const int kw = k >> 2, ci = k & 3;
            if (kw < 9 && ci < 3 && co < q.Cout) v = w[((kh * 9 + kw) * 3 + ci) * q.Cout + co];
        }
        q.dst[i] = f2bf(v);
    }
}

// ---------------------------------------------------------------------------------------------- elementwise
// h = a*z + b + T(skip[y+2, x+2]) on bf16 tensors (im_transf_net.py:268-274); 8 channels (16 bytes) per thread
__global__ __launch_bounds__(256) void apply_res_bf16_kernel(const unsigned short* __restrict__ z, const float* __restrict__ a,
                                                             const float* __restrict__ b, const unsigned short* __restrict__ skip,
                                                             const float* __restrict__ sa, const float* __restrict__ sb,
                                                             int skip_relu, unsigned short* __restrict__ out, int H, int W) {
    // grid (row segments, H, N), C = 64: a thread owns 8 channels (16 bytes) of one pixel; no divisions on the path
    constexpr int C = 64, c8n = 8;
    const int n = blockIdx.z, y = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;  // (x, c8) within the row
    if (j >= W * c8n) return;
    const int x = j >> 3, c8 = j & 7;
    const size_t i = ((size_t)n * H + y) * W * c8n + j;
    const uint4 zv = *reinterpret_cast<const uint4*>(z + i * 8);
    const uint4 sv = *reinterpret_cast<const uint4*>(skip + ((((size_t)n * (H + 4) + y + 2) * (W + 4) + x + 2) * C + c8 * 8));
    const unsigned* z32 = reinterpret_cast<const unsigned*>(&zv);
    const unsigned* s32 = reinterpret_cast<const unsigned*>(&sv);
    const int k0 = n * C + c8 * 8;
    const float4 a0 = *reinterpret_cast<const float4*>(a + k0), a1 = *reinterpret_cast<const float4*>(a + k0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(b + k0), b1 = *reinterpret_cast<const float4*>(b + k0 + 4);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    // the skip's own on-load affine (block 0: skip = IN + ReLU of the third conv's raw output), as two 16-byte loads per vector like a / b -- the sixteen
    // 4-byte loads per thread it replaces made the first apply of a 1080p batch-8 forward 172 us against 72 us for the other four
    float sav[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, sbv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (sa) {
        const float4 s0 = *reinterpret_cast<const float4*>(sa + k0), s1 = *reinterpret_cast<const float4*>(sa + k0 + 4);
        const float4 t0 = *reinterpret_cast<const float4*>(sb + k0), t1 = *reinterpret_cast<const float4*>(sb + k0 + 4);
        sav[0] = s0.x, sav[1] = s0.y, sav[2] = s0.z, sav[3] = s0.w, sav[4] = s1.x, sav[5] = s1.y, sav[6] = s1.z, sav[7] = s1.w;
        sbv[0] = t0.x, sbv[1] = t0.y, sbv[2] = t0.z, sbv[3] = t0.w, sbv[4] = t1.x, sbv[5] = t1.y, sbv[6] = t1.z, sbv[7] = t1.w;
    }
    uint4 ov;
    unsigned* o32 = reinterpret_cast<unsigned*>(&ov);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float sk = bf2f((unsigned short)(h ? s32[k] >> 16 : s32[k] & 0xFFFFu));
            if (sa) sk = fmaf(sk, sav[2 * k + h], sbv[2 * k + h]);
            if (skip_relu) sk = fmaxf(sk, 0.f);
            const float zz = bf2f((unsigned short)(h ? z32[k] >> 16 : z32[k] & 0xFFFFu));
            r[h] = fmaf(zz, av[2 * k + h], bv[2 * k + h]) + sk;
        }
        o32[k] = pack2(r[0], r[1]);
    }
    *reinterpret_cast<uint4*>(out + i * 8) = ov;
}

// ---------------------------------------------------------------------------------------------- the forward
static size_t take(size_t& off, size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
}

static ConvBArgs unit_bargs(const Unit& u, int N) {
    ConvBArgs a{};
    a.N = N;
    a.H = u.Hsrc;
    a.W = u.Wsrc;
    a.Cin = u.Cin;
    a.Ho = u.Hc;
    a.Wo = u.Wc;
    a.Cout = u.Cc;
    a.KH = u.K;
    a.KW = u.KWx;
    a.dil_x = u.dil_x;
    a.stride = u.stride;
    a.pad_t = u.pad_t;
    a.pad_l = u.pad_l;
    a.src_mode = u.src_mode;
    a.refl = u.refl;
    a.shuffle = u.kind == 1;
    a.x_f32 = u.Cin == 3;
    a.y_f32 = 0;  // (the 16 virtual channels of the kw-folded output layer are bf16 too: 51.8 dB either way)
    return a;
}

int tnet_layout_bf16(int N, int H, int W, BTnetLayout* L) {
    memset(L, 0, sizeof(*L));
    tnet_layout(N, H, W, 0, &L->geo);
    size_t off = 0;
    for (int i = 0; i < 16; ++i) {
        const Unit& u = L->geo.u[i];
        ConvBArgs a = unit_bargs(u, N);
        L->plan[i] = conv_bf16_plan(a);
        const ConvBPlan& p = L->plan[i];
        L->tiles[i] = u.kind == 2 ? cdiv(u.Hout * u.Wout, 256) : p.tiles_y * p.tiles_x;
        const size_t act = (size_t)N * u.Hout * u.Wout * u.Cout;
        L->z[i] = take(off, act * (i == 15 ? 4 : 2));
        L->stats[i] = take(off, (size_t)N * (L->tiles[i] + kFinalizeSplit) * (u.kind == 2 ? u.Cout : u.Cc) * 3 * 4);
        L->mean[i] = take(off, (size_t)N * u.Cout * 4);
        L->rstd[i] = take(off, (size_t)N * u.Cout * 4);
        L->a[i] = take(off, (size_t)N * u.Cout * 4);
        L->b[i] = take(off, (size_t)N * u.Cout * 4);
        const int G = p.c4 ? u.K : u.K * u.KWx;
        L->wpk_elems[i] = (size_t)G * p.cout_pad * (p.c4 ? 48 : u.Cin);
        L->wpk[i] = take(off, L->wpk_elems[i] * 2);
    }
    for (int k = 0; k < 5; ++k) {
        const Unit& u2 = L->geo.u[3 + 2 * k + 1];
        L->h[k] = take(off, (size_t)N * u2.Hout * u2.Wout * 64 * 2);
    }
    L->zfold = take(off, (size_t)N * L->geo.u[15].Hc * L->geo.u[15].Wc * 16 * 2);
    L->total_bytes = off;
    return 0;
}

int tnet_forward_bf16(const BTnetLayout& L, const float* params, const float* x, float* y, void* ws_v, hipStream_t s) {
    unsigned char* ws = static_cast<unsigned char*>(ws_v);
    const int N = L.geo.N;
    PackBatch pb{};
    int mx = 0;
    for (int i = 0; i < 16; ++i) {
        const Unit& u = L.geo.u[i];
        PackJob& q = pb.j[pb.n++];
        q.kind = i == 0 ? PK_C4 : (u.kind == 1 ? PK_UP : (u.kind == 2 ? PK_FOLD : PK_CONV));
        q.G = L.plan[i].c4 ? u.K : u.K * u.KWx;
        q.Cin = u.Cin;
        q.Cout = u.kind == 2 ? 16 : (i == 0 ? u.Cout : u.Cc);
        q.cout_pad = L.plan[i].cout_pad;
        q.total = (int)L.wpk_elems[i];
        q.src = params + u.w_off;
        q.dst = reinterpret_cast<unsigned short*>(ws + L.wpk[i]);
        if (q.total > mx) mx = q.total;
    }
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(cdiv(mx, 256), pb.n), dim3(256), 0, s, pb);

    const void* src = x;
    const float* src_a = nullptr;
    const float* src_b = nullptr;
    for (int i = 0; i < 16; ++i) {
        const Unit& u = L.geo.u[i];
        ConvBArgs a = unit_bargs(u, N);
        a.p = L.plan[i];
        a.x = src;
        a.in_a = src_a;
        a.in_b = src_b;
        a.in_nstride = src_a ? u.Cin : 0;
        a.in_relu = src_a ? 1 : 0;
        a.w = reinterpret_cast<const unsigned short*>(ws + L.wpk[i]);
        a.y = u.kind == 2 ? static_cast<void*>(ws + L.zfold) : static_cast<void*>(ws + L.z[i]);
        a.stats = u.kind == 2 ? nullptr : reinterpret_cast<float*>(ws + L.stats[i]);
        int rc = conv_bf16_launch(a, s);
        if (rc) return rc;
        float* stats = reinterpret_cast<float*>(ws + L.stats[i]);
        if (u.kind == 2) {
            rc = fold5_fwd_bf16(reinterpret_cast<const unsigned short*>(ws + L.zfold), reinterpret_cast<float*>(ws + L.z[i]), stats,
                                N, u.Hout, u.Wout, s);
            if (rc) return rc;
        }
        float* ua = reinterpret_cast<float*>(ws + L.a[i]);
        float* ub = reinterpret_cast<float*>(ws + L.b[i]);
        rc = in_finalize(stats, N, L.tiles[i], u.Cout, u.kind == 1 ? 4 : 1, params + u.g_off, params + u.b_off, 1e-3f,
                         reinterpret_cast<float*>(ws + L.mean[i]), reinterpret_cast<float*>(ws + L.rstd[i]), ua, ub, s,
                         stats + (size_t)N * L.tiles[i] * (u.kind == 2 ? u.Cout : u.Cc) * 3);
        if (rc) return rc;
        src = ws + L.z[i];
        src_a = ua;
        src_b = ub;
        if (i >= 4 && i <= 12 && ((i - 3) & 1)) {  // second conv of a residual block -> materialise h_k (bf16)
            const int k = (i - 4) / 2;
            const unsigned short* skip;
            const float *sa = nullptr, *sb = nullptr;
            if (k == 0) {
                skip = reinterpret_cast<const unsigned short*>(ws + L.z[2]);
                sa = reinterpret_cast<const float*>(ws + L.a[2]);
                sb = reinterpret_cast<const float*>(ws + L.b[2]);
            } else {
                skip = reinterpret_cast<const unsigned short*>(ws + L.h[k - 1]);
            }
            hipLaunchKernelGGL(apply_res_bf16_kernel, dim3(cdiv(u.Wout * 8, 256), u.Hout, N), dim3(256), 0, s,
                               reinterpret_cast<const unsigned short*>(ws + L.z[i]), ua, ub, skip, sa, sb, k == 0 ? 1 : 0,
                               reinterpret_cast<unsigned short*>(ws + L.h[k]), u.Hout, u.Wout);
            src = ws + L.h[k];
            src_a = src_b = nullptr;
        }
    }
    const Unit& u = L.geo.u[15];
    return apply_tanh(reinterpret_cast<const float*>(ws + L.z[15]), reinterpret_cast<const float*>(ws + L.a[15]),
                      reinterpret_cast<const float*>(ws + L.b[15]), y, N, u.Hout * u.Wout, 3, s);
}

}  // namespace fs
