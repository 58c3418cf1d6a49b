// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32: exact fp32, bit-identical to an fmaf chain).
//
// One kernel family serves every dense contraction with a sliding window on the path:
//   * transform-net convs (reference im_transf_net.py:91-119): 9x9/3x3, stride 1/2, SAME /
//     VALID / REFLECT-40 fused, instance-norm+ReLU of the producer applied on load
//     (im_transf_net.py:218-247), per-tile instance-norm statistics in the epilogue;
//   * the phase-collapsed resize-conv (im_transf_net.py:122-155) as a 2x2-tap conv with a
//     pixel-shuffle store;
//   * VGG16 3x3 convs + bias + ReLU (libs/vgg16.py:45-173);
//   * every dgrad (the same kernel on flipped/transposed filters; stride-2 dgrad through the
//     zero-dilated virtual input) and the Gram backward dF = F*S as a 1x1 conv with
//     per-sample filters (utils.py:78-81 adjoint).
//
// Mapping (wave64, 4 waves per workgroup): the GEMM M dimension is a TH x TW tile of output
// pixels flattened row-major (256 pixels per workgroup, 32 or 16 consecutive pixels per MFMA
// tile), N is a block of 64/32/16 output channels, K runs over (tap, input channel).  The
// input patch of the tile (with halo) is staged ONCE per channel chunk into LDS as
// [pixel][CC+1] (odd pitch: the 32 lanes of an A-fragment read hit 32 distinct banks) and
// re-used by all KHxKW taps; filters are staged as [k][BN] (B-fragment reads are contiguous).
#include "fs_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace fs {

__device__ __forceinline__ bool src_coord(int mode, int refl, int v, int n_src, int& s) {
    if (mode == SRC_PLAIN) {
        s = v;
        return v >= 0 && v < n_src;
    } else if (mode == SRC_REFLECT) {
        if (v < 0 || v >= n_src + 2 * refl) return false;
        s = v - refl;
        if (s < 0) s = -s;
        if (s >= n_src) s = 2 * (n_src - 1) - s;
        return true;
    } else if (mode == SRC_DILATE2) {
        if (v < 0 || (v & 1)) return false;
        s = v >> 1;
        return s < n_src;
    } else {  // SRC_UP4
        if (v < 0) return false;
        s = v >> 2;
        return s < n_src;
    }
}

template <int MT>
struct Frag;
template <>
struct Frag<32> {
    typedef f32x16 acc_t;
    static constexpr int NACC = 16, KSTEP = 2;
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // accumulator register r of lane l holds row(r,l) of the 32x32 tile, column l&31
    __device__ static __forceinline__ int row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};
template <>
struct Frag<16> {
    typedef f32x4 acc_t;
    static constexpr int NACC = 4, KSTEP = 4;
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ int row(int r, int lane) { return 4 * (lane >> 4) + r; }
};

#ifdef FS_CONV_TRACE
// debug build only (tools/conv_trace.py): per-workgroup phase cycle counts of the last launch
__device__ long long g_conv_trace[4096 * 8];
#define FS_TRACE_NOW() ((long long)__builtin_readcyclecounter())
#endif

template <int MT, int WM, int WN, bool FLAT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    typedef Frag<MT> F;
    typedef typename F::acc_t acc_t;
    constexpr int KSTEP = F::KSTEP, NACC = F::NACC;
    constexpr int BN = WN * MT;
    HIP_DYNAMIC_SHARED(float, smem)
    const ConvPlan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef FS_CONV_TRACE
    const long long tr_t0 = FS_TRACE_NOW();
    long long tr_pro = tr_t0, tr_sweep = 0, tr_commit = 0, tr_bar = 0;
#endif
    if (p.skew > 0) {
        // Two workgroups share a CU (one wave of each per SIMD).  Launched together and doing identical work they
        // run in lockstep: both stage, both multiply, both store at the same time, and the MFMA pipe idles whenever
        // they are not both in a sweep.  Delaying the workgroup in the odd wave slot by about half a chunk period
        // puts its staging / epilogue phases under the other one's sweeps.  Only the first round needs it: later
        // workgroups inherit the stagger from the slot they replace.
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (lin < 512u) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID; bits 3:0 = wave slot on the SIMD
            if (hw & 1u)
                for (int i = 0; i < p.skew; ++i) __builtin_amdgcn_s_sleep(32);
        }
    }
    const int tiles = p.tiles_y * p.tiles_x;
    // XCD-aware workgroup -> tile map.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in
    // linear dispatch order; give every XCD a CONTIGUOUS range of (tile, channel-block) work, channel blocks of a
    // tile adjacent: the 2..8 workgroups that read the same input patch then run on the same XCD at about the
    // same time and share it through that L2, and neighbouring tiles (shared halos) stay on one XCD too.
    int tile_id = blockIdx.x, cob = blockIdx.y;
    if (p.xcd_swizzle) {
        const unsigned total = gridDim.x * gridDim.y;
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned xcd = lin & 7u, slot = lin >> 3;
        const unsigned q = total >> 3, r = total & 7u;
        const unsigned j = xcd * q + (xcd < r ? xcd : r) + slot;
        tile_id = __builtin_amdgcn_readfirstlane((int)(j / gridDim.y));
        cob = (int)(j - (unsigned)tile_id * gridDim.y);
    }
    // x / d through a float reciprocal: exact for these magnitudes (x < 2^22), ~5 instructions instead of ~35
    auto fdiv = [](int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); };
    // (the divisions run on the vector ALU; readfirstlane tells the compiler the results are wave-uniform, so that
    // everything derived from them -- base pointers, buffer resources, loop bounds -- lives in scalar registers)
    const int n = __builtin_amdgcn_readfirstlane(fdiv(tile_id, 1.0f / (float)tiles));
    const int tr = tile_id - n * tiles;
    const int tyi = __builtin_amdgcn_readfirstlane(fdiv(tr, 1.0f / (float)p.tiles_x));
    const int ty0 = tyi * p.TH, tx0 = (tr - tyi * p.tiles_x) * p.TW;
    const int co0 = cob * BN;
    const int S = p.S, PW = p.PW, PH = p.PH, LG = p.LG, CC = p.CC;
    const float inv_pw = 1.0f / (float)PW;
    const int patch_floats = (PH * PW * S + 8 + 3) & ~3;
    float* patch = smem;
    float* wl = smem + patch_floats;

    const int lm = lane & (MT - 1), kq = lane / MT;
    const int tile_px = p.TH * p.TW;
    int laneA[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        int t = (wave * WM + m) * MT + lm;
        if (t >= tile_px) t = 0;
        const int py = fdiv(t, 1.0f / (float)p.TW), px = t - py * p.TW;
        laneA[m] = (py * a.stride * PW + px * a.stride) * S + kq;
    }
    const int laneB = kq * BN + lm;

    acc_t acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int nn = 0; nn < WN; ++nn)
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[m][nn][r] = 0.f;

    const int G = FLAT ? a.KH : a.KH * a.KW;
    const int nchunks = FLAT ? 1 : a.Cin / CC;
    auto uniform_ptr = [](const float* ptr) {  // a wave-uniform pointer, pinned to scalar registers
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };
    const float* wbase = uniform_ptr(a.w + (size_t)n * a.w_nstride);
    const float* xn = uniform_ptr(a.x + (size_t)n * a.H * a.W * a.Cin);
    const int vy0 = ty0 * a.stride - a.pad_t, vx0 = tx0 * a.stride - a.pad_l;
    const bool has_ab = a.in_a != nullptr;
    const float* ia = has_ab ? a.in_a + (size_t)n * a.in_nstride : nullptr;
    const float* ib = has_ab ? a.in_b + (size_t)n * a.in_nstride : nullptr;
    const int buf_floats = patch_floats + G * LG * BN;  // one (patch, filter) stage
    constexpr int J4N = BN >> 2;                        // float4 per filter row

    // MFMA sweep over the taps x channels of one staged chunk.
    // Fast path: the (tap, k) loop is flattened into groups of two k-steps and software-pipelined
    // through two register sets: the LDS reads of group q+1 are issued before the MFMAs of group q,
    // so a wave never waits on ds_read latency inside a chunk (B advances linearly through the
    // staged filter; A's tap offset is tracked with scalar counters).
    auto sweep = [&](const float* patch, const float* wl) {
        // LG is a whole number of 2-k-step groups for MT = 32 (4, 8, 16, 32) and, for MT = 16, unless CC = 4
        if (!FLAT && (MT == 32 || (LG & 7) == 0)) {
            // Group size: two k-steps for the wide register tiles (>= 4 MFMAs per k-step); four for the narrow ones,
            // whose two-step groups hold only 2-4 matrix instructions -- less than the LDS latency the prefetch of
            // the next group has to cover, so a wave alone on its SIMD (small launches) idled between groups.
            auto run = [&](auto UC) {
                constexpr int U = decltype(UC)::value;
                const int gpt = LG / (U * KSTEP);
                const int ngroups = G * gpt;
                const float* pb = wl + laneB;
                int kh = 0, kw = 0, kkg = 0, qload = 0;
                float a0[U][WM], b0[U][WN], a1[U][WM], b1[U][WN];
                auto load = [&](float (&av)[U][WM], float (&bv)[U][WN]) {
                    const int aoff = (kh * PW + kw * a.dil_x) * S + kkg * U * KSTEP;
                    const float* pa = patch + aoff;
                    const float* pq = pb + qload * U * KSTEP * BN;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
#pragma unroll
                        for (int m = 0; m < WM; ++m) av[u][m] = pa[laneA[m] + u * KSTEP];
#pragma unroll
                        for (int nn = 0; nn < WN; ++nn) bv[u][nn] = pq[u * KSTEP * BN + nn * MT];
                    }
                    ++qload;
                    if (++kkg == gpt) {
                        kkg = 0;
                        if (++kw == a.KW) {
                            kw = 0;
                            ++kh;
                        }
                    }
                };
                auto mma = [&](float (&av)[U][WM], float (&bv)[U][WN]) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int m = 0; m < WM; ++m)
#pragma unroll
                            for (int nn = 0; nn < WN; ++nn) acc[m][nn] = F::mma(av[u][m], bv[u][nn], acc[m][nn]);
                };
                load(a0, b0);
                int q = 0;
                for (; q + 2 <= ngroups; q += 2) {
                    load(a1, b1);
                    mma(a0, b0);
                    if (q + 2 < ngroups) load(a0, b0);
                    mma(a1, b1);
                }
                if (q < ngroups) mma(a0, b0);
            };
            if (WM * WN < 4 && (LG % (4 * KSTEP)) == 0)
                run(std::integral_constant<int, 4>{});
            else
                run(std::integral_constant<int, 2>{});
        } else {
            // General path (the flat Cin == 3 layout; CC = 4 with the 16-row MFMA): one k-step per stage, software-
            // pipelined through two register sets like the fast path -- without it the compiler emits
            // "ds_read, s_waitcnt, mfma" per step and the LDS latency is paid once per matrix instruction.
            const int spg = LG / KSTEP;  // k-steps per tap group (LG is a multiple of KSTEP)
            const int nsteps = G * spg;
            int g = 0, kk = 0;
            float a0[WM], b0[WN], a1[WM], b1[WN];
            auto load = [&](float (&av)[WM], float (&bv)[WN]) {
                const int aoff = FLAT ? g * PW * S : ((g / a.KW) * PW + (g % a.KW) * a.dil_x) * S;
                const float* pa = patch + aoff + kk;
                const float* pb = wl + (g * LG + kk) * BN + laneB;
#pragma unroll
                for (int m = 0; m < WM; ++m) av[m] = pa[laneA[m]];
#pragma unroll
                for (int nn = 0; nn < WN; ++nn) bv[nn] = pb[nn * MT];
                kk += KSTEP;
                if (kk >= LG) {
                    kk = 0;
                    ++g;
                }
            };
            auto mma = [&](const float (&av)[WM], const float (&bv)[WN]) {
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int nn = 0; nn < WN; ++nn) acc[m][nn] = F::mma(av[m], bv[nn], acc[m][nn]);
            };
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
    };

    if constexpr (FLAT) {
        // ---- Cin == 3: one chunk, K runs over (kw,ci) contiguously per kernel row ----
        float* patch = smem;
        float* wl = smem + patch_floats;
        for (int pix = tid; pix < PH * PW; pix += 256) {
            const int py = fdiv(pix, inv_pw), px = pix - py * PW;
            int sy, sx;
            const bool ok = src_coord(a.src_mode, a.refl, vy0 + py, a.H, sy) &&
                            src_coord(a.src_mode, a.refl, vx0 + px, a.W, sx);
            float v[3] = {0.f, 0.f, 0.f};
            if (ok) {
                const float* src = xn + ((size_t)sy * a.W + sx) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float t = src[c];
                    if (has_ab) t = fmaf(t, ia[c], ib[c]);
                    if (a.in_relu) t = fmaxf(t, 0.f);
                    v[c] = t;
                }
            }
            patch[pix * 3 + 0] = v[0];
            patch[pix * 3 + 1] = v[1];
            patch[pix * 3 + 2] = v[2];
        }
        if (tid < 8) patch[PH * PW * S + tid] = 0.f;  // slack read by the zero-weight k padding
        const bool vec = (a.Cout & 3) == 0;
        for (int e = tid; e < G * LG * J4N; e += 256) {
            const int k = e / J4N, j4 = e - k * J4N;
            const int g = k / LG, r = k - g * LG;
            const int co = co0 + j4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < a.KW * a.Cin) {
                const float* src = wbase + ((size_t)g * a.KW * a.Cin + r) * a.Cout + co;
                if (vec && co + 3 < a.Cout) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (co + 0 < a.Cout) v.x = src[0];
                    if (co + 1 < a.Cout) v.y = src[1];
                    if (co + 2 < a.Cout) v.z = src[2];
                    if (co + 3 < a.Cout) v.w = src[3];
                }
            }
            *reinterpret_cast<float4*>(wl + e * 4) = v;
        }
        __syncthreads();
        sweep(patch, wl);
    } else {
        // ---- software pipeline over input-channel chunks: two LDS stages; the next chunk's global
        // loads are issued BEFORE the MFMA sweep of the current one and committed to the other stage
        // after it (one barrier per chunk) ----
        constexpr int PMAX = 6, WMAX = 6;  // float4 elements per thread and chunk (plan guarantees the bound)
        const int c4n = CC >> 2;
        const int c4sh = c4n == 1 ? 0 : (c4n == 2 ? 1 : (c4n == 4 ? 2 : 3));
        const int lgsh = c4sh + 2;  // log2(CC)
        const int ne_p = PH * PW * c4n, ne_w = G * LG * J4N;
        float* abl = smem + 2 * buf_floats;  // [2][Cin] on-load affine
        if (has_ab)
            for (int c = tid; c < a.Cin; c += 256) {
                abl[c] = ia[c];
                abl[a.Cin + c] = ib[c];
            }
        // Staging is branch-free.  Global reads go through buffer resources: an element outside the image (zero
        // padding) or outside this thread's share gets the offset kOOB, which the hardware range check turns into a
        // load of zeros -- no predicate, no select.  LDS writes of elements the thread does not own go to the
        // patch's slack floats (as zeros).  The MFMA pipe idles while a workgroup stages, so every instruction
        // removed here is time gained (tools/conv_trace.py).
        constexpr unsigned kOOB = 0x80000000u;
        unsigned gvo[PMAX];  // byte offset of this thread's i-th patch element in the image, or kOOB
        int pdst[PMAX];      // its LDS position (floats, relative to the stage)
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int e = tid + i * 256;
            gvo[i] = kOOB;
            pdst[i] = PH * PW * S;  // slack
            if (e < ne_p) {
                const int pix = e >> c4sh, c4 = e & (c4n - 1);
                const int py = fdiv(pix, inv_pw), px = pix - py * PW;
                int sy, sx;
                const bool ok = src_coord(a.src_mode, a.refl, vy0 + py, a.H, sy) &&
                                src_coord(a.src_mode, a.refl, vx0 + px, a.W, sx);
                if (ok) gvo[i] = (unsigned)((sy * a.W + sx) * a.Cin + c4 * 4) * 4u;
                pdst[i] = pix * S + c4 * 4;
            }
        }
        float4 pv[PMAX], wv[WMAX];
        const bool vec = (a.Cout & 3) == 0;
        const bool wfast = vec && co0 + BN <= a.Cout;  // whole filter rows in range: precomputed offsets
        unsigned wvo[WMAX];  // byte offset of the i-th filter element (chunk 0), or kOOB
        int wdst[WMAX];      // LDS position (floats, relative to the stage's filter area)
#pragma unroll
        for (int i = 0; i < WMAX; ++i) {
            const int e = tid + i * 256;
            wvo[i] = kOOB;
            wdst[i] = -4;  // the last four slack floats of the patch area (16-byte aligned)
            if (e < ne_w) {
                const int k = e / J4N, j4 = e - k * J4N;
                wvo[i] = (unsigned)(((k >> lgsh) * a.Cin + (k & (CC - 1))) * a.Cout + co0 + j4 * 4) * 4u;
                wdst[i] = e * 4;
            }
        }
        const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * a.Cin) * 4u);
        const unsigned w_bytes = __builtin_amdgcn_readfirstlane((unsigned)(G * a.Cin * a.Cout) * 4u);
        auto issue = [&](int chunk) {
            const int ci0 = chunk * CC;
            // (descriptors are rebuilt from the scalar base pointers at every call: kept across the loop the compiler
            // parks them in vector registers and wraps every load in a readfirstlane loop)
            const __amdgpu_buffer_rsrc_t xr =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(xn)), 0, x_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t wr =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(wbase)), 0, w_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < PMAX; ++i)
                pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, gvo[i], ci0 * 4, 0));
            if (wfast) {
                const int wso = ci0 * a.Cout * 4;
#pragma unroll
                for (int i = 0; i < WMAX; ++i)
                    wv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, wvo[i], wso, 0));
                return;
            }
#pragma unroll
            for (int i = 0; i < WMAX; ++i) {
                const int e = tid + i * 256;
                wv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < ne_w) {
                    const int k = e / J4N, j4 = e - k * J4N;
                    const int g = k >> lgsh, r = k & (CC - 1);
                    const int co = co0 + j4 * 4;
                    const float* src = wbase + ((size_t)g * a.Cin + ci0 + r) * a.Cout + co;
                    if (vec && co + 3 < a.Cout) {
                        wv[i] = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (co + 0 < a.Cout) wv[i].x = src[0];
                        if (co + 1 < a.Cout) wv[i].y = src[1];
                        if (co + 2 < a.Cout) wv[i].z = src[2];
                        if (co + 3 < a.Cout) wv[i].w = src[3];
                    }
                }
            }
        };
        // ReLU as ONE instruction (v_med3_f32); fmaxf would add a canonicalising max per value
        auto relu1 = [](float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); };
        auto commit_as = [&](auto AB, auto RELU, int chunk, float* patch, float* wl) {
            const int ci0 = chunk * CC;
#pragma unroll
            for (int i = 0; i < PMAX; ++i) {
                float4 v = pv[i];
                if (decltype(AB)::value) {  // producer instance norm folded into the load; padding stays zero
                    const int c4 = (tid + i * 256) & (c4n - 1);
                    const float* pa_ = abl + ci0 + c4 * 4;
                    const float* pb_ = pa_ + a.Cin;
                    // padding arrives as 0 and must stay 0: clear the shift with a bit mask (fma(0, a, 0) = 0);
                    // a select here makes the compiler branch around the table reads, once per component
                    const unsigned okm = gvo[i] != kOOB ? 0xFFFFFFFFu : 0u;
                    const float4 sc = *reinterpret_cast<const float4*>(pa_);
                    const uint4 sh = *reinterpret_cast<const uint4*>(pb_);
                    v.x = fmaf(v.x, sc.x, __uint_as_float(sh.x & okm));
                    v.y = fmaf(v.y, sc.y, __uint_as_float(sh.y & okm));
                    v.z = fmaf(v.z, sc.z, __uint_as_float(sh.z & okm));
                    v.w = fmaf(v.w, sc.w, __uint_as_float(sh.w & okm));
                }
                if (decltype(RELU)::value) {
                    v.x = relu1(v.x);
                    v.y = relu1(v.y);
                    v.z = relu1(v.z);
                    v.w = relu1(v.w);
                }
                float* d = patch + pdst[i];
                d[0] = v.x;
                d[1] = v.y;
                d[2] = v.z;
                d[3] = v.w;
            }
#pragma unroll
            for (int i = 0; i < WMAX; ++i) *reinterpret_cast<float4*>(wl + wdst[i]) = wv[i];
        };
        const int cmode = (has_ab ? 2 : 0) + (a.in_relu ? 1 : 0);  // wave-uniform: one straight-line variant each
        auto commit = [&](int chunk, float* patch, float* wl) {
            if (cmode == 0)
                commit_as(std::false_type{}, std::false_type{}, chunk, patch, wl);
            else if (cmode == 1)
                commit_as(std::false_type{}, std::true_type{}, chunk, patch, wl);
            else if (cmode == 2)
                commit_as(std::true_type{}, std::false_type{}, chunk, patch, wl);
            else
                commit_as(std::true_type{}, std::true_type{}, chunk, patch, wl);
        };
        // split-K: blockIdx.z owns the chunk range [cbeg, cend)
        const int cbeg = __builtin_amdgcn_readfirstlane(p.ksplit > 1 ? (int)blockIdx.z * nchunks / p.ksplit : 0);
        const int cend = __builtin_amdgcn_readfirstlane(p.ksplit > 1 ? ((int)blockIdx.z + 1) * nchunks / p.ksplit : nchunks);
        issue(cbeg);
        __syncthreads();  // abl visible
        commit(cbeg, smem, smem + patch_floats);
        __syncthreads();
#ifdef FS_CONV_TRACE
        tr_pro = FS_TRACE_NOW();
#endif
        for (int chunk = cbeg; chunk < cend; ++chunk) {
            float* cur = smem + ((chunk - cbeg) & 1) * buf_floats;
            float* nxt = smem + ((chunk - cbeg + 1) & 1) * buf_floats;
            const bool more = chunk + 1 < cend;
#ifdef FS_CONV_TRACE
            const long long q0 = FS_TRACE_NOW();
#endif
            if (more) issue(chunk + 1);
            sweep(cur, cur + patch_floats);
#ifdef FS_CONV_TRACE
            const long long q1 = FS_TRACE_NOW();
#endif
            if (more) commit(chunk + 1, nxt, nxt + patch_floats);
#ifdef FS_CONV_TRACE
            const long long q2 = FS_TRACE_NOW();
#endif
            __syncthreads();
#ifdef FS_CONV_TRACE
            const long long q3 = FS_TRACE_NOW();
            tr_sweep += q1 - q0;
            tr_commit += q2 - q1;
            tr_bar += q3 - q2;
#endif
        }
    }
#ifdef FS_CONV_TRACE
    const long long tr_main = FS_TRACE_NOW();
#endif

    // ---- epilogue ----
    const int th_valid = min(p.TH, a.Ho - ty0), tw_valid = min(p.TW, a.Wo - tx0);
    // The accumulator registers of a lane come in groups of four consecutive tile rows (Frag::row).  Walk them with
    // ONE division per group (float reciprocal: exact for these small integers) and increments inside the group:
    // fn(m, r, ok, py, px) with (py, px) the pixel of the workgroup tile that acc[m][.][r] belongs to.
    const float inv_tw = 1.0f / (float)p.TW;
    auto for_rows = [&](auto&& fn) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int rg = 0; rg < NACC; rg += 4) {
                const int t = (wave * WM + m) * MT + F::row(rg, lane);
                int py = (int)(((float)t + 0.5f) * inv_tw), px = t - py * p.TW;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fn(m, rg + j, t + j < tile_px && py < th_valid && px < tw_valid, py, px);
                    ++px;
                    if (px == p.TW) {
                        px = 0;
                        ++py;
                    }
                }
            }
    };
    if (a.stats) {
        // Per-tile statistics in ONE pass over the accumulators: sums of (x - c) and (x - c)^2 around a per-channel
        // shift c taken from the tile itself (its first pixel, always valid), so that M2 = S2 - S1^2/n has no
        // cancellation to speak of (|mean - c| is of the order of the standard deviation).  Two barriers instead
        // of four and one sweep over the registers instead of two.
        float* red = smem;                 // [4 waves][2][BN]
        float* shiftl = smem + 8 * BN;     // [BN]
        if constexpr (FLAT) __syncthreads();  // (the chunked loop ends on a barrier; the flat path's sweep does not)
        if (wave == 0 && kq == 0)
#pragma unroll
            for (int nn = 0; nn < WN; ++nn) shiftl[nn * MT + lm] = acc[0][nn][0];
        __syncthreads();
        float cs[WN], s1[WN], s2[WN];
#pragma unroll
        for (int nn = 0; nn < WN; ++nn) {
            cs[nn] = shiftl[nn * MT + lm];
            s1[nn] = 0.f;
            s2[nn] = 0.f;
        }
        for_rows([&](int m, int r, bool ok, int, int) {
#pragma unroll
            for (int nn = 0; nn < WN; ++nn) {
                const float d = ok ? acc[m][nn][r] - cs[nn] : 0.f;
                s1[nn] += d;
                s2[nn] = fmaf(d, d, s2[nn]);
            }
        });
#pragma unroll
        for (int nn = 0; nn < WN; ++nn) {
            s1[nn] += __shfl_xor(s1[nn], 32);
            s2[nn] += __shfl_xor(s2[nn], 32);
            if (MT == 16) {
                s1[nn] += __shfl_xor(s1[nn], 16);
                s2[nn] += __shfl_xor(s2[nn], 16);
            }
        }
        if (lane < MT)
#pragma unroll
            for (int nn = 0; nn < WN; ++nn) {
                red[(wave * 2 + 0) * BN + nn * MT + lane] = s1[nn];
                red[(wave * 2 + 1) * BN + nn * MT + lane] = s2[nn];
            }
        __syncthreads();
        if (tid < BN && co0 + tid < a.Cout) {
            const float cnt = (float)(th_valid * tw_valid);
            const float S1 = (red[0 * BN + tid] + red[2 * BN + tid]) + (red[4 * BN + tid] + red[6 * BN + tid]);
            const float S2 = (red[1 * BN + tid] + red[3 * BN + tid]) + (red[5 * BN + tid] + red[7 * BN + tid]);
            float* st = a.stats + ((size_t)tile_id * a.Cout + co0 + tid) * 3;
            st[0] = shiftl[tid] + S1 / cnt;
            st[1] = fmaxf(S2 - S1 * S1 / cnt, 0.f);
            st[2] = cnt;
        }
    }

    // store: per-image base pointers are 64-bit scalars, everything per element is 32-bit
    const int Cr = a.shuffle ? a.Cout >> 2 : a.Cout;
    const int SH = a.shuf_H > 0 ? a.shuf_H : 2 * a.Ho, SW = a.shuf_W > 0 ? a.shuf_W : 2 * a.Wo;  // shuffled extent
    float* yn = a.shuffle ? a.y + (size_t)n * SH * SW * Cr
                          : a.y + ((size_t)n + (p.ksplit > 1 ? (size_t)blockIdx.z * a.N : 0)) * a.Ho * a.Wo * a.Cout;
    const int ap = a.add_pad;
    const int aW = a.Wo - 2 * ap;
    const float* asn = a.add_src ? a.add_src + (size_t)n * (a.Ho - 2 * ap) * aW * a.Cout : nullptr;
    const float* msn = a.mask_src ? a.mask_src + (size_t)n * a.Ho * a.Wo * a.Cout : nullptr;
    int cof[WN], qa[WN], qb[WN];  // per n-tile: channel offset inside a pixel, pixel-shuffle phase
    float bs[WN];
    bool cok[WN];
#pragma unroll
    for (int nn = 0; nn < WN; ++nn) {
        const int co = co0 + nn * MT + lm;
        cok[nn] = co < a.Cout;
        const int q = a.shuffle ? co / Cr : 0;
        cof[nn] = a.shuffle ? co - q * Cr : co;
        qa[nn] = q >> 1;
        qb[nn] = q & 1;
        bs[nn] = (a.bias && cok[nn]) ? a.bias[co] : 0.f;
    }
    const bool relu_out = a.out_relu != 0;
    if (!a.shuffle && !asn) {
        // the common stores (every VGG conv, most transform-net convs): bias, ReLU floor, optional consumer mask.
        // Offsets advance by additions: row offset + column offset, no multiply per element.
        const int row_stride = a.Wo * a.Cout;
        const int Hp = (a.Ho + 1) >> 1, Wp = (a.Wo + 1) >> 1;
        const float* rsn = a.route_src ? a.route_src + (size_t)n * Hp * Wp * a.Cout : nullptr;
        auto run = [&](auto MASKED, auto FULL, auto ROUTED) {
            constexpr bool kFull = decltype(FULL)::value;  // every lane has a valid pixel and channel: no per-element predicate
            constexpr bool kRoute = decltype(ROUTED)::value;  // max-pool gradient routing through the mask tensor (needs MASKED)
            constexpr int RB = kRoute ? 4 : (NACC < 8 ? NACC : 8);  // rows per batch (bounds the registers of the mask batch)
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int r0 = 0; r0 < NACC; r0 += RB) {
                    int off[RB];  // element offset of the pixel of row r0+i (channel 0); -1: outside the tile / image
                    int par[RB];  // kRoute: bit 0 = column parity, bit 1 = row parity, bit 2 / 3 = the horizontal / vertical window partner exists
                    int pof[RB];  // kRoute: element offset of the pixel's pooling window in route_src
#pragma unroll
                    for (int rg = 0; rg < RB; rg += 4) {
                        const int t = (wave * WM + m) * MT + F::row(r0 + rg, lane);
                        int py = (int)(((float)t + 0.5f) * inv_tw), px = t - py * p.TW;
                        int roff = (ty0 + py) * row_stride, coff = (tx0 + px) * a.Cout;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            off[rg + j] = (kFull || (t + j < tile_px && py < th_valid && px < tw_valid)) ? roff + coff : -1;
                            if (kRoute) {
                                const int oy = ty0 + py, ox = tx0 + px;
                                par[rg + j] = (ox & 1) | ((oy & 1) << 1) | (((ox ^ 1) < a.Wo) ? 4 : 0) | (((oy ^ 1) < a.Ho) ? 8 : 0);
                                pof[rg + j] = ((oy >> 1) * Wp + (ox >> 1)) * a.Cout;
                            }
                            ++px;
                            coff += a.Cout;
                            if (px == p.TW) {
                                px = 0;
                                ++py;
                                coff = tx0 * a.Cout;
                                roff += row_stride;
                            }
                        }
                    }
                    // consumer-ReLU mask: ALL loads of the batch first (unconditional, clamped to element 0 where the
                    // lane has nothing to store), so that their latencies overlap instead of chaining
                    // load -> wait -> store per element
                    float mk[RB][WN];
                    float nA[kRoute ? RB : 1][WN], nB[kRoute ? RB : 1][WN], nC[kRoute ? RB : 1][WN], da[kRoute ? RB : 1][WN];
                    if (decltype(MASKED)::value) {
#pragma unroll
                        for (int i = 0; i < RB; ++i)
#pragma unroll
                            for (int nn = 0; nn < WN; ++nn) {
                                const bool live = kFull || (off[i] >= 0 && cok[nn]);
                                const int o = live ? off[i] + cof[nn] : 0;
                                mk[i][nn] = msn[o];
                                if (kRoute) {
                                    // the three other pixels of the 2x2 window (clamped to the pixel itself where the image ends)
                                    const int dx = (par[i] & 1) ? -a.Cout : a.Cout, dyo = (par[i] & 2) ? -row_stride : row_stride;
                                    const bool hA = live && (par[i] & 4), hB = live && (par[i] & 8);
                                    nA[i][nn] = msn[hA ? o + dx : o];
                                    nB[i][nn] = msn[hB ? o + dyo : o];
                                    nC[i][nn] = msn[(hA && hB) ? o + dx + dyo : o];
                                    da[i][nn] = rsn[live ? pof[i] + cof[nn] : 0];
                                }
                            }
                    }
#pragma unroll
                    for (int i = 0; i < RB; ++i) {
                        if (!kFull && off[i] < 0) continue;
#pragma unroll
                        for (int nn = 0; nn < WN; ++nn) {
                            if (!kFull && !cok[nn]) continue;
                            float v = acc[m][nn][r0 + i] + bs[nn];
                            v = relu_out ? fmaxf(v, 0.f) : v;
                            if (kRoute) {
                                // tf.nn.max_pool's gradient goes to the FIRST maximum of the window in row-major order: a partner
                                // that comes earlier wins ties, one that comes later must be strictly greater
                                const float me = mk[i][nn];
                                const bool cx = par[i] & 1, cy = par[i] & 2;
                                bool win = true;
                                if (par[i] & 4) win = win && !(cx ? nA[i][nn] >= me : nA[i][nn] > me);
                                if (par[i] & 8) win = win && !(cy ? nB[i][nn] >= me : nB[i][nn] > me);
                                if ((par[i] & 12) == 12) win = win && !(cy ? nC[i][nn] >= me : nC[i][nn] > me);
                                if (win) v += da[i][nn];
                            }
                            if (decltype(MASKED)::value) v = mk[i][nn] > 0.f ? v : 0.f;
                            yn[off[i] + cof[nn]] = v;
                        }
                    }
                }
        };
        const bool full = th_valid == p.TH && tw_valid == p.TW && tile_px == 4 * WM * MT && co0 + BN <= a.Cout;
        if (msn && rsn) {
            if (full)
                run(std::true_type{}, std::true_type{}, std::true_type{});
            else
                run(std::true_type{}, std::false_type{}, std::true_type{});
        } else if (msn) {
            if (full)
                run(std::true_type{}, std::true_type{}, std::false_type{});
            else
                run(std::true_type{}, std::false_type{}, std::false_type{});
        } else {
            if (full)
                run(std::false_type{}, std::true_type{}, std::false_type{});
            else
                run(std::false_type{}, std::false_type{}, std::false_type{});
        }
    } else if (!a.shuffle && !msn) {
        // residual-gradient add (the W1 dgrads of the residual blocks): the cropped addend is loaded for a batch of
        // rows before the first store, like the consumer mask above
        const int row_stride = a.Wo * a.Cout;
        constexpr int RB = NACC < 8 ? NACC : 8;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r0 = 0; r0 < NACC; r0 += RB) {
                int off[RB], aof[RB];  // output / addend element offsets of the row's pixel (channel 0); -1: none
#pragma unroll
                for (int rg = 0; rg < RB; rg += 4) {
                    const int t = (wave * WM + m) * MT + F::row(r0 + rg, lane);
                    int py = (int)(((float)t + 0.5f) * inv_tw), px = t - py * p.TW;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = t + j < tile_px && py < th_valid && px < tw_valid;
                        const int oy = ty0 + py, ox = tx0 + px;
                        const bool inner = ok && oy >= ap && oy < a.Ho - ap && ox >= ap && ox < a.Wo - ap;
                        off[rg + j] = ok ? oy * row_stride + ox * a.Cout : -1;
                        aof[rg + j] = inner ? ((oy - ap) * aW + (ox - ap)) * a.Cout : -1;
                        ++px;
                        if (px == p.TW) {
                            px = 0;
                            ++py;
                        }
                    }
                }
                float ad[RB][WN];
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int nn = 0; nn < WN; ++nn) ad[i][nn] = asn[(aof[i] >= 0 && cok[nn]) ? aof[i] + cof[nn] : 0];
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    if (off[i] < 0) continue;
#pragma unroll
                    for (int nn = 0; nn < WN; ++nn) {
                        if (!cok[nn]) continue;
                        float v = acc[m][nn][r0 + i] + bs[nn];
                        v = relu_out ? fmaxf(v, 0.f) : v;
                        if (aof[i] >= 0) v += ad[i][nn];
                        yn[off[i] + cof[nn]] = v;
                    }
                }
            }
    } else {
        for_rows([&](int m, int r, bool ok, int py, int px) {
            if (!ok) return;
            const int oy = ty0 + py, ox = tx0 + px;
            const bool inner = asn && oy >= ap && oy < a.Ho - ap && ox >= ap && ox < a.Wo - ap;
            const int aoff = ((oy - ap) * aW + (ox - ap)) * a.Cout;
            const int poff = (oy * a.Wo + ox) * a.Cout;
#pragma unroll
            for (int nn = 0; nn < WN; ++nn) {
                if (!cok[nn]) continue;
                float v = acc[m][nn][r] + bs[nn];
                v = relu_out ? fmaxf(v, 0.f) : v;
                if (inner) v += asn[aoff + cof[nn]];
                int o;
                if (a.shuffle) {
                    if (2 * oy + qa[nn] >= SH || 2 * ox + qb[nn] >= SW) continue;  // odd extents: the last row/column is clipped
                    o = ((2 * oy + qa[nn]) * SW + 2 * ox + qb[nn]) * Cr + cof[nn];
                } else
                    o = poff + cof[nn];
                if (msn) v = msn[o] > 0.f ? v : 0.f;
                yn[o] = v;
            }
        });
    }
#ifdef FS_CONV_TRACE
    {
        const long long tr_end = FS_TRACE_NOW();
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (tid == 0 && lin < 4096) {
            long long* t = g_conv_trace + lin * 8;
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            t[0] = tr_t0;
            t[1] = tr_pro - tr_t0;
            t[2] = tr_sweep;
            t[3] = tr_commit;
            t[4] = tr_bar;
            t[5] = tr_end - tr_main;
            t[6] = tr_end;
            t[7] = hw;
        }
    }
#endif
}

// -------------------------------------------------------------------------------------- host
#ifdef FS_CONV_TRACE
extern "C" int fs_debug_conv_trace_reset() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_conv_trace)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(long long) * 8 * 4096);
}
extern "C" int fs_debug_conv_trace(long long* out, int n_wg) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_trace), sizeof(long long) * 8 * (size_t)n_wg, 0, hipMemcpyDeviceToHost);
}
#endif
const char* prof_family_name(int f) {
    static const char* const names[Profiler::kFamilies] = {
        "conv_igemm_kernel<32,2,2>", "conv_igemm_kernel<32,2,1>", "conv_igemm_kernel<16,4,1>", "conv_igemm_kernel<32,1,2>",
        "conv_igemm_kernel<32,1,1>", "wino_conv_kernel", "wino2_conv_kernel (VGG16 convs)", "wino2_conv_kernel (transform-net residual convs)",
        "conv_stream_kernel", "conv3x3_to3_kernel", "wgrad2_kernel", "conv_wgrad_kernel", "gram_stream_kernel",
        "conv_wgrad_kernel (Gram forward)", "gram_bwd_kernel", "conv_igemm_kernel (Gram backward, 1x1 per-sample filters)",
        "wino2h_conv_kernel (transform-net residual convs, half items)", "conv_s16_kernel", "wino4_conv_kernel", "wgw_kernel",
        "wino4t_conv_kernel (transform-net residual convs)", "wino4t_conv_kernel (VGG16 convs)",
        "wino6 pipeline (VGG16 convs: input transform + split-bf16 GEMMs + output transform)"};
    return f >= 0 && f < Profiler::kFamilies ? names[f] : "";
}

Profiler*& Profiler::current() {
    static thread_local Profiler* p = nullptr;
    return p;
}
void Profiler::begin(int fam, double flops, hipStream_t s) {
    if (n == cap) {
        const int ncap = cap ? cap * 2 : 1024;
        Rec* nr = (Rec*)realloc(recs, sizeof(Rec) * ncap);
        if (!nr) return;
        recs = nr;
        for (int i = cap; i < ncap; ++i) {
            (void)hipEventCreate(&recs[i].a);
            (void)hipEventCreate(&recs[i].b);
        }
        cap = ncap;
    }
    recs[n].fam = fam;
    recs[n].flops = flops;
    (void)hipEventRecord(recs[n].a, s);
}
void Profiler::end(hipStream_t s) {
    if (n < cap) (void)hipEventRecord(recs[n++].b, s);
}
int Profiler::collect(double out[kFamilies][3]) {
    for (int f = 0; f < kFamilies; ++f) out[f][0] = out[f][1] = out[f][2] = 0;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(recs[i].b) != hipSuccess) return -1;
        float ms = 0;
        if (hipEventElapsedTime(&ms, recs[i].a, recs[i].b) != hipSuccess) return -1;
        out[recs[i].fam][0] += 1;
        out[recs[i].fam][1] += recs[i].flops;
        out[recs[i].fam][2] += ms;
    }
    return 0;
}
void Profiler::reset() { n = 0; }
Profiler::~Profiler() {
    for (int i = 0; i < cap; ++i) {
        (void)hipEventDestroy(recs[i].a);
        (void)hipEventDestroy(recs[i].b);
    }
    free(recs);
}

static int env_int(const char* name, int dflt) { return tune_int(name, dflt); }

void plan_tile(int Ho, int Wo, int KH, int KW, int stride, int max_px, int* TH, int* TW) {
    double best = -1;
    int bth = 1, btw = 1;
    for (int tw = 1; tw <= Wo && tw <= max_px; ++tw) {
        // only widths that split Wo evenly-ish: tw = ceil(Wo / k)
        const int k = cdiv(Wo, tw);
        if (tw != cdiv(Wo, k)) continue;
        int th = max_px / tw;
        if (th > Ho) th = Ho;
        if (th < 1) continue;
        th = cdiv(Ho, cdiv(Ho, th));  // shrink to the even split
        const double eff = (double)Ho * Wo / ((double)cdiv(Ho, th) * cdiv(Wo, tw) * max_px);
        const double halo = (double)((th - 1) * stride + KH) * ((tw - 1) * stride + KW) / ((double)th * tw * stride * stride);
        const double score = eff / (1.0 + 0.1 * (halo - 1.0));
        if (score > best) {
            best = score;
            bth = th;
            btw = tw;
        }
    }
    *TH = bth;
    *TW = btw;
}

// variant -> (MFMA tile, m-tiles per wave, n-tiles per wave); pixels per workgroup = 4*WM*MT
static const int kVarMT[5] = {32, 32, 16, 32, 32};
static const int kVarWM[5] = {2, 2, 4, 1, 1};
static const int kVarWN[5] = {2, 1, 1, 2, 1};

static void plan_variant(const ConvArgs& a, int variant, ConvPlan* out) {
    ConvPlan p{};
    p.flat = a.Cin == 3;
    p.variant = variant;
    p.BN = kVarMT[variant] * kVarWN[variant];
    const int max_px = 4 * kVarWM[variant] * kVarMT[variant];
    const int kstep = kVarMT[variant] == 16 ? 4 : 2;
    const int budget = env_int("FS_CONV_LDS_KB", 36) * 1024;  // per pipeline stage (two stages + affine table)
    for (int px = max_px; px >= 16; px >>= 1) {  // shrink the pixel tile until a channel chunk fits
        plan_tile(a.Ho, a.Wo, a.KH, a.KW, a.stride, px, &p.TH, &p.TW);
        p.tiles_y = cdiv(a.Ho, p.TH);
        p.tiles_x = cdiv(a.Wo, p.TW);
        p.PH = (p.TH - 1) * a.stride + a.KH;
        p.PW = (p.TW - 1) * a.stride + (a.KW - 1) * (a.dil_x > 0 ? a.dil_x : 1) + 1;
        if (p.flat) {
            p.CC = 3;
            p.S = 3;
            p.LG = cdiv(a.KW * 3, kstep) * kstep;
            const int G = a.KH;
            p.lds_bytes = 4 * (((p.PH * p.PW * p.S + 8 + 3) & ~3) + G * p.LG * p.BN);
            break;
        }
        const int G = a.KH * a.KW;
        const int forced = env_int("FS_CONV_CC", 0);
        int chosen = 0, chosen_bytes = 0;
        for (int cc = 32; cc >= 4; cc >>= 1) {
            if (a.Cin % cc) continue;
            const int stage = 4 * (((p.PH * p.PW * (cc + 1) + 8 + 3) & ~3) + G * cc * p.BN);
            const int bytes = 2 * stage + 8 * a.Cin;
            // the pipelined kernel keeps <= 6 float4 of patch and of filter per thread in flight
            const bool fits = p.PH * p.PW * (cc / 4) <= 6 * 256 && G * cc * (p.BN / 4) <= 6 * 256;
            if (forced == cc || (!forced && fits && stage <= budget)) {
                chosen = cc;
                chosen_bytes = bytes;
                break;
            }
        }
        p.CC = chosen;
        p.S = chosen + 1;
        p.LG = chosen;
        p.lds_bytes = chosen_bytes;
        if (chosen) break;
    }
    if (p.lds_bytes < 4 * 9 * p.BN) p.lds_bytes = 4 * 9 * p.BN;  // stats scratch
    p.ksplit = 1;
    p.xcd_swizzle = env_int("FS_CONV_XCD", 1);
    p.skew = env_int("FS_CONV_SKEW", 0);
    *out = p;
}

// The specialised conv kernel families in order of preference (each takes a launch only when the caller provided ITS filter layout / scratch and the shape
// fits; what falls through runs on conv_igemm_kernel below).  Generations of the Winograd kernels: WinoGen in fs_kernels.h.
const ConvFamily* conv_families(int* n) {
    static const ConvFamily kFamilies[] = {
        {"wino6: split-bf16 F(4x4,3x3) pipeline (deep VGG16 layers)", 12, true, wino6_eligible, wino6_plan, wino6_launch},
        {"wino4: F(4x4,3x3), filter through LDS (FS_WINO_V=4)", 10, true, wino4_eligible, wino4_plan, wino4_launch},
        {"wino4t: F(4x4,3x3), filter in registers", 11, true, wino4t_eligible, wino4t_plan, wino4t_launch},
        {"wino2h: F(2x2,3x3), half items (small grids)", 8, true, wino2h_eligible, wino2h_plan, wino2h_launch},
        {"wino2: F(2x2,3x3), second generation", 6, true, wino2_eligible, wino2_plan, wino2_launch},
        {"wino: F(2x2,3x3), first kernel", 5, true, wino_eligible, wino_plan, wino_launch},
        {"cstream: narrow full-resolution layers, persistent streaming", 7, false, cstream_eligible, cstream_plan, cstream_launch},
        {"s16: 16-output-channel blocks (9x9 image layer, folded output layer)", 9, false, s16_eligible, s16_plan, s16_launch},
    };
    *n = (int)(sizeof(kFamilies) / sizeof(kFamilies[0]));
    return kFamilies;
}

ConvPlan conv_plan(const ConvArgs& a) {
    ConvPlan p;
    {   // the specialised families, in the table's order (first taker wins)
        int nf = 0;
        const ConvFamily* fam = conv_families(&nf);
        const bool wino_on = env_int("FS_CONV_WINO", 1) != 0;
        for (int i = 0; i < nf; ++i)
            if ((wino_on || !fam[i].winograd) && fam[i].eligible(a)) {
                fam[i].plan(a, &p);
                return p;
            }
    }
    if (a.Cout <= 16 || a.Cin == 3) {  // narrow outputs, and the flat Cin==3 path, have one variant each
        plan_variant(a, a.Cout <= 16 ? 2 : 0, &p);
        return p;
    }
    // widest tile first; halve the workgroup tile while the launch cannot fill the chip
    // (256 CUs x >= 2 workgroups), e.g. VGG conv4_x at batch 4 or the 64-channel residual convs
    const int min_wgs = env_int("FS_CONV_MIN_WGS", 512);
    const int forced_variant = env_int("FS_CONV_VARIANT", -1);  // tuning aid (tools/micro_conv.py)
    if (forced_variant >= 0 && forced_variant <= 4 && forced_variant != 2) {
        plan_variant(a, forced_variant, &p);
        return p;
    }
    // Deep-K layers on a small pixel grid (VGG conv4_x at batch 4: 4096 px x 512 co, K = 4608) cannot fill
    // the chip with wide tiles; the narrow tiles that do fill it move twice the bytes per FLOP through LDS.
    // With scratch available, keep the wide tile and split the input-channel chunks over blockIdx.z instead;
    // a streaming epilogue kernel sums the partials and applies bias / ReLU / tap-add / mask.
    const int max_split = env_int("FS_CONV_KSPLIT", 4);
    if (a.split_ws && max_split > 1 && a.Cout > 32 && !a.stats && !a.shuffle && !a.add_pad && a.w_nstride == 0) {
        const int force = env_int("FS_CONV_FORCE_KSPLIT", 0);  // test hook: split regardless of the grid size
        if (force > 1) {
            plan_variant(a, 0, &p);
            const int nchunks = p.CC > 0 ? a.Cin / p.CC : 0;
            const int ks = force < nchunks ? force : nchunks;
            if (ks > 1 && (size_t)ks * a.N * a.Ho * a.Wo * a.Cout <= a.split_ws_floats) {
                p.ksplit = ks;
                return p;
            }
        }
        const int try_order[2] = {0, 3};
        for (int i = 0; i < 2; ++i) {
            plan_variant(a, try_order[i], &p);
            if (p.CC <= 0) continue;
            const long wgs = (long)a.N * p.tiles_y * p.tiles_x * cdiv(a.Cout, p.BN);
            const int nchunks = a.Cin / p.CC;
            int ks = (int)((min_wgs + wgs - 1) / wgs);
            if (i == 0 && wgs * env_int("FS_CONV_KSPLIT_FILL", 4) > min_wgs) break;  // the plain variants already come close: not worth the extra pass
            if (ks > max_split) ks = max_split;
            while (ks > 1 && (nchunks / ks) * p.CC * a.KH * a.KW < 1024) --ks;  // keep >= 1024 of K per split
            if (ks > 1 && wgs * ks >= min_wgs && (size_t)ks * a.N * a.Ho * a.Wo * a.Cout <= a.split_ws_floats) {
                p.ksplit = ks;
                return p;
            }
        }
    }
    const int order_wide[3] = {0, 3, 4}, order_narrow[2] = {1, 4};
    const int* order = a.Cout > 32 ? order_wide : order_narrow;
    const int n = a.Cout > 32 ? 3 : 2;
    for (int i = 0; i < n; ++i) {
        plan_variant(a, order[i], &p);
        const long wgs = (long)a.N * p.tiles_y * p.tiles_x * cdiv(a.Cout, p.BN);
        if (wgs >= min_wgs) break;
    }
    return p;
}

// Sum of the split-K partials + the conv epilogue (bias, tap-gradient add, consumer ReLU mask, ReLU); float4 per thread.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ part, int ks, size_t n4, int C,
                                                              const float* __restrict__ bias, const float* __restrict__ add_src,
                                                              const float* __restrict__ mask_src, int relu,
                                                              float* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4* p4 = reinterpret_cast<const float4*>(part);
    float4 v = p4[i];
    for (int k = 1; k < ks; ++k) {
        const float4 u = p4[i + (size_t)k * n4];
        v.x += u.x;
        v.y += u.y;
        v.z += u.z;
        v.w += u.w;
    }
    if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + (int)((i * 4) % (size_t)C));
        v.x += b.x;
        v.y += b.y;
        v.z += b.z;
        v.w += b.w;
    }
    if (add_src) {
        const float4 u = reinterpret_cast<const float4*>(add_src)[i];
        v.x += u.x;
        v.y += u.y;
        v.z += u.z;
        v.w += u.w;
    }
    if (mask_src) {
        const float4 m = reinterpret_cast<const float4*>(mask_src)[i];
        v.x = m.x > 0.f ? v.x : 0.f;
        v.y = m.y > 0.f ? v.y : 0.f;
        v.z = m.z > 0.f ? v.z : 0.f;
        v.w = m.w > 0.f ? v.w : 0.f;
    }
    if (relu) {
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
    }
    reinterpret_cast<float4*>(y)[i] = v;
}

#define FS_TRY_(x)           \
    do {                     \
        int rc_ = (x);       \
        if (rc_) return rc_; \
    } while (0)

// route_src (ConvArgs) is served by the store path of the direct kernel that also takes the consumer mask: plain stores, the
// sum complete in one workgroup (the window partners are read from mask_src in global memory: no tile alignment needed)
bool conv_route_ok(const ConvArgs& a) {
    const ConvPlan& p = a.p;
    return a.mask_src && !a.add_src && !a.shuffle && !a.stats && p.ksplit <= 1 && p.variant < 5;
}

int conv_launch(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    if (env_int("FS_CONV_DEBUG", 0))  // tuning aid: one line per launch with the chosen plan
        fprintf(stderr, "conv N%d %dx%dx%d -> %dx%dx%d k%dx%d s%d src%d: variant %d tile %dx%d (patch %dx%d) CC %d ksplit %d rem %d/%d lds %d wgs %d\n",
                a.N, a.H, a.W, a.Cin, a.Ho, a.Wo, a.Cout, a.KH, a.KW, a.stride, a.src_mode, a.p.variant, a.p.TH, a.p.TW, a.p.PH,
                a.p.PW, a.p.CC, a.p.ksplit, a.p.rem_full, a.p.rem_ks, a.p.lds_bytes, a.N * a.p.tiles_y * a.p.tiles_x * cdiv(a.Cout, a.p.BN));
    if (a.dil_x < 1) a.dil_x = 1;
    const ConvPlan& p = a.p;
    if (p.CC <= 0 || (!p.flat && (a.Cin % 4 || a.Cin % p.CC))) return -1;
    if (p.lds_bytes > 160 * 1024) return -2;
    if (a.route_src && !conv_route_ok(a)) return -8;
    dim3 grid((unsigned)(a.N * p.tiles_y * p.tiles_x), (unsigned)cdiv(a.Cout, p.BN), (unsigned)(p.ksplit > 1 ? p.ksplit : 1));
    if (p.ksplit > 1) {  // raw partial sums to scratch; the epilogue runs in splitk_epilogue_kernel
        if (!a.split_ws || a.stats || a.shuffle || a.add_pad || p.flat) return -6;
        a.y = a.split_ws;
        a.bias = nullptr;
        a.out_relu = 0;
        a.add_src = nullptr;
        a.mask_src = nullptr;
    }
    Profiler* prof = Profiler::current();
    if (prof) {
        // algorithmic FLOPs: 2*M*K*N with the true extents; a zero-dilated dgrad only does 1/4 useful work
        double fl = 2.0 * a.N * a.Ho * a.Wo * (double)a.KH * a.KW * a.Cin * a.Cout;
        if (a.src_mode == SRC_DILATE2) fl *= 0.25;
        if (a.shuffle) fl *= 9.0 / 16.0;  // phase-collapsed resize-conv / stride-2 dgrad: 9 of the 16 tap-parity slots are non-zero
        // Winograd F(2x2,3x3): 16 products per 2x2 output tile instead of 36 -- the FLOPs actually executed
        if (p.variant == 5 || p.variant == 6 || p.variant == 8) fl = 2.0 * a.N * cdiv(a.Ho, 2) * cdiv(a.Wo, 2) * 16.0 * a.Cin * a.Cout;
        if (p.variant == 10 || p.variant == 11 || p.variant == 12) fl = 2.0 * a.N * cdiv(a.Ho, 4) * cdiv(a.Wo, 4) * 36.0 * a.Cin * a.Cout;   // F(4x4,3x3): 36 products per 4x4 outputs
        int fam = p.variant;   // conv_igemm_kernel<..> instances 0..4, wino_conv_kernel 5
        if (a.w_nstride) fam = PF_GRAM_BWD_IGEMM;
        else if (p.variant == 7) fam = PF_CSTREAM;
        else if (p.variant == 9) fam = PF_S16;
        else if (p.variant == 10) fam = PF_WINO4;
        else if (p.variant == 12) fam = PF_WINO6;
        else if (p.variant == 11) fam = a.prof_tag ? PF_WINO4T_TNET : PF_WINO4T_VGG;
        else if (p.variant == 8) fam = PF_WINO2H_TNET;
        else if (p.variant == 6) fam = a.prof_tag ? PF_WINO2_TNET : PF_WINO2_VGG;
        prof->begin(fam, fl, s);
    }
#define FS_LAUNCH(MT_, WM_, WN_, FL_)                                                                              \
    do {                                                                                                           \
        static BigLds lds_attr; /* > 64 KiB of dynamic LDS */                                                      \
        lds_attr.ensure(reinterpret_cast<const void*>(conv_igemm_kernel<MT_, WM_, WN_, FL_>));                       \
        hipLaunchKernelGGL((conv_igemm_kernel<MT_, WM_, WN_, FL_>), grid, dim3(256), (size_t)p.lds_bytes, s, a);   \
    } while (0)
    const ConvFamily* special = nullptr;
    {
        int nf = 0;
        const ConvFamily* fam = conv_families(&nf);
        for (int i = 0; i < nf; ++i)
            if (fam[i].variant == p.variant) special = &fam[i];
    }
    if (special) {
        if (!special->eligible(a_in)) return -7;
        FS_TRY_(special->launch(a, s));
    } else if (p.flat) {
        if (p.variant == 0)
            FS_LAUNCH(32, 2, 2, true);
        else if (p.variant == 2)
            FS_LAUNCH(16, 4, 1, true);
        else
            return -4;
    } else if (p.variant == 0) {
        FS_LAUNCH(32, 2, 2, false);
    } else if (p.variant == 1) {
        FS_LAUNCH(32, 2, 1, false);
    } else if (p.variant == 2) {
        FS_LAUNCH(16, 4, 1, false);
    } else if (p.variant == 3) {
        FS_LAUNCH(32, 1, 2, false);
    } else {
        FS_LAUNCH(32, 1, 1, false);
    }
#undef FS_LAUNCH
    if (p.ksplit > 1) {
        const size_t n4 = (size_t)a.N * a.Ho * a.Wo * a.Cout / 4;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a.split_ws, p.ksplit, n4,
                           a.Cout, a_in.bias, a_in.add_src, a_in.mask_src, a_in.out_relu, a_in.y);
    }
    if (prof) prof->end(s);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
