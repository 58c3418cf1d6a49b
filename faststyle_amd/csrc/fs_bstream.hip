// Streaming bf16 convolution for the mixed-precision INFERENCE path of the transform net (BASELINE config 5: 1080p, batch 8
// per GPU; reference im_transf_net.py:14-75) -- all sixteen conv launches: the 9x9 image layer, the two stride-2 convs, the ten
// residual convs, the two phase-collapsed resize-convs and the kw-folded 9x9 output layer.
//
// Round 2's kernels for these layers (fs_bf16.hip: one tile per workgroup, or a resident filter in LDS with one dependent
// matrix instruction per loop trip, ~1300 instructions and three barriers per 256-pixel tile) ran 4-8x above their HBM
// floor: at 16x the fp32 matrix rate these launches are about BYTES and INSTRUCTIONS PER TILE, not FLOPs.  This is the fp32
// streaming recipe of fs_cstream.hip in bf16:
//   * ONE persistent workgroup per CU (256 threads, one wave per SIMD) walks a strided list of 16x16-pixel tiles as a software
//     pipeline: loads of tile t+1 in flight (registers) | sweep of tile t out of LDS | barrier | commit of tile t+1 over the
//     patch (producer instance norm + ReLU applied, bf16 -> fp32 -> bf16) | epilogue of tile t | barrier | 16-byte stores;
//   * the whole filter lives in REGISTERS for the workgroup's lifetime as MFMA B fragments (v_mfma_f32_32x32x16_bf16: 8 bf16
//     = 4 registers per k-step and 32-channel block; 36..144 registers) -- no filter traffic through LDS;
//   * the sweep is fully unrolled: every A-fragment address is the lane's base + a compile-time offset (one ds_read_b128 per
//     matrix instruction), fragments of step j+D are read ahead of the matrix instructions of step j (sched_barrier-pinned);
//   * fp32 accumulation; the instance-norm partials {mean, M2, count} come from the fp32 accumulators before they are
//     rounded (merged per tile one pipeline step later: no barrier of their own); the tile is rounded to bf16, staged through
//     LDS as [pixel][channel] and leaves as 16-byte stores of whole pixel rows (2x2 pixel-shuffle for the resize-convs).
// Same rounding points as the kernels it replaces (oracle.tnet.create_net_bf16): bf16 inputs, weights and stored activations,
// fp32 everything else.
#include "fs_bf16.h"

#include <type_traits>

namespace fs {

namespace {
typedef __bf16 bs_bf16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned kOOB = 0x80000000u;
constexpr int kBTH = 16, kBTW = 16;   // output tile

__device__ __forceinline__ unsigned bs_pack2(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 v;
    v.x = (__bf16)lo;
    v.y = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);   // v_cvt_pk_bf16_f32: round to nearest even
#else
    auto r = [](float f) {
        unsigned u = __builtin_bit_cast(unsigned, f);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return u >> 16;
    };
    return r(lo) | (r(hi) << 16);
#endif
}
// element k of a 96-bit buffer load (a 3-vector of dwords on the GPU; a plain struct in the CPU emulator of the tests)
template <class T>
__device__ __forceinline__ unsigned bs_u3(const T& t, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    return t[k];
#else
    unsigned w[4] = {0u, 0u, 0u, 0u};
    __builtin_memcpy(w, &t, sizeof(T) < 16 ? sizeof(T) : 16);
    return w[k];
#endif
}
__device__ __forceinline__ unsigned short bs_f2bf(float f) { return (unsigned short)(bs_pack2(f, 0.f) & 0xFFFFu); }
}  // namespace

// BN output channels per workgroup (32: four waves over the pixel blocks; 64: 2 x 2 waves); CIN input channels; KH x KW taps,
// horizontal tap spacing DILX; STRIDE 1 or 2.  CIN == 3 is the IMAGE LAYER (9x9, fp32 RGB input, REFLECT-40 fused): pixels
// are staged as 4-channel bf16 (8 bytes), K runs over (12 taps of a kernel row) x 4 = 3 k-steps per row, 27 in all -- the
// packed filter is [9][cout_pad][48] with zeros for the 3 padding taps and the 4th channel (fs_bf16.hip PK_C4).
#ifdef FS_BSTREAM_TRACE
// debug build only (tools/bstream_trace.py): per-workgroup phase cycle counts of the last launch
__device__ long long g_bstream_trace[512 * 10];
extern "C" int fs_debug_conv_trace(long long* out, int n_wg) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bstream_trace), sizeof(long long) * 10 * (size_t)n_wg, 0, hipMemcpyDeviceToHost);
}
extern "C" int fs_debug_conv_trace_reset() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_bstream_trace)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(long long) * 10 * 512);
}
#define FS_BS_NOW() ((long long)__builtin_readcyclecounter())
#define FS_BS_T(var) const long long var = FS_BS_NOW()
#define FS_BS_ACC(dst, t1, t0) dst += (t1) - (t0)
#else
#define FS_BS_T(var)
#define FS_BS_ACC(dst, t1, t0)
#endif

template <int BN, int CIN, int KH, int KW, int STRIDE, int DILX>
__global__ __launch_bounds__(256, (BN == 32 ? 2 : 1)) void conv_bstream_kernel(ConvBArgs a) {
    HIP_DYNAMIC_SHARED(float, smem_f)
#ifdef FS_BSTREAM_TRACE
    const long long tr_t0 = FS_BS_NOW();
    long long tr_issue = 0, tr_sweep = 0, tr_barA = 0, tr_commit = 0, tr_epw = 0, tr_barB = 0, tr_store = 0;
#endif
    const ConvBPlan& p = a.p;
    constexpr bool C4 = CIN == 3;
    constexpr int WN = BN / 32, WMW = 4 / WN, WM = 8 / WMW;
    constexpr int KC = C4 ? 3 : CIN / 16, G = C4 ? KH : KH * KW, KSTEPS = G * KC;
    constexpr int PP = C4 ? 4 : CIN + 8;                       // LDS elements per patch pixel
    constexpr int WROW = C4 ? 48 : CIN;                        // packed filter elements per (tap group, output channel)
    constexpr int PH = (kBTH - 1) * STRIDE + KH, PW = C4 ? (kBTW - 1) + 12 : (kBTW - 1) * STRIDE + (KW - 1) * DILX + 1, NPX = PH * PW;
    constexpr int G8 = C4 ? 1 : CIN / 8, G8SH = G8 == 1 ? 0 : (G8 == 2 ? 1 : (G8 == 4 ? 2 : 3));
    constexpr int NGR = NPX * G8, SX = (NGR + 255) / 256;
    constexpr int PATCH_E = (NPX * PP + 8 + 7) & ~7;           // + 8 elements of sink behind the patch
    constexpr int REDF = WMW * 3 * BN;
    constexpr int OG = BN / 8, OGSH = OG == 4 ? 2 : 3;         // 16-byte output granules per staged pixel
    constexpr int SO = OG;                                     // ... per thread and tile (256 pixels x OG / 256 threads)
    unsigned short* const patch = reinterpret_cast<unsigned short*>(smem_f);
    float* const red = reinterpret_cast<float*>(patch + PATCH_E);                  // [2][REDF]
    unsigned short* const stage = reinterpret_cast<unsigned short*>(red + 2 * REDF);  // [256][BN]
    // (every lambda below is force-inlined: a lambda the inliner declines -- it happened with two call sites per lambda --
    // becomes a real call whose captured accumulators live in scratch memory: measured 28x slower)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, kq = lane >> 5;
    const int mw = wave % WMW, nbw = wave / WMW;
    const int co0 = blockIdx.y * BN;
    auto fdiv = [](int x, float inv_d) __attribute__((always_inline)) { return (int)(((float)x + 0.5f) * inv_d); };
    auto uniform_ptr = [](const void* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    };

    // ---- the filter, once, into registers: packed bf16 [tap group][cout_pad][WROW]; lane (lm, kq) of channel block nbw
    // holds, for k-step j = (group j / KC, 16-element slice j % KC), elements (j % KC)*16 + kq*8 .. +7 of output channel
    // co0 + nbw*32 + lm
    bs_bf16x8 breg[KSTEPS];
    {
        const unsigned short* wl = a.w + ((size_t)(co0 + nbw * 32 + lm)) * WROW + kq * 8;
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j)
            breg[j] = __builtin_bit_cast(bs_bf16x8, *reinterpret_cast<const uint4*>(wl + (size_t)(j / KC) * p.cout_pad * WROW + (j % KC) * 16));
    }

    // ---- this lane's A-fragment bases: pixel t = (mw*WM + m)*32 + lm of the tile (row t >> 4, column t & 15)
    int laneA[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int t = (mw * WM + m) * 32 + lm;
        laneA[m] = (((t >> 4) * STRIDE) * PW + (t & 15) * STRIDE) * PP + kq * 8;
    }

    // ---- staging descriptors (tile-invariant): element e = tid + i*256 is granule g8 (8 channels; the whole pixel for the
    // image layer) of patch pixel e / G8
    const int g8 = tid & (G8 - 1);
    int pq[SX], pdst[SX];
    unsigned poffb[SX];   // byte offset of the granule from the patch's first source pixel (interior tiles: no per-granule arithmetic)
#pragma unroll
    for (int i = 0; i < SX; ++i) {
        const int e = tid + i * 256;
        pq[i] = -1;
        pdst[i] = NPX * PP;   // sink
        poffb[i] = kOOB;
        if (e < NGR) {
            const int pix = e >> G8SH;
            const int py = fdiv(pix, 1.0f / (float)PW), px = pix - py * PW;
            pq[i] = (py << 8) | px;
            pdst[i] = pix * PP + g8 * 8;
            poffb[i] = C4 ? (unsigned)((py * a.W + px) * 3) * 4u : (unsigned)((py * a.W + px) * CIN + g8 * 8) * 2u;
        }
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * CIN) * (C4 ? 4u : 2u));
    const bool has_ab = !C4 && a.in_a != nullptr;
    const bool in_relu = a.in_relu != 0;
    // zero padding must stay zero through the on-load affine; a VALID conv (the residual convs) never reads padding that
    // reaches a stored output, so its commit skips the masking
    const bool need_mask = a.pad_t != 0 || a.pad_l != 0 || (a.Ho - 1) * STRIDE + KH > a.H || (a.Wo - 1) * STRIDE + (KW - 1) * DILX + 1 > a.W;

    // ---- items: tile lin = blockIdx.x + it * gridDim.x over (sample, tile row, tile column)
    const int tiles = p.tiles_y * p.tiles_x;
    const int total = a.N * tiles;
    const int GX = (int)gridDim.x;
    const int my_items = ((int)blockIdx.x < total) ? (total - 1 - (int)blockIdx.x) / GX + 1 : 0;
    const float inv_tiles = 1.0f / (float)tiles, inv_tx = 1.0f / (float)p.tiles_x;
    struct Item {
        int n, ty0, tx0, lin;
    };
    auto decode = [&](int it) __attribute__((always_inline)) {
        Item r;
        r.lin = (int)blockIdx.x + it * GX;
        r.n = fdiv(r.lin, inv_tiles);
        const int tr = r.lin - r.n * tiles;
        const int tyi = fdiv(tr, inv_tx);
        r.ty0 = tyi * kBTH;
        r.tx0 = (tr - tyi * p.tiles_x) * kBTW;
        r.lin = __builtin_amdgcn_readfirstlane(r.lin);
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.ty0 = __builtin_amdgcn_readfirstlane(r.ty0);
        r.tx0 = __builtin_amdgcn_readfirstlane(r.tx0);
        return r;
    };
    // ONE set of staging registers: the loads of tile t+1 travel during the sweep of tile t.  (A second set -- loads of tile
    // t+2 in flight as well, tile loop unrolled by two -- was built and measured: no gain, the phase trace of
    // tools/bstream_trace.py shows these kernels bound by vector-ALU instruction issue, not by load latency; and the extra
    // 44-68 registers made the 64-channel instances spill.)
    constexpr int NSET = 1;
    uint4 pvs[NSET][SX];
    unsigned poks[NSET] = {0u};   // bit i: granule i came from inside the image; bit 31: the whole patch did
    f32x2 vas[NSET][4], vbs[NSET][4];
#pragma unroll
    for (int q = 0; q < NSET; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            vas[q][k].x = vas[q][k].y = 1.f;
            vbs[q][k].x = vbs[q][k].y = 0.f;
        }
    auto refl = [&](int v, int n_src, int& s) __attribute__((always_inline)) {   // tf.pad REFLECT by a.refl (im_transf_net.py:78-88); false: zero padding beyond it
        if (v < 0 || v >= n_src + 2 * a.refl) return false;
        s = v - a.refl;
        if (s < 0) s = -s;
        if (s >= n_src) s = 2 * (n_src - 1) - s;
        return true;
    };
    // poks bit 31: the tile's patch lies wholly inside the source image (no padding, no reflection): loads and commit
    // take the predicate-free path
    auto issue = [&](const Item& I, auto SET) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        uint4(&pv)[SX] = pvs[S];
        unsigned& pok = poks[S];
        f32x2(&va)[4] = vas[S];
        f32x2(&vb)[4] = vbs[S];
        const int vy0 = I.ty0 * STRIDE - a.pad_t - (C4 ? a.refl : 0), vx0 = I.tx0 * STRIDE - a.pad_l - (C4 ? a.refl : 0);   // first SOURCE pixel of the patch
        const bool interior = vy0 >= 0 && vx0 >= 0 && vy0 + PH <= a.H && vx0 + PW <= a.W;
        const void* xn = C4 ? static_cast<const void*>(static_cast<const float*>(a.x) + (size_t)I.n * a.H * a.W * 3)
                            : static_cast<const void*>(static_cast<const unsigned short*>(a.x) + (size_t)I.n * a.H * a.W * CIN);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(xn), 0, x_bytes, 0x00020000);
        if (interior) {
            pok = 0xFFFFFFFFu;
            const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)((vy0 * a.W + vx0) * (C4 ? 12 : CIN * 2)));
#pragma unroll
            for (int i = 0; i < SX; ++i) {
                if constexpr (C4) {
                    const auto t = __builtin_amdgcn_raw_buffer_load_b96(xr, poffb[i], base, 0);
                    pv[i] = make_uint4(bs_u3(t, 0), bs_u3(t, 1), bs_u3(t, 2), 0u);
                } else {
                    pv[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, poffb[i], base, 0));
                }
            }
        } else {
            pok = 0;
            const int wy0 = I.ty0 * STRIDE - a.pad_t, wx0 = I.tx0 * STRIDE - a.pad_l;   // first VIRTUAL pixel (before reflection)
#pragma unroll
            for (int i = 0; i < SX; ++i) {
                if constexpr (C4) {
                    int sy = 0, sx = 0;
                    const bool ok = pq[i] >= 0 && refl(wy0 + (pq[i] >> 8), a.H, sy) && refl(wx0 + (pq[i] & 255), a.W, sx);
                    const auto t = __builtin_amdgcn_raw_buffer_load_b96(xr, ok ? (unsigned)((sy * a.W + sx) * 3) * 4u : kOOB, 0, 0);
                    pv[i] = make_uint4(bs_u3(t, 0), bs_u3(t, 1), bs_u3(t, 2), 0u);
                } else {
                    const int sy = wy0 + (pq[i] >> 8), sx = wx0 + (pq[i] & 255);
                    const bool ok = pq[i] >= 0 && (unsigned)sy < (unsigned)a.H && (unsigned)sx < (unsigned)a.W;
                    pok |= ok ? (1u << i) : 0u;
                    pv[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((sy * a.W + sx) * CIN + g8 * 8) * 2u : kOOB, 0, 0));
                }
            }
        }
        if constexpr (!C4) {
            if (has_ab) {
                const float* pa = a.in_a + (size_t)I.n * a.in_nstride + g8 * 8;
                const float* pb = a.in_b + (size_t)I.n * a.in_nstride + g8 * 8;
                const float4 a0 = *reinterpret_cast<const float4*>(pa), a1 = *reinterpret_cast<const float4*>(pa + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(pb), b1 = *reinterpret_cast<const float4*>(pb + 4);
                va[0].x = a0.x, va[0].y = a0.y, va[1].x = a0.z, va[1].y = a0.w, va[2].x = a1.x, va[2].y = a1.y, va[3].x = a1.z, va[3].y = a1.w;
                vb[0].x = b0.x, vb[0].y = b0.y, vb[1].x = b0.z, vb[1].y = b0.w, vb[2].x = b1.x, vb[2].y = b1.y, vb[3].x = b1.z, vb[3].y = b1.w;
            }
        }
    };
    auto relu1 = [](float x) __attribute__((always_inline)) {   // ONE v_max_f32 (fmaxf / fmed3 come with a canonicalising second one)
#if defined(__HIP_DEVICE_COMPILE__)
        float r;
        asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
        return r;
#else
        return x > 0.f ? x : 0.f;
#endif
    };
    auto commit_as = [&](auto SET, auto MASKED) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        constexpr bool masked = decltype(MASKED)::value;
        const uint4(&pv)[SX] = pvs[S];
        const unsigned pok = poks[S];
        const f32x2(&va)[4] = vas[S];
        const f32x2(&vb)[4] = vbs[S];
#pragma unroll
        for (int i = 0; i < SX; ++i) {
            if constexpr (C4) {   // fp32 RGB -> one 4-channel bf16 pixel (8 bytes)
                const uint2 v = make_uint2(bs_pack2(__uint_as_float(pv[i].x), __uint_as_float(pv[i].y)), bs_pack2(__uint_as_float(pv[i].z), 0.f));
                *reinterpret_cast<uint2*>(patch + pdst[i]) = v;
            } else {
                uint4 v = pv[i];
                if (has_ab) {   // producer instance norm (+ ReLU) folded into the load: bf16 -> fp32, packed fma, -> bf16
                    unsigned* w32 = reinterpret_cast<unsigned*>(&v);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        f32x2 x2, sh = vb[k];
                        x2.x = __uint_as_float(w32[k] << 16);
                        x2.y = __uint_as_float(w32[k] & 0xFFFF0000u);
                        if (masked && !((pok >> i) & 1u)) sh.x = sh.y = 0.f;   // padding arrives as 0 and must stay 0
                        f32x2 r2 = fs_pk_fma(x2, va[k], sh);
                        if (in_relu) {
                            r2.x = relu1(r2.x);
                            r2.y = relu1(r2.y);
                        }
                        w32[k] = bs_pack2(r2.x, r2.y);
                    }
                }
                *reinterpret_cast<uint4*>(patch + pdst[i]) = v;
            }
        }
    };
    auto commit = [&](auto SET) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        if (need_mask && !(poks[S] >> 31))   // (wave-uniform: a tile at the image border of a padded conv)
            commit_as(SET, std::true_type{});
        else
            commit_as(SET, std::false_type{});
    };

    f32x16 acc[WM];
    auto aoff = [&](int j) __attribute__((always_inline)) { return C4 ? ((j / KC) * PW) * PP + (j % KC) * 16 : (((j / KC) / KW) * PW + ((j / KC) % KW) * DILX) * PP + (j % KC) * 16; };
    auto afrag = [&](int m, int j) __attribute__((always_inline)) {
        const unsigned short* src = patch + laneA[m] + aoff(j);
        if constexpr (C4) {   // 8-byte aligned only: two 8-byte reads
            const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 4);
            return make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
            return *reinterpret_cast<const uint4*>(src);
        }
    };
    auto sweep = [&]() __attribute__((always_inline)) {
        constexpr int D = WM >= 4 ? 1 : 2;
        uint4 av[D + 1][WM];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int m = 0; m < WM; ++m)
                if (d < KSTEPS) av[d][m] = afrag(m, d);
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
            if (j + D < KSTEPS) {
#pragma unroll
                for (int m = 0; m < WM; ++m) av[(j + D) % (D + 1)][m] = afrag(m, j + D);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                f32x16 c = acc[m];
                if (j == 0) {   // the first k-step starts from zero: no accumulator clearing pass per tile
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = 0.f;
                }
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bs_bf16x8, av[j % (D + 1)][m]), breg[j], c, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- epilogue, first half (before barrier B): statistics partials of this wave and the rounded tile into LDS.
    // Accumulator register r of lane (lm, kq), block m: tile pixel t = (mw*WM + m)*32 + (r & 3) + 8 (r >> 2) + 4 kq, i.e. row
    // 2 (mw*WM + m) + (r >> 3), column 4 kq + (r & 3) + 8 ((r >> 2) & 1); channel nbw*32 + lm.
    const int pyb = mw * WM * 2, pxb = kq * 4;
    const int st_lane = ((mw * WM) * 32 + 4 * kq) * BN + nbw * 32 + lm;   // stage element of (m = 0, r = 0)
    auto epilogue_write = [&](const Item& I, float* rbuf) __attribute__((always_inline)) {
        const int th_valid = min(kBTH, a.Ho - I.ty0), tw_valid = min(kBTW, a.Wo - I.tx0);
        if (a.stats) {
            const float other = __shfl_xor(acc[0][0], 32);
            const float cs = kq ? other : acc[0][0];   // shift: the wave's own first pixel of the channel
            float s1 = 0.f, s2 = 0.f;
            if (th_valid == kBTH && tw_valid == kBTW) {   // interior tile: packed arithmetic, no per-pixel predicates
                f32x2 c2, p1, p2;
                c2.x = c2.y = cs;
                p1.x = p1.y = p2.x = p2.y = 0.f;
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 v;
                        v.x = acc[m][r];
                        v.y = acc[m][r + 1];
                        const f32x2 d = fs_pk_sub(v, c2);
                        p1 = fs_pk_add(p1, d);
                        p2 = fs_pk_fma(d, d, p2);
                    }
                s1 = p1.x + p1.y;
                s2 = p2.x + p2.y;
            } else {
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool ok = pyb + 2 * m + (r >> 3) < th_valid && pxb + (r & 3) + 8 * ((r >> 2) & 1) < tw_valid;
                        const float d = ok ? acc[m][r] - cs : 0.f;
                        s1 += d;
                        s2 = fmaf(d, d, s2);
                    }
            }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lane < 32) {
                rbuf[(mw * 3 + 0) * BN + nbw * 32 + lane] = s1;
                rbuf[(mw * 3 + 1) * BN + nbw * 32 + lane] = s2;
                rbuf[(mw * 3 + 2) * BN + nbw * 32 + lane] = cs;
            }
        }
        unsigned short* sl = stage + st_lane;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const unsigned pk = bs_pack2(acc[m][r], acc[m][r + 1]);   // pixels t, t + 1 of this lane's channel: one conversion
                const int t = m * 32 + (r & 3) + 8 * (r >> 2);
                sl[t * BN] = (unsigned short)(pk & 0xFFFFu);
                sl[(t + 1) * BN] = (unsigned short)(pk >> 16);
            }
    };
    // ---- second half (after barrier B): 16-byte stores of whole pixel rows, every thread SO granules
    const int Cr = a.shuffle ? a.Cout >> 2 : a.Cout;
    const float inv_cr = 1.0f / (float)Cr;
    int sto[SO];   // tile-invariant part of the output element offset of granule i (relative to the tile's first pixel)
#pragma unroll
    for (int i = 0; i < SO; ++i) {
        const int e = tid + i * 256;
        const int pix = e >> OGSH, g = e & (OG - 1);
        const int py = pix >> 4, px = pix & 15, co = co0 + g * 8;
        if (a.shuffle) {
            const int q = fdiv(co, inv_cr), cof = co - q * Cr;
            sto[i] = ((2 * py + (q >> 1)) * (2 * a.Wo) + 2 * px + (q & 1)) * Cr + cof;
        } else {
            sto[i] = (py * a.Wo + px) * a.Cout + co;
        }
    }
    auto epilogue_store = [&](const Item& I) __attribute__((always_inline)) {
        const int th_valid = min(kBTH, a.Ho - I.ty0), tw_valid = min(kBTW, a.Wo - I.tx0);
        unsigned short* yb = static_cast<unsigned short*>(a.y) + (size_t)I.n * a.Ho * a.Wo * a.Cout +   // (same element count shuffled or not)
                             (a.shuffle ? ((size_t)(2 * I.ty0) * (2 * a.Wo) + 2 * I.tx0) * Cr : ((size_t)I.ty0 * a.Wo + I.tx0) * a.Cout);
        if (th_valid == kBTH && tw_valid == kBTW && co0 + BN <= a.Cout) {   // whole tile, whole channel block: no predicates
#pragma unroll
            for (int i = 0; i < SO; ++i) {
                const int e = tid + i * 256;
                *reinterpret_cast<uint4*>(yb + sto[i]) = *reinterpret_cast<const uint4*>(stage + (e >> OGSH) * BN + (e & (OG - 1)) * 8);
            }
        } else {
#pragma unroll
            for (int i = 0; i < SO; ++i) {
                const int e = tid + i * 256;
                const int pix = e >> OGSH, g = e & (OG - 1);
                if ((pix >> 4) < th_valid && (pix & 15) < tw_valid && co0 + g * 8 < a.Cout)
                    *reinterpret_cast<uint4*>(yb + sto[i]) = *reinterpret_cast<const uint4*>(stage + pix * BN + g * 8);
            }
        }
    };
    // merge of the per-wave records of one tile -> {mean, M2, count} of the tile.  Every count is wave-uniform (rows x valid
    // columns), so every division is by a uniform value: the reciprocals are computed once per tile on uniform operands and
    // the per-channel work is a few multiply-adds (the threads that do this are one wave, and the other three wait for it at
    // barrier A: with per-record float divisions this was a third of an image-layer tile).
    auto finalize = [&](const Item& I, const float* rbuf) __attribute__((always_inline)) {
        if (co0 + tid >= a.Cout) return;
        const int th_valid = min(kBTH, a.Ho - I.ty0), tw_valid = min(kBTW, a.Wo - I.tx0);
        float cbw[WMW], icb[WMW], tot = 0.f;
#pragma unroll
        for (int w = 0; w < WMW; ++w) {
            const int rows = min(2 * WM, max(0, th_valid - 2 * WM * w));
            cbw[w] = (float)(rows * tw_valid);
            icb[w] = cbw[w] > 0.f ? 1.0f / cbw[w] : 0.f;
            tot += cbw[w];
        }
        const float itot = 1.0f / tot;
        float mw_[WMW], qw_[WMW], msum = 0.f;
#pragma unroll
        for (int w = 0; w < WMW; ++w) {
            const float S1 = rbuf[(w * 3 + 0) * BN + tid], S2 = rbuf[(w * 3 + 1) * BN + tid], sh = rbuf[(w * 3 + 2) * BN + tid];
            mw_[w] = sh + S1 * icb[w];
            qw_[w] = fmaxf(S2 - S1 * S1 * icb[w], 0.f);
            msum = fmaf(cbw[w], mw_[w], msum);
        }
        const float mean = msum * itot;
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < WMW; ++w) {
            const float d = mw_[w] - mean;
            m2 += qw_[w] + cbw[w] * d * d;
        }
        float* st = a.stats + ((size_t)I.lin * a.Cout + co0 + tid) * 3;
        st[0] = mean;
        st[1] = m2;
        st[2] = tot;
    };

    // ---- the pipeline: one patch stage in LDS, one tile of loads in flight (as fs_cstream.hip)
    if (my_items == 0) return;
    if (tid < 8) patch[NPX * PP + tid] = 0;   // the sink (only ever rewritten with the zeros of out-of-range loads)
    using S0 = std::integral_constant<int, 0>;
    Item cur = decode(0), prev = cur;
    issue(cur, S0{});
    commit(S0{});
    __syncthreads();
    for (int it = 0; it < my_items; ++it) {
        FS_BS_T(q0);
        if (it > 0 && a.stats && tid < BN) finalize(prev, red + ((it - 1) & 1) * REDF);
        Item nxt = cur;
        if (it + 1 < my_items) {
            nxt = decode(it + 1);
            issue(nxt, S0{});
        }
        FS_BS_T(q1);
        sweep();
        FS_BS_T(q2);
        FS_LDS_BARRIER();   // A: every wave is done reading the patch (and the previous tile's staged pixels have left)
        FS_BS_T(q3);
        if (it + 1 < my_items) commit(S0{});
        FS_BS_T(q4);
        epilogue_write(cur, red + (it & 1) * REDF);
        FS_BS_T(q5);
        FS_LDS_BARRIER();   // B: next patch, this tile's statistics records and staged pixels visible
        FS_BS_T(q6);
        epilogue_store(cur);
        FS_BS_T(q7);
        FS_BS_ACC(tr_issue, q1, q0);
        FS_BS_ACC(tr_sweep, q2, q1);
        FS_BS_ACC(tr_barA, q3, q2);
        FS_BS_ACC(tr_commit, q4, q3);
        FS_BS_ACC(tr_epw, q5, q4);
        FS_BS_ACC(tr_barB, q6, q5);
        FS_BS_ACC(tr_store, q7, q6);
        prev = cur;
        cur = nxt;
    }
    if (a.stats && tid < BN) finalize(prev, red + ((my_items - 1) & 1) * REDF);
#ifdef FS_BSTREAM_TRACE
    if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 512) {
        long long* t = g_bstream_trace + (size_t)blockIdx.x * 10;
        t[0] = tr_t0;
        t[1] = tr_issue;
        t[2] = tr_sweep;
        t[3] = tr_barA;
        t[4] = tr_commit;
        t[5] = tr_epw;
        t[6] = tr_barB;
        t[7] = tr_store;
        t[8] = FS_BS_NOW();
        t[9] = my_items;
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------ host
namespace {
struct BsInst {
    int BN, Cin, KH, KW, stride, dil;
};
// 1: initconv_1 (16 -> 32, 3x3/2)   2: initconv_2 (32 -> 64, 3x3/2)   3: the residual convs (64 -> 64, 3x3)
// 4: 64 -> 32 resize-conv (2x2 taps, 128 virtual channels: two channel halves per tile)   5: 32 -> 16 resize-conv (2x2, 64 virtual)
// 6: the kw-folded output layer (9 x 2 taps, spacing 5, 16 virtual channels of a 32-wide block)
// 7: the image layer (3 -> 16, 9x9, fp32 RGB in, REFLECT-40 fused; 12-tap kernel rows of 4-channel bf16 pixels)
// The three 32-wide instances (1, 6, 7) fit 256 registers (__launch_bounds__(256, 2)): with the default persistent grid of 512
// TWO of their workgroups share a CU, and one's commit / epilogue issues beside the other's matrix instructions (bf16 MFMA and
// vector ALU co-issue across waves, unlike fp32 MFMA): 1080p batch 8 1861 -> 2154 fps.  The 64-wide instances keep the filter of
// a 32-channel block (144 registers for the residual convs) beside 64 accumulator registers: one workgroup per CU (32-channel
// blocks for them were tried: 343-358 registers, 91-108 spilled when forced under 256).
const BsInst kBs[7] = {{32, 16, 3, 3, 2, 1}, {64, 32, 3, 3, 2, 1}, {64, 64, 3, 3, 1, 1}, {64, 64, 2, 2, 1, 1}, {64, 32, 2, 2, 1, 1}, {32, 16, 9, 2, 1, 5},
                       {32, 3, 9, 9, 1, 1}};
}  // namespace

int bstream_instance(const ConvBArgs& a) {
    if (!tune_int("FS_BSTREAM", 1) || a.y_f32) return 0;
    const bool image = a.Cin == 3;
    if (image ? (!a.x_f32 || (a.src_mode != SRC_REFLECT && a.src_mode != SRC_PLAIN) || a.Cout > 32) : (a.x_f32 || a.src_mode != SRC_PLAIN)) return 0;
    const int dil = a.dil_x > 0 ? a.dil_x : 1;
    const int bn = a.Cout > 32 ? 64 : 32;
    for (int i = 0; i < 7; ++i)
        if (a.Cin == kBs[i].Cin && a.KH == kBs[i].KH && a.KW == kBs[i].KW && a.stride == kBs[i].stride && dil == kBs[i].dil && bn == kBs[i].BN) {
            if (!((tune_int("FS_BSTREAM_MASK", 127) >> i) & 1)) return 0;
            if (a.Cout % 8 || (a.shuffle && ((a.Cout >> 2) % 8 || a.Cout % 4))) return 0;   // 16-byte output granules
            if (a.in_a && !a.in_b) return 0;
            return i + 1;
        }
    return 0;
}

void bstream_plan(const ConvBArgs& a, ConvBPlan* out) {
    ConvBPlan p{};
    const int inst = bstream_instance(a);
    const BsInst& I = kBs[inst - 1];
    p.bs = inst;
    p.WM = 8 / (4 / (I.BN / 32));
    p.BN = I.BN;
    p.cout_pad = (a.Cout + I.BN - 1) / I.BN * I.BN;
    p.c4 = a.Cin == 3;
    p.CC = p.c4 ? 4 : a.Cin;
    p.PP = p.c4 ? 4 : a.Cin + 8;
    p.TH = kBTH;
    p.TW = kBTW;
    p.tiles_y = cdiv(a.Ho, kBTH);
    p.tiles_x = cdiv(a.Wo, kBTW);
    p.PH = (kBTH - 1) * I.stride + I.KH;
    p.PW = p.c4 ? (kBTW - 1) + 12 : (kBTW - 1) * I.stride + (I.KW - 1) * I.dil + 1;
    const int patch_e = (p.PH * p.PW * p.PP + 8 + 7) & ~7;
    const int redf = (4 / (I.BN / 32)) * 3 * I.BN;
    p.lds_bytes = patch_e * 2 + 2 * redf * 4 + 256 * I.BN * 2;
    p.wst_off = 0;
    *out = p;
}

template <int BN, int CIN, int KH, int KW, int STRIDE, int DILX>
static void bs_launch(const ConvBArgs& a, dim3 grid, hipStream_t s) {
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(conv_bstream_kernel<BN, CIN, KH, KW, STRIDE, DILX>));
    hipLaunchKernelGGL((conv_bstream_kernel<BN, CIN, KH, KW, STRIDE, DILX>), grid, dim3(256), (size_t)a.p.lds_bytes, s, a);
}

int bstream_launch(const ConvBArgs& a_in, hipStream_t s) {
    ConvBArgs a = a_in;
    if (a.dil_x < 1) a.dil_x = 1;
    const ConvBPlan& p = a.p;
    if (p.lds_bytes > 160 * 1024) return -2;
    const long total = (long)a.N * p.tiles_y * p.tiles_x;
    const int wgs = p.BN == 32 ? tune_int("FS_BSTREAM_WGS", 512) : tune_int("FS_BSTREAM_WGS64", 256);
    const int ny = p.cout_pad / p.BN;
    long gx = wgs / ny;   // persistent workgroups over both grid dimensions: two per CU where 256 registers allow it, else two rounds of one
    if (gx < 1) gx = 1;
    if (gx > total) gx = total;
    const dim3 grid((unsigned)gx, (unsigned)ny);
    switch (p.bs) {
        case 1: bs_launch<32, 16, 3, 3, 2, 1>(a, grid, s); break;
        case 2: bs_launch<64, 32, 3, 3, 2, 1>(a, grid, s); break;
        case 3: bs_launch<64, 64, 3, 3, 1, 1>(a, grid, s); break;
        case 4: bs_launch<64, 64, 2, 2, 1, 1>(a, grid, s); break;
        case 5: bs_launch<64, 32, 2, 2, 1, 1>(a, grid, s); break;
        case 6: bs_launch<32, 16, 9, 2, 1, 5>(a, grid, s); break;
        case 7: bs_launch<32, 3, 9, 9, 1, 1>(a, grid, s); break;
        default: return -4;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
