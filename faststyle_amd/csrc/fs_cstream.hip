// Streaming convolution for the narrow, full-resolution layers of the transform net (reference im_transf_net.py:37-70:
// the 16- and 32-channel stride-2 / resize-conv units next to the image) and their input gradients.
//
// These launches hold few FLOPs per byte (K = taps x Cin <= 144, 32..64 output channels): in conv_igemm_kernel a workgroup
// lives for one tile -- load, wait, multiply, store -- two workgroups per CU in lockstep, and the chip alternates between
// an HBM burst and an MFMA burst (tools/conv_trace.py: the sweep is 1/3 of a workgroup's life).  Here ONE persistent
// workgroup per CU (256 threads, one wave per SIMD) walks a strided list of tiles as a software pipeline:
//     loads of tile t+1 in flight (registers)  |  MFMA sweep of tile t out of LDS  |  barrier  |
//     commit of tile t+1 over the patch (producer instance norm + ReLU applied)  |  epilogue of tile t (stores drain
//     during the next sweep; statistics records merged one step later: no barrier of their own)  |  barrier
// (one patch stage: loads of tile t+1 travel in registers during the sweep of tile t) with the whole filter (K x Cout <= 37 KB)
// resident in REGISTERS for the workgroup's lifetime.  v_mfma_f32_32x32x2_f32, 16x16-pixel tiles, a wave owns two 32-pixel
// blocks x NB 32-channel blocks.  Epilogue options are the ones these layers use: per-tile instance-norm
// partials {mean, M2, count} and the 2x2 pixel-shuffle store of the phase-collapsed resize-conv / stride-2 input gradient.
#include "fs_kernels.h"

#include <type_traits>
#include <utility>

namespace fs {

namespace {
constexpr unsigned kOOB = 0x80000000u;
constexpr int kTW = 16;     // output tile width; TH (8 or 16) rows: TH/2 32-pixel blocks
}  // namespace
#ifndef FS_CS_ABL
#define FS_CS_ABL 0   /* timing experiments (results wrong): 1 no sweep, 2 no commit, 4 no epilogue stores, 8 no patch loads, 16 no statistics */
#endif

// TH x 16 output pixels per tile; the four waves tile it as (4/WN waves over the 32-pixel blocks) x (WN waves over the channel
// blocks), a wave owns WM = (TH/2)/(4/WN) pixel blocks x NB channel blocks (Cout = WN*NB*32); CIN input channels; KS x KS taps;
// STRIDE 1 or 2.  The register budget of the resident filter decides the split: KS*KS*CIN/2 * NB registers per lane.
// X6 (round 6, FS_CSTREAM_SPLIT): the products on the bf16 matrix cores as SIX EXACT products of bf16 pieces (the arithmetic of fs_wino6.hip / conv_s16x_kernel:
// x = h + m + l exactly by truncation; hh, hm, mh, hl, lh, mm issued, fp32 accumulation, the five small products in an accumulator of their own).
// v_mfma_f32_32x32x16_bf16 takes 32 cycles for 16 k where eight v_mfma_f32_32x32x2_f32 take 512: six products cost 192.  What changes against the fp32 form:
//   * the filter in registers as three bf16 pieces (12 registers per 16 k and channel block: 108-216 of the 512 a lone wave owns);
//   * the LDS patch holds [piece 3][CIN] bf16 per pixel + one 16-byte slot (an odd number of slots per pixel: the 16 pixels of a tile row fall on 16
//     distinct slots of a ds_read_b128 lane group), row pitch 32 pixels (stride 1) / 40 (stride 2): the second tile row of a fragment lies a multiple of
//     256 bytes behind the first -- conflict-free; stride 2 stores a row's even and odd columns apart, so that a tap reads ONE parity at unit stride;
//   * one k-step = 16 channels of one tap: lane (pixel lm, half kq) reads channels 8 kq .. + 7, one ds_read_b128 per piece, address = lane base + immediate;
//   * the commit splits every staged value (4 + 1.5 vector instructions on top of the affine + ReLU) and writes three 8-byte pieces.
typedef __bf16 cs_bf16x8 __attribute__((ext_vector_type(8)));
__host__ __device__ __forceinline__ void cs_split(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    h = u & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, h);
    m = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, m);
    l = __builtin_bit_cast(unsigned, r2);   // (<= 8 significant bits: its low half is zero)
}
namespace {
// geometry of the split form's patch, shared with the host (LDS bytes)
template <int TH, int CIN, int KS, int STRIDE>
struct CsxGeo {
    static constexpr int PH = (TH - 1) * STRIDE + KS, PW = (kTW - 1) * STRIDE + KS;
    static constexpr int PXB = 3 * CIN * 2 + 16;             // bytes per patch pixel
    static constexpr int HALF = (PW + 1) / 2;                // stride 2: odd columns start here
    static constexpr int PWP = STRIDE == 2 ? 40 : 32;        // row pitch in pixels
    static constexpr int PATCH_B = (PH * PWP + 1) * PXB;     // + one pixel: the sink of elements a thread does not own
    static_assert(PW <= PWP && 2 * HALF <= PWP, "patch row pitch");
};
}  // namespace

template <int TH, int WN, int NB, int CIN, int KS, int STRIDE, bool X6 = false>
__global__ __launch_bounds__(256) void conv_stream_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    char* const lds = reinterpret_cast<char*>(smem);
    const ConvPlan& p = a.p;
    constexpr int WMW = 4 / WN, WM = (TH / 2) / WMW, BN = WN * NB * 32, S = CIN + 1, C4 = CIN / 4, G = KS * KS, KSTEPS = G * CIN / 2;
    constexpr int C4SH = C4 == 4 ? 2 : (C4 == 8 ? 3 : 4);
    constexpr int PH = (TH - 1) * STRIDE + KS, PW = (kTW - 1) * STRIDE + KS, NPX = PH * PW;
    constexpr int REDF = WMW * 3 * BN;                    // one statistics buffer: [waves over pixels][s1, s2, shift][BN]
    constexpr int SX = (NPX * C4 + 255) / 256;            // 16-byte patch loads per thread and tile
    using XG = CsxGeo<TH, CIN, KS, STRIDE>;
    constexpr int PXB = XG::PXB, PWP = XG::PWP, HALF = XG::HALF;
    constexpr int KST = G * CIN / 16, SPT = CIN / 16;     // X6: 16-k steps, steps per tap
    constexpr int PATCH_F = X6 ? (XG::PATCH_B + 15) / 16 * 4 : ((NPX * S + 4 + 3) & ~3);       // fp32: + four slack floats: the LDS sink of elements a thread does not own
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, kq = lane >> 5;
    const int mw = wave % WMW, nbw = wave / WMW;          // this wave's place among the pixel blocks / channel blocks
    float* const red = smem + PATCH_F;   // [2 buffers][REDF]
    auto fdiv = [](int x, float inv_d) __attribute__((always_inline)) { return (int)(((float)x + 0.5f) * inv_d); };   // exact for these magnitudes (x < 2^22)
    auto uniform_ptr = [](const float* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };

    // ---- the filter, once, into REGISTERS: lane (lm, kq) multiplies rows k = 2j + kq of [k = tap*Cin + ci][co] against
    // its columns nn*32 + lm -- KSTEPS x NB values that never change (no filter traffic through LDS at all)
    float breg[X6 ? 1 : KSTEPS][NB];
    cs_bf16x8 bx[X6 ? KST : 1][NB][3];   // X6: lane (lm, kq) holds rows k = 16 s + 8 kq .. + 7 of step s as three pieces
    if constexpr (!X6) {
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j)
#pragma unroll
            for (int nn = 0; nn < NB; ++nn) breg[j][nn] = a.w[(2 * j + kq) * BN + (nbw * NB + nn) * 32 + lm];
    } else {
#pragma unroll
        for (int sI = 0; sI < KST; ++sI)
#pragma unroll
            for (int nn = 0; nn < NB; ++nn) {
                unsigned hh[8], mm[8], ll[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) cs_split(a.w[(16 * sI + 8 * kq + e) * BN + (nbw * NB + nn) * 32 + lm], hh[e], mm[e], ll[e]);
                bx[sI][nn][0] = __builtin_bit_cast(cs_bf16x8, make_uint4((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3], (hh[4] >> 16) | hh[5], (hh[6] >> 16) | hh[7]));
                bx[sI][nn][1] = __builtin_bit_cast(cs_bf16x8, make_uint4((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3], (mm[4] >> 16) | mm[5], (mm[6] >> 16) | mm[7]));
                bx[sI][nn][2] = __builtin_bit_cast(cs_bf16x8, make_uint4((ll[0] >> 16) | (ll[1] & 0xffff0000u), (ll[2] >> 16) | (ll[3] & 0xffff0000u),
                                                                         (ll[4] >> 16) | (ll[5] & 0xffff0000u), (ll[6] >> 16) | (ll[7] & 0xffff0000u)));
            }
    }

    // ---- this lane's pixels: pixel t of the tile = (mw*WM + m)*32 + rr, rr = (r & 3) + 8 (r >> 2) + 4 kq for accumulator
    // register r; with 16 columns that is row py = 2 (mw*WM + m) + (r >> 3), column px = 4 kq + (r & 3) + 8 ((r >> 2) & 1)
    const int pyb = mw * WM * 2, pxb = kq * 4;
    int laneA[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int t = (mw * WM + m) * 32 + lm;            // A operand: lane lm feeds pixel t (all 32 rows of the block)
        laneA[m] = X6 ? (((t >> 4) * STRIDE) * PWP + (t & 15)) * PXB + kq * 16      // (bytes; stride 2: column (t & 15) of the tap's parity half)
                      : (((t >> 4) * STRIDE) * PW + (t & 15) * STRIDE) * S + kq;
    }

    // ---- staging descriptors (tile-invariant): element e = tid + i*256 is channel quad c4 of patch pixel e / C4
    const int c4 = tid & (C4 - 1);   // the same for every element of a thread (256 is a multiple of C4)
    int pq[SX], pdst[SX];
    unsigned poffb[SX];   // byte offset of the element from the patch's first pixel (interior tiles: no per-element arithmetic)
#pragma unroll
    for (int i = 0; i < SX; ++i) {
        const int e = tid + i * 256;
        pq[i] = -1;
        pdst[i] = X6 ? PH * PWP * PXB : NPX * S;   // slack
        poffb[i] = kOOB;
        if (e < NPX * C4) {
            const int pix = e >> C4SH;
            const int py = fdiv(pix, 1.0f / (float)PW), px = pix - py * PW;
            pq[i] = (py << 8) | px;
            pdst[i] = X6 ? (py * PWP + (STRIDE == 2 ? (px & 1) * HALF + (px >> 1) : px)) * PXB + c4 * 8 : pix * S + c4 * 4;
            poffb[i] = (unsigned)((py * a.W + px) * CIN + c4 * 4) * 4u;
        }
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * CIN) * 4u);
    const bool has_ab = a.in_a != nullptr;
    const bool in_relu = a.in_relu != 0;

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
        r.ty0 = tyi * TH;
        r.tx0 = (tr - tyi * p.tiles_x) * kTW;
        r.lin = __builtin_amdgcn_readfirstlane(r.lin);
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.ty0 = __builtin_amdgcn_readfirstlane(r.ty0);
        r.tx0 = __builtin_amdgcn_readfirstlane(r.tx0);
        return r;
    };
    float4 pv[SX];
    unsigned pok = 0;   // bit i: element i came from inside the image (padding must stay 0 through the on-load affine);
                        // bit 31: the WHOLE patch did (interior tile: predicate-free loads and commit)
    float4 va = make_float4(1.f, 1.f, 1.f, 1.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
    auto issue = [&](const Item& I) __attribute__((always_inline)) {   // global loads of the item's patch (zero padding: out-of-range offset -> zeros)
        const int vy0 = I.ty0 * STRIDE - a.pad_t, vx0 = I.tx0 * STRIDE - a.pad_l;
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * CIN);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, x_bytes, 0x00020000);
        if (vy0 >= 0 && vx0 >= 0 && vy0 + PH <= a.H && vx0 + PW <= a.W) {
            // interior tile (the vast majority): per-thread offsets are tile-invariant, the tile's origin rides in the scalar
            // offset operand -- no address arithmetic, no bounds checks (measured on the bf16 twin of this kernel,
            // tools/bstream_trace.py: the per-element form cost as much as the matrix instructions of a tile)
            pok = 0xFFFFFFFFu;
            const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)((vy0 * a.W + vx0) * CIN) * 4u);
#pragma unroll
            for (int i = 0; i < SX; ++i) pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, poffb[i], base, 0));
        } else {
            pok = 0;
#pragma unroll
            for (int i = 0; i < SX; ++i) {
                const int sy = vy0 + (pq[i] >> 8), sx = vx0 + (pq[i] & 255);
                const bool ok = pq[i] >= 0 && (unsigned)sy < (unsigned)a.H && (unsigned)sx < (unsigned)a.W;
                pok |= ok ? (1u << i) : 0u;
                pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((sy * a.W + sx) * CIN + c4 * 4) * 4u : kOOB, 0, 0));
            }
        }
        if (has_ab) {
            va = *reinterpret_cast<const float4*>(a.in_a + (size_t)I.n * a.in_nstride + c4 * 4);
            vb = *reinterpret_cast<const float4*>(a.in_b + (size_t)I.n * a.in_nstride + c4 * 4);
        }
    };
    auto relu1 = [](float x) __attribute__((always_inline)) {   // ONE v_max_f32 (fmaxf / fmed3 come with a canonicalising second instruction)
#if defined(__HIP_DEVICE_COMPILE__)
        float r;
        asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
        return r;
#else
        return x > 0.f ? x : 0.f;
#endif
    };
    auto commit_as = [&](auto MASKED) __attribute__((always_inline)) {
        constexpr bool masked = decltype(MASKED)::value;
#pragma unroll
        for (int i = 0; i < SX; ++i) {
            float4 v = pv[i];
            if (has_ab) {   // producer instance norm folded into the load; padding arrives as 0 and must stay 0
                const unsigned okm = (!masked || ((pok >> i) & 1u)) ? 0xFFFFFFFFu : 0u;
                v.x = fmaf(v.x, va.x, __uint_as_float(__float_as_uint(vb.x) & okm));
                v.y = fmaf(v.y, va.y, __uint_as_float(__float_as_uint(vb.y) & okm));
                v.z = fmaf(v.z, va.z, __uint_as_float(__float_as_uint(vb.z) & okm));
                v.w = fmaf(v.w, va.w, __uint_as_float(__float_as_uint(vb.w) & okm));
            }
            if (in_relu) {
                v.x = relu1(v.x);
                v.y = relu1(v.y);
                v.z = relu1(v.z);
                v.w = relu1(v.w);
            }
            if constexpr (X6) {
                const float vv[4] = {v.x, v.y, v.z, v.w};
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) cs_split(vv[c], h[c], m[c], l[c]);
                char* d = lds + pdst[i];
                *reinterpret_cast<uint2*>(d) = make_uint2((h[0] >> 16) | h[1], (h[2] >> 16) | h[3]);
                *reinterpret_cast<uint2*>(d + CIN * 2) = make_uint2((m[0] >> 16) | m[1], (m[2] >> 16) | m[3]);
                *reinterpret_cast<uint2*>(d + CIN * 4) = make_uint2((l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u));
            } else {
                float* d = smem + pdst[i];
                d[0] = v.x;
                d[1] = v.y;
                d[2] = v.z;
                d[3] = v.w;
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
        if (pok >> 31)   // (wave-uniform) interior tile: no masking
            commit_as(std::false_type{});
        else
            commit_as(std::true_type{});
    };

    f32x16 acc[WM][NB];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int nn = 0; nn < NB; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;
    };
    zero_acc();
    // the sweep is fully unrolled: every A address is the lane's base + a compile-time offset, B comes from registers.  Explicit
    // software pipeline pinned with sched_barrier: the A operands of step j + D are read before the matrix instructions of
    // step j (the compiler's own schedule waits for every read right before its MFMA)
    auto aoff = [&](int j) __attribute__((always_inline)) { return (((j / (CIN / 2)) / KS) * PW + ((j / (CIN / 2)) % KS)) * S + 2 * (j % (CIN / 2)); };
    // X6: the five small products of a block in an accumulator of their own where the register file has room (instance 1: 108 filter + 64 accumulator
    // registers; the others hold 192-216 filter registers and accumulate all six products of a step, smallest first, in one)
    constexpr bool SEP = X6 && (2 * WM * NB * 16 + KST * NB * 12 <= 250);
    f32x16 acs[SEP ? WM : 1][SEP ? NB : 1];
    auto zero_acs = [&]() __attribute__((always_inline)) {
        if constexpr (SEP) {
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int nn = 0; nn < NB; ++nn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acs[m][nn][r] = 0.f;
        }
    };
    zero_acs();
    // X6: units u = (step s, block m); the three operand pieces of unit u + 1 are read in front of the 6 NB matrix instructions of unit u
    auto xoff = [&](int sI, int pc) __attribute__((always_inline)) {   // compile-time byte offset of step sI, piece pc
        const int tap = sI / SPT, kh = tap / KS, kw = tap % KS;
        return (kh * PWP + (STRIDE == 2 ? (kw & 1) * HALF + (kw >> 1) : kw)) * PXB + pc * CIN * 2 + (sI % SPT) * 32;
    };
    auto sweep_x6 = [&]() __attribute__((always_inline)) {
        cs_bf16x8 av[2][3];
        auto rd = [&](int u, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) av[buf][pc] = __builtin_bit_cast(cs_bf16x8, *reinterpret_cast<const uint4*>(lds + laneA[u % WM] + xoff(u / WM, pc)));
        };
        rd(0, 0);
#pragma unroll
        for (int u = 0; u < KST * WM; ++u) {
            const int sI = u / WM, m = u % WM, b = u & 1;
            if (u + 1 < KST * WM) rd(u + 1, b ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nn = 0; nn < NB; ++nn) {   // smallest first: h l, l h, m m, h m, m h into the small accumulator, h h into the leading one
                f32x16& sm = SEP ? acs[SEP ? m : 0][SEP ? nn : 0] : acc[m][nn];
                sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[b][0], bx[sI][nn][2], sm, 0, 0, 0);
                sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[b][2], bx[sI][nn][0], sm, 0, 0, 0);
                sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[b][1], bx[sI][nn][1], sm, 0, 0, 0);
                sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[b][0], bx[sI][nn][1], sm, 0, 0, 0);
                sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[b][1], bx[sI][nn][0], sm, 0, 0, 0);
                acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[b][0], bx[sI][nn][0], acc[m][nn], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SEP) {
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int nn = 0; nn < NB; ++nn) {
                    acc[m][nn] += acs[m][nn];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acs[m][nn][r] = 0.f;
                }
        }
    };
    auto sweep = [&]() __attribute__((always_inline)) {
        if constexpr (X6) {
            sweep_x6();
        } else {
        constexpr int D = WM * NB >= 8 ? 1 : (WM * NB >= 4 ? 2 : 3);
        float av[D + 1][WM];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int m = 0; m < WM; ++m) av[d][m] = smem[laneA[m] + aoff(d)];
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
            if (j + D < KSTEPS) {
#pragma unroll
                for (int m = 0; m < WM; ++m) av[(j + D) % (D + 1)][m] = smem[laneA[m] + aoff(j + D)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nn = 0; nn < NB; ++nn)
#pragma unroll
                for (int m = 0; m < WM; ++m) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j % (D + 1)][m], breg[j][nn], acc[m][nn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
    };

    // ---- epilogue of one item
    const int Cr = a.shuffle ? BN >> 2 : BN;
    const int SH = a.shuf_H > 0 ? a.shuf_H : 2 * a.Ho, SW = a.shuf_W > 0 ? a.shuf_W : 2 * a.Wo;
    const unsigned y_bytes = __builtin_amdgcn_readfirstlane((unsigned)((a.shuffle ? SH * SW * Cr : a.Ho * a.Wo * BN)) * 4u);
    const int rowb = a.shuffle ? 2 * SW * Cr * 4 : a.Wo * BN * 4;      // bytes between two tile rows in the output
    const int colb = a.shuffle ? 2 * Cr * 4 : BN * 4;                  // ... two tile columns
    int chb[NB], qa[NB], qb[NB];                                       // per channel block: byte offset of the lane's channel
#pragma unroll
    for (int nn = 0; nn < NB; ++nn) {
        const int co = (nbw * NB + nn) * 32 + lm;
        const int q = a.shuffle ? co / Cr : 0;
        qa[nn] = q >> 1;
        qb[nn] = q & 1;
        chb[nn] = a.shuffle ? ((qa[nn] * SW + qb[nn]) * Cr + (co - q * Cr)) * 4 : co * 4;
    }
    auto epilogue = [&](const Item& I, float* rbuf) __attribute__((always_inline)) {
        const int th_valid = min(TH, a.Ho - I.ty0), tw_valid = min(kTW, a.Wo - I.tx0);
        if (a.stats && !(FS_CS_ABL & 16)) {
            // per-WAVE partial sums of (x - c), (x - c)^2 over the wave's 2*WM tile rows, c = the wave's own first pixel of the
            // channel; the records of the WMW waves that share a channel are merged when the next pipeline step starts
            // (finalize below): no barrier here
            float cs[NB], s1[NB], s2[NB];
#pragma unroll
            for (int nn = 0; nn < NB; ++nn) {
                const float other = __shfl_xor(acc[0][nn][0], 32);
                cs[nn] = kq ? other : acc[0][nn][0];
                s1[nn] = 0.f;
                s2[nn] = 0.f;
            }
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = pyb + 2 * m + (r >> 3) < th_valid && pxb + (r & 3) + 8 * ((r >> 2) & 1) < tw_valid;
#pragma unroll
                    for (int nn = 0; nn < NB; ++nn) {
                        const float d = ok ? acc[m][nn][r] - cs[nn] : 0.f;
                        s1[nn] += d;
                        s2[nn] = fmaf(d, d, s2[nn]);
                    }
                }
#pragma unroll
            for (int nn = 0; nn < NB; ++nn) {
                s1[nn] += __shfl_xor(s1[nn], 32);
                s2[nn] += __shfl_xor(s2[nn], 32);
            }
            if (lane < 32)
#pragma unroll
                for (int nn = 0; nn < NB; ++nn) {
                    rbuf[(mw * 3 + 0) * BN + (nbw * NB + nn) * 32 + lane] = s1[nn];
                    rbuf[(mw * 3 + 1) * BN + (nbw * NB + nn) * 32 + lane] = s2[nn];
                    rbuf[(mw * 3 + 2) * BN + (nbw * NB + nn) * 32 + lane] = cs[nn];
                }
        }
        // stores through a buffer resource.  Byte offset = lane part (column, channel: a register, or the out-of-range offset
        // for a pixel outside the image / the clipped shuffle extent) + scalar part (row) + compile-time part (r)
        float* yn = a.shuffle ? a.y + (size_t)I.n * SH * SW * Cr : a.y + (size_t)I.n * a.Ho * a.Wo * BN;
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yn)), 0, y_bytes, 0x00020000);
#pragma unroll
        for (int nn = 0; nn < NB; ++nn) {
            // rows / columns this channel block may store (shuffle: odd extents clip the last phase row / column)
            const int thv = a.shuffle ? min(th_valid, ((SH - qa[nn] + 1) >> 1) - I.ty0) : th_valid;
            const int twv = a.shuffle ? min(tw_valid, ((SW - qb[nn] + 1) >> 1) - I.tx0) : tw_valid;
            const int lane_off = (I.tx0 + pxb) * colb + chb[nn];
            float ad[WM][16];
            if (a.add_src) {
                // residual-gradient addend [N][Ho - 2 ap][Wo - 2 ap][BN], added in the interior: all loads of the block first
                const int ap = a.add_pad, aW = a.Wo - 2 * ap, aH = a.Ho - 2 * ap;
                const float* an = uniform_ptr(a.add_src + (size_t)I.n * aH * aW * BN);
                const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(an), 0, (unsigned)(aH * aW * BN) * 4u, 0x00020000);
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ay = I.ty0 + pyb + 2 * m + (r >> 3) - ap, ax = I.tx0 + pxb + (r & 3) + 8 * ((r >> 2) & 1) - ap;
                        const bool in = (unsigned)ay < (unsigned)aH && (unsigned)ax < (unsigned)aW;
                        ad[m][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ar, in ? (unsigned)((ay * aW + ax) * BN) * 4u + (unsigned)chb[nn] : kOOB, 0, 0));
                    }
            }
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pyl = pyb + 2 * m + (r >> 3), pxc = (r & 3) + 8 * ((r >> 2) & 1);
                    const bool ok = pyl < thv && pxb + pxc < twv;
                    const int row_off = (I.ty0 + pyl) * rowb;
                    float v = acc[m][nn][r];
                    if (a.add_src) v += ad[m][r];
                    if (!(FS_CS_ABL & 4) || v == 12345.678f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, ok ? (unsigned)(lane_off + row_off + pxc * colb) : kOOB, 0, 0);
                }
        }
        zero_acc();
    };
    // merge of the per-wave records of one tile (Chan's update, fixed order) -> {mean, M2, count} of the tile
    auto finalize = [&](const Item& I, const float* rbuf) __attribute__((always_inline)) {
        const int th_valid = min(TH, a.Ho - I.ty0), tw_valid = min(kTW, a.Wo - I.tx0);
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int w = 0; w < WMW; ++w) {
            const int rows = min(2 * WM, max(0, th_valid - 2 * WM * w));
            const float cb = (float)(rows * tw_valid);
            if (cb > 0.f) {
                const float S1 = rbuf[(w * 3 + 0) * BN + tid], S2 = rbuf[(w * 3 + 1) * BN + tid], sh = rbuf[(w * 3 + 2) * BN + tid];
                const float mb = sh + S1 / cb, qb2 = fmaxf(S2 - S1 * S1 / cb, 0.f);
                const float nn_ = cnt + cb, d = mb - mean, rr = cb / nn_;
                mean += d * rr;
                m2 += qb2 + d * d * cnt * rr;
                cnt = nn_;
            }
        }
        float* st = a.stats + ((size_t)I.lin * BN + tid) * 3;
        if (a.fin.counter) {   // read by the launch's last workgroup (fused finalize): coherent stores
            FS_COHERENT_STORE(st, mean);
            FS_COHERENT_STORE(st + 1, m2);
            FS_COHERENT_STORE(st + 2, cnt);
        } else {
            st[0] = mean;
            st[1] = m2;
            st[2] = cnt;
        }
    };

    // ---- the pipeline: ONE patch stage.  While tile t is multiplied the loads of tile t+1 are in flight (registers); after
    // the sweep (barrier A) they are committed over the patch, the epilogue of tile t follows (its stores drain during the
    // next sweep), barrier B, next tile.
    if (my_items == 0) return;
    Item cur = decode(0), prev = cur;
    issue(cur);
    commit();
    __syncthreads();
    for (int it = 0; it < my_items; ++it) {
        const bool more = it + 1 < my_items;
        if (it > 0 && a.stats && tid < BN) finalize(prev, red + ((it - 1) & 1) * REDF);
        Item nxt = cur;
        if (more) {
            nxt = decode(it + 1);
            if (!(FS_CS_ABL & 8)) issue(nxt);
        }
        if (!(FS_CS_ABL & 1)) sweep();
        FS_LDS_BARRIER();   // A: every wave is done reading the patch
        if (more && !(FS_CS_ABL & 2)) commit();
        epilogue(cur, red + (it & 1) * REDF);
        FS_LDS_BARRIER();   // B: next patch and this tile's statistics records (LDS) visible.  LDS-only on purpose: __syncthreads()
                            // waits for vmcnt(0), i.e. for every store of the epilogue to be acknowledged, before the next sweep
                            // may start -- the stores are meant to drain DURING it
        prev = cur;
        cur = nxt;
    }
    if (a.stats && tid < BN) finalize(prev, red + ((my_items - 1) & 1) * REDF);
    fs_fused_in_finalize(a.fin, a.stats, a.N, smem);   // (every workgroup has at least one tile: grid <= tiles)
}


// ---------------------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the FORWARD residual convs (64 -> 64, 3 x 3, stride 1, VALID; reference im_transf_net.py:201-215) as a DIRECT convolution in six exact
// bf16-piece products.  The fp32 Winograd F(4x4) kernel executes a quarter of the direct form's products at the fp32 matrix rate and pays an input
// transform per 8-channel step plus 8-12 k cycles of output transform per item, beside 8 steps of 3.7 k cycles (K = 64 is short); the direct form in
// split bf16 executes 6 products at 16 x the fp32 rate -- 6/16 x 4 = 1.5 x the Winograd kernel's matrix cycles on paper, and no transforms:
// 13.8 k cycles of matrix instructions per 8 x 16-pixel tile (the F(4x4) item: 36.9 k per 16 x 16 pixels + ~18 k around them).
//   * v_mfma_f32_16x16x32_bf16; wave w owns output channels 16 w .. + 15 and keeps ITS filter slice in registers as three pieces (18 k-steps x 12 = 216
//     registers; a 32-channel block per wave would need 432), every wave multiplies all TH pixel rows of the tile (a block = one tile row of 16 pixels);
//   * one k-step = half a tap: 32 channels, lane (pixel m16, kg) reads channels 32 (s & 1) + 8 kg .. + 7;
//   * LDS patch: eight planes of 8 channels, [pixel][piece 3][8 channels] bf16 = 48 bytes per pixel (an odd number of 16-byte slots), every plane a
//     multiple of 256 bytes: the lanes of a ds_read_b128 group differ in pixel column and in plane only -- conflict-free;
//   * the pipeline, the on-load affine + ReLU, the per-tile {mean, M2, count} records are conv_stream_kernel's (a wave owns its channels over the whole
//     tile: the record needs no merge across waves).  One workgroup per CU (~430 registers).
#ifdef FS_R64X_TRACE
// debug build only (tools/r64x_trace.py): per-workgroup phase cycle counts of the last launch: [wg][8] = issue, sweep, barrier A, commit, epilogue, barrier B, prologue, tiles
static __device__ long long g_r64x_trace[4096 * 8];
extern "C" int fs_debug_r64x_trace(long long* out, int n_wg) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_r64x_trace), sizeof(long long) * 8 * (size_t)n_wg, 0, hipMemcpyDeviceToHost);
}
#define FS_R64X_NOW() ((long long)__builtin_readcyclecounter())
#endif
typedef float cs_f32x4 __attribute__((ext_vector_type(4)));
template <int TH>
__global__ __launch_bounds__(256) void conv_r64x_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    char* const lds = reinterpret_cast<char*>(smem);
    const ConvPlan& p = a.p;
    constexpr int CIN = 64, COUT = 64, KS = 3;
    constexpr int PH = TH + KS - 1, PW = kTW + KS - 1, NPX = PH * PW;
    constexpr int GPB = ((NPX + 1) * 48 + 255) & ~255;   // one 8-channel plane (+ the sink pixel)
    constexpr int KST = KS * KS * CIN / 32;              // 32-k steps
    constexpr int SX = (NPX * 16 + 255) / 256;           // 16-byte patch loads per thread and tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, kg = lane >> 4;
    auto fdiv = [](int x, float inv_d) __attribute__((always_inline)) { return (int)(((float)x + 0.5f) * inv_d); };
    auto uniform_ptr = [](const float* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };

    // ---- the wave's filter slice, once, into registers: lane (channel 16 wave + m16, kg) holds k = 8 kg .. + 7 of step sI = (tap sI / 2, ci 32 (sI & 1) + 8 kg ..)
    cs_bf16x8 bx[KST][3];
#pragma unroll
    for (int sI = 0; sI < KST; ++sI) {
        unsigned hh[8], mm[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cs_split(a.w[((sI >> 1) * CIN + 32 * (sI & 1) + 8 * kg + e) * COUT + 16 * wave + m16], hh[e], mm[e], ll[e]);
        bx[sI][0] = __builtin_bit_cast(cs_bf16x8, make_uint4((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3], (hh[4] >> 16) | hh[5], (hh[6] >> 16) | hh[7]));
        bx[sI][1] = __builtin_bit_cast(cs_bf16x8, make_uint4((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3], (mm[4] >> 16) | mm[5], (mm[6] >> 16) | mm[7]));
        bx[sI][2] = __builtin_bit_cast(cs_bf16x8, make_uint4((ll[0] >> 16) | (ll[1] & 0xffff0000u), (ll[2] >> 16) | (ll[3] & 0xffff0000u), (ll[4] >> 16) | (ll[5] & 0xffff0000u),
                                                             (ll[6] >> 16) | (ll[7] & 0xffff0000u)));
    }
    const int laneA = kg * GPB + m16 * 48;
    auto xoff = [](int sI, int b, int pc) __attribute__((always_inline)) {   // compile-time byte offset of step sI, tile row b, piece pc
        const int tap = sI >> 1, kh = tap / KS, kw = tap % KS;
        return 4 * (sI & 1) * GPB + ((b + kh) * PW + kw) * 48 + pc * 16;
    };

    // ---- staging descriptors (tile-invariant): element e = tid + i*256 is channel quad c4 of patch pixel e / 16
    const int c4 = tid & 15;
    int pq[SX], pdst[SX];
    unsigned poffb[SX];
#pragma unroll
    for (int i = 0; i < SX; ++i) {
        const int e = tid + i * 256;
        pq[i] = -1;
        pdst[i] = (c4 >> 1) * GPB + NPX * 48 + (c4 & 1) * 8;   // the sink pixel
        poffb[i] = kOOB;
        if (e < NPX * 16) {
            const int pix = e >> 4;
            const int py = fdiv(pix, 1.0f / (float)PW), px = pix - py * PW;
            pq[i] = (py << 8) | px;
            pdst[i] = (c4 >> 1) * GPB + pix * 48 + (c4 & 1) * 8;
            poffb[i] = (unsigned)((py * a.W + px) * CIN + c4 * 4) * 4u;
        }
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * CIN) * 4u);
    const bool has_ab = a.in_a != nullptr;
    const bool in_relu = a.in_relu != 0;

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
        r.ty0 = tyi * TH;
        r.tx0 = (tr - tyi * p.tiles_x) * kTW;
        r.lin = __builtin_amdgcn_readfirstlane(r.lin);
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.ty0 = __builtin_amdgcn_readfirstlane(r.ty0);
        r.tx0 = __builtin_amdgcn_readfirstlane(r.tx0);
        return r;
    };
    float4 pv[SX];
    unsigned pok = 0;
    float4 va = make_float4(1.f, 1.f, 1.f, 1.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
    auto issue = [&](const Item& I) __attribute__((always_inline)) {
        const int vy0 = I.ty0 - a.pad_t, vx0 = I.tx0 - a.pad_l;
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * CIN);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, x_bytes, 0x00020000);
        if (vy0 >= 0 && vx0 >= 0 && vy0 + PH <= a.H && vx0 + PW <= a.W) {
            pok = 0xFFFFFFFFu;
            const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)((vy0 * a.W + vx0) * CIN) * 4u);
#pragma unroll
            for (int i = 0; i < SX; ++i) pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, poffb[i], base, 0));
        } else {
            pok = 0;
#pragma unroll
            for (int i = 0; i < SX; ++i) {
                const int sy = vy0 + (pq[i] >> 8), sx = vx0 + (pq[i] & 255);
                const bool ok = pq[i] >= 0 && (unsigned)sy < (unsigned)a.H && (unsigned)sx < (unsigned)a.W;
                pok |= ok ? (1u << i) : 0u;
                pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((sy * a.W + sx) * CIN + c4 * 4) * 4u : kOOB, 0, 0));
            }
        }
        if (has_ab) {
            va = *reinterpret_cast<const float4*>(a.in_a + (size_t)I.n * a.in_nstride + c4 * 4);
            vb = *reinterpret_cast<const float4*>(a.in_b + (size_t)I.n * a.in_nstride + c4 * 4);
        }
    };
    auto relu1 = [](float x) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        float r;
        asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
        return r;
#else
        return x > 0.f ? x : 0.f;
#endif
    };
    auto commit_as = [&](auto MASKED) __attribute__((always_inline)) {
        constexpr bool masked = decltype(MASKED)::value;
#pragma unroll
        for (int i = 0; i < SX; ++i) {
            float v[4] = {pv[i].x, pv[i].y, pv[i].z, pv[i].w};
            if (has_ab) {   // producer instance norm folded into the load; padding arrives as 0 and must stay 0
                const unsigned okm = (!masked || ((pok >> i) & 1u)) ? 0xFFFFFFFFu : 0u;
                v[0] = fmaf(v[0], va.x, __uint_as_float(__float_as_uint(vb.x) & okm));
                v[1] = fmaf(v[1], va.y, __uint_as_float(__float_as_uint(vb.y) & okm));
                v[2] = fmaf(v[2], va.z, __uint_as_float(__float_as_uint(vb.z) & okm));
                v[3] = fmaf(v[3], va.w, __uint_as_float(__float_as_uint(vb.w) & okm));
            }
            if (in_relu) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = relu1(v[c]);
            }
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) cs_split(v[c], h[c], m[c], l[c]);
            char* d = lds + pdst[i];
            *reinterpret_cast<uint2*>(d) = make_uint2((h[0] >> 16) | h[1], (h[2] >> 16) | h[3]);
            *reinterpret_cast<uint2*>(d + 16) = make_uint2((m[0] >> 16) | m[1], (m[2] >> 16) | m[3]);
            *reinterpret_cast<uint2*>(d + 32) = make_uint2((l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u));
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
        if (pok >> 31)
            commit_as(std::false_type{});
        else
            commit_as(std::true_type{});
    };

    cs_f32x4 acc[TH], acs[TH];   // the leading product / the five small ones, per tile row
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < TH; ++b) acc[b] = acs[b] = cs_f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    // units in PAIRS (two tile rows of one k-step): their matrix instructions alternate, so that an instruction never waits for the result of the one in front
    // of it (a 16-cycle v_mfma_f32_16x16x32_bf16 that accumulates into the register of its predecessor stalls for the predecessor's latency; with one wave per
    // SIMD nothing else fills the gap)
    auto sweep = [&]() __attribute__((always_inline)) {
        static_assert(!(TH & 1), "tile rows in pairs");
        cs_bf16x8 av[2][2][3];
        auto rd = [&](int pr, int buf) __attribute__((always_inline)) {   // pair pr = (step pr / (TH/2), rows 2 (pr % (TH/2)), + 1)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    av[buf][k][pc] = __builtin_bit_cast(cs_bf16x8, *reinterpret_cast<const uint4*>(lds + laneA + xoff(pr / (TH / 2), 2 * (pr % (TH / 2)) + k, pc)));
        };
        rd(0, 0);
#pragma unroll
        for (int pr = 0; pr < KST * TH / 2; ++pr) {
            const int sI = pr / (TH / 2), b0 = 2 * (pr % (TH / 2)), b1 = b0 + 1, q = pr & 1;
            if (pr + 1 < KST * TH / 2) rd(pr + 1, q ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            // smallest first: h l, l h, m m, h m, m h into the small accumulators, h h into the leading ones
            acs[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][2], av[q][0][0], acs[b0], 0, 0, 0);
            acs[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][2], av[q][1][0], acs[b1], 0, 0, 0);
            acs[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[q][0][2], acs[b0], 0, 0, 0);
            acs[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[q][1][2], acs[b1], 0, 0, 0);
            acs[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][1], av[q][0][1], acs[b0], 0, 0, 0);
            acs[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][1], av[q][1][1], acs[b1], 0, 0, 0);
            acs[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][1], av[q][0][0], acs[b0], 0, 0, 0);
            acs[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][1], av[q][1][0], acs[b1], 0, 0, 0);
            acs[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[q][0][1], acs[b0], 0, 0, 0);
            acs[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[q][1][1], acs[b1], 0, 0, 0);
            acc[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[q][0][0], acc[b0], 0, 0, 0);
            acc[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[q][1][0], acc[b1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < TH; ++b) acc[b] += acs[b];
    };

    // ---- epilogue of one item.  The filter is the matrix instruction's A operand (rows = channels), the pixels its B operand (columns): accumulator
    // register r of tile row b, lane (m16, kg) = pixel (row b, column m16), channel 16 wave + 4 kg + r -- four consecutive channels of one pixel, ONE 16-byte
    // store (a quarter of the store instructions of the pixel-major form: the epilogue's stores were issue-bound, and the next tile's loads queue behind them)
    const unsigned y_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * COUT) * 4u);
    auto epilogue = [&](const Item& I) __attribute__((always_inline)) {
        const int th_valid = min(TH, a.Ho - I.ty0), tw_valid = min(kTW, a.Wo - I.tx0);
        if (a.stats && !(FS_CS_ABL & 16)) {
            // the wave owns its 16 channels over the whole tile: sums of (x - c), (x - c)^2 with c = the tile's first pixel, reduced over the 16 pixel columns
            float cs[4], s1[4], s2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cs[r] = __shfl(acc[0][r], lane & 48);
                s1[r] = 0.f;
                s2[r] = 0.f;
            }
            const bool colok = m16 < tw_valid;
#pragma unroll
            for (int b = 0; b < TH; ++b) {
                const bool ok = colok && b < th_valid;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = ok ? acc[b][r] - cs[r] : 0.f;
                    s1[r] += d;
                    s2[r] = fmaf(d, d, s2[r]);
                }
            }
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s1[r] += __shfl_xor(s1[r], sh);
                    s2[r] += __shfl_xor(s2[r], sh);
                }
            if (m16 == 0) {
                const float cb = (float)(th_valid * tw_valid);
                float* st = a.stats + ((size_t)I.lin * COUT + 16 * wave + 4 * kg) * 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[3 * r + 0] = cs[r] + s1[r] / cb;
                    st[3 * r + 1] = fmaxf(s2[r] - s1[r] * s1[r] / cb, 0.f);
                    st[3 * r + 2] = cb;
                }
            }
        }
        float* yn = a.y + (size_t)I.n * a.Ho * a.Wo * COUT;
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yn)), 0, y_bytes, 0x00020000);
        const int lane_off = ((I.tx0 + m16) * COUT + 16 * wave + 4 * kg) * 4;
        const bool colok = m16 < tw_valid;
#pragma unroll
        for (int b = 0; b < TH; ++b) {
            const int row_off = (I.ty0 + b) * a.Wo * COUT * 4;
            const bool ok = colok && b < th_valid;
            if (!(FS_CS_ABL & 4) || acc[b][0] == 12345.678f)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fs_u32x4, acc[b]), yr, ok ? (unsigned)(lane_off + row_off) : kOOB, 0, 0);
        }
        zero_acc();
    };

    if (my_items == 0) return;
#ifdef FS_R64X_TRACE
    long long tr[7] = {0, 0, 0, 0, 0, 0, 0};
    long long t0 = FS_R64X_NOW(), t1;
#define FS_R64X_MARK(i) (t1 = FS_R64X_NOW(), tr[i] += t1 - t0, t0 = t1)
#else
#define FS_R64X_MARK(i) ((void)0)
#endif
    Item cur = decode(0);
    issue(cur);
    commit();
    FS_TOUCH_F4(va);
    FS_TOUCH_F4(vb);
    __syncthreads();
    FS_R64X_MARK(6);
    for (int it = 0; it < my_items; ++it) {
        const bool more = it + 1 < my_items;
        Item nxt = cur;
        if (more) {
            nxt = decode(it + 1);
            issue(nxt);
        }
        FS_R64X_MARK(0);
        if (!(FS_CS_ABL & 1)) sweep();
        FS_R64X_MARK(1);
        FS_LDS_BARRIER();   // A: every wave is done reading the patch
        FS_R64X_MARK(2);
        if (more && !(FS_CS_ABL & 2)) commit();
        FS_TOUCH_F4(va);
        FS_TOUCH_F4(vb);
        FS_R64X_MARK(3);
        epilogue(cur);
        FS_R64X_MARK(4);
        FS_LDS_BARRIER();   // B: next patch visible; the stores drain during the next sweep
        FS_R64X_MARK(5);
        cur = nxt;
    }
#ifdef FS_R64X_TRACE
    if (tid == 0 && blockIdx.x < 4096) {
        for (int i = 0; i < 7; ++i) g_r64x_trace[blockIdx.x * 8 + i] = tr[i];
        g_r64x_trace[blockIdx.x * 8 + 7] = my_items;
    }
#endif
}

// AFF: the producer's instance norm + ReLU on load (a.in_a / a.in_b / a.in_relu all set) -- a template parameter: the sweep that carries the commit is ONE basic block
// fn(integral_constant<BEG>, ..., integral_constant<BEG + N - 1>)
template <int BEG, class F, int... J>
__host__ __device__ __forceinline__ void cs_seq_impl(F& fn, std::integer_sequence<int, J...>) {
    fn(std::integral_constant<int, BEG + J>{}...);
}
template <int BEG, int N, class F>
__host__ __device__ __forceinline__ void cs_seq(F& fn) {
    cs_seq_impl<BEG>(fn, std::make_integer_sequence<int, N>{});
}
template <int TH, bool AFF>
__global__ __launch_bounds__(256) void conv_r64p_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    char* const lds = reinterpret_cast<char*>(smem);
    const ConvPlan& p = a.p;
    constexpr int CIN = 64, COUT = 64, KS = 3;
    constexpr int PH = TH + KS - 1, PW = kTW + KS - 1, NPX = PH * PW;
    constexpr int GPB = ((NPX + 1) * 48 + 255) & ~255;   // one 8-channel plane (+ the sink pixel)
    constexpr int KST = KS * KS * CIN / 32;              // 32-k steps
    constexpr int SX = (NPX * 16 + 255) / 256;           // 16-byte patch loads per thread and tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, kg = lane >> 4;
    auto fdiv = [](int x, float inv_d) __attribute__((always_inline)) { return (int)(((float)x + 0.5f) * inv_d); };
    auto uniform_ptr = [](const float* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };

    // ---- the wave's filter slice, once, into registers: lane (channel 16 wave + m16, kg) holds k = 8 kg .. + 7 of step sI = (tap sI / 2, ci 32 (sI & 1) + 8 kg ..)
    cs_bf16x8 bx[KST][3];
#pragma unroll
    for (int sI = 0; sI < KST; ++sI) {
        unsigned hh[8], mm[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cs_split(a.w[((sI >> 1) * CIN + 32 * (sI & 1) + 8 * kg + e) * COUT + 16 * wave + m16], hh[e], mm[e], ll[e]);
        bx[sI][0] = __builtin_bit_cast(cs_bf16x8, make_uint4((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3], (hh[4] >> 16) | hh[5], (hh[6] >> 16) | hh[7]));
        bx[sI][1] = __builtin_bit_cast(cs_bf16x8, make_uint4((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3], (mm[4] >> 16) | mm[5], (mm[6] >> 16) | mm[7]));
        bx[sI][2] = __builtin_bit_cast(cs_bf16x8, make_uint4((ll[0] >> 16) | (ll[1] & 0xffff0000u), (ll[2] >> 16) | (ll[3] & 0xffff0000u), (ll[4] >> 16) | (ll[5] & 0xffff0000u),
                                                             (ll[6] >> 16) | (ll[7] & 0xffff0000u)));
    }
    const int laneA = kg * GPB + m16 * 48;
    auto xoff = [](int sI, int b, int pc) __attribute__((always_inline)) {   // compile-time byte offset of step sI, tile row b, piece pc
        const int tap = sI >> 1, kh = tap / KS, kw = tap % KS;
        return 4 * (sI & 1) * GPB + ((b + kh) * PW + kw) * 48 + pc * 16;
    };

    // ---- staging descriptors (tile-invariant): element e = tid + i*256 is channel quad c4 of patch pixel e / 16
    const int c4 = tid & 15;
    int pq[SX], pdst[SX];
    unsigned poffb[SX];
#pragma unroll
    for (int i = 0; i < SX; ++i) {
        const int e = tid + i * 256;
        pq[i] = -1;
        pdst[i] = (c4 >> 1) * GPB + NPX * 48 + (c4 & 1) * 8;   // the sink pixel
        poffb[i] = kOOB;
        if (e < NPX * 16) {
            const int pix = e >> 4;
            const int py = fdiv(pix, 1.0f / (float)PW), px = pix - py * PW;
            pq[i] = (py << 8) | px;
            pdst[i] = (c4 >> 1) * GPB + pix * 48 + (c4 & 1) * 8;
            poffb[i] = (unsigned)((py * a.W + px) * CIN + c4 * 4) * 4u;
        }
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * CIN) * 4u);
    constexpr bool has_ab = AFF, in_relu = AFF;

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
        r.ty0 = tyi * TH;
        r.tx0 = (tr - tyi * p.tiles_x) * kTW;
        r.lin = __builtin_amdgcn_readfirstlane(r.lin);
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.ty0 = __builtin_amdgcn_readfirstlane(r.ty0);
        r.tx0 = __builtin_amdgcn_readfirstlane(r.tx0);
        return r;
    };
    // ---- the pipeline (VALID convs only: a patch never needs padding; columns / rows beyond the image read the next row / zeros beyond the sample and only
    // feed outputs that are neither stored nor counted).  TWO patch buffers: while tile t is multiplied out of buffer t & 1, the loads of tile t + 1 go out
    // (one per pair of the sweep's first twelve pairs), and their commit -- affine + ReLU, split, three 8-byte stores per 16-byte load -- is threaded between the
    // matrix instructions of the sweep's second half into the other buffer: a 16-cycle v_mfma_f32_16x16x32_bf16 leaves three issue slots to the SAME wave, and
    // this kernel has one wave per SIMD (the serial commit was 4.0 k cycles per tile beside 15.7 k of matrix instructions, the issue phase 1.9 k)
    constexpr int PATCHB = 8 * GPB;
    float4 pv[SX];
    float4 va = make_float4(1.f, 1.f, 1.f, 1.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
    struct Nxt {
        __amdgpu_buffer_rsrc_t xr;
        unsigned base;
        const float *pa, *pb;
    };
    auto next_of = [&](const Item& I, int live) __attribute__((always_inline)) {
        Nxt q;
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * CIN);
        q.xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, live ? x_bytes : 0u, 0x00020000);
        q.base = __builtin_amdgcn_readfirstlane((unsigned)((I.ty0 * a.W + I.tx0) * CIN) * 4u);
        q.pa = has_ab ? a.in_a + (size_t)I.n * a.in_nstride + c4 * 4 : a.x;
        q.pb = has_ab ? a.in_b + (size_t)I.n * a.in_nstride + c4 * 4 : a.x;
        return q;
    };
    // (an element past the sample's last byte -- the rows of a ragged bottom tile -- is dropped through the out-of-range offset, as everywhere in this library:
    // the hardware's range check would return the same zeros, the emulator insists on the marker)
    auto load_one = [&](const Nxt& q, int i) __attribute__((always_inline)) {
        const unsigned vo = poffb[i] != kOOB && poffb[i] + q.base + 16u <= x_bytes ? poffb[i] : kOOB;
        pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(q.xr, vo, q.base, 0));
    };
    auto relu1 = [](float x) __attribute__((always_inline)) {   // ONE v_max_i32 on the bit pattern (negative floats are negative integers); no inline assembly inside the unrolled sweep
        const int b = __builtin_bit_cast(int, x);
        return __builtin_bit_cast(float, b > 0 ? b : 0);
    };
    // the commit of element i in three steps of <= 12 vector instructions
    unsigned cw[3];
    auto commit_a = [&](int i) __attribute__((always_inline)) {
        float4& v = pv[i];
        if constexpr (has_ab) {
            v.x = fmaf(v.x, va.x, vb.x);
            v.y = fmaf(v.y, va.y, vb.y);
            v.z = fmaf(v.z, va.z, vb.z);
            v.w = fmaf(v.w, va.w, vb.w);
        }
        if constexpr (in_relu) {
            v.x = relu1(v.x);
            v.y = relu1(v.y);
            v.z = relu1(v.z);
            v.w = relu1(v.w);
        }
    };
    auto commit_b = [&](int i) __attribute__((always_inline)) {
        unsigned h0, m0, l0, h1, m1, l1;
        cs_split(pv[i].x, h0, m0, l0);
        cs_split(pv[i].y, h1, m1, l1);
        cw[0] = (h0 >> 16) | h1;
        cw[1] = (m0 >> 16) | m1;
        cw[2] = (l0 >> 16) | (l1 & 0xffff0000u);
    };
    auto commit_c = [&](int i, int wbuf) __attribute__((always_inline)) {
        unsigned h0, m0, l0, h1, m1, l1;
        cs_split(pv[i].z, h0, m0, l0);
        cs_split(pv[i].w, h1, m1, l1);
        char* d = lds + wbuf + pdst[i];
        *reinterpret_cast<uint2*>(d) = make_uint2(cw[0], (h0 >> 16) | h1);
        *reinterpret_cast<uint2*>(d + 16) = make_uint2(cw[1], (m0 >> 16) | m1);
        *reinterpret_cast<uint2*>(d + 32) = make_uint2(cw[2], (l0 >> 16) | (l1 & 0xffff0000u));
    };

    cs_f32x4 acc[TH], acs[TH];   // the leading product / the five small ones, per tile row
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < TH; ++b) acc[b] = acs[b] = cs_f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    // units in PAIRS (two tile rows of one k-step): their matrix instructions alternate, so that an instruction never waits for the result of the one in front
    // of it; per pair: its 12 matrix instructions, the 6 operand reads of the NEXT pair, and one piece of the next tile's staging
    auto sweep = [&](int rbuf, int wbuf, const Nxt& q) __attribute__((always_inline)) {
        static_assert(!(TH & 1), "tile rows in pairs");
        constexpr int NP = KST * TH / 2;
        static_assert(NP >= 36 + 3 * SX && SX <= 12, "the staging pieces fit the sweep");
        const int ra = laneA + rbuf;
        cs_bf16x8 av[2][2][3];
        auto rd = [&](int pr, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    av[buf][k][pc] = __builtin_bit_cast(cs_bf16x8, *reinterpret_cast<const uint4*>(lds + ra + xoff(pr / (TH / 2), 2 * (pr % (TH / 2)) + k, pc)));
        };
        rd(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // one pair: the next pair's operand reads, `extra` (a piece of the next tile's staging), the 12 matrix instructions, and the order the scheduler is asked for
        auto pair = [&](auto PR, auto extra) __attribute__((always_inline)) {
            constexpr int pr = decltype(PR)::value;
            constexpr int sI = pr / (TH / 2), b0 = 2 * (pr % (TH / 2)), b1 = b0 + 1, qq = pr & 1;
            if constexpr (pr + 1 < NP) rd(pr + 1, qq ^ 1);
            extra();
            acs[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][2], av[qq][0][0], acs[b0], 0, 0, 0);
            acs[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][2], av[qq][1][0], acs[b1], 0, 0, 0);
            acs[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[qq][0][2], acs[b0], 0, 0, 0);
            acs[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[qq][1][2], acs[b1], 0, 0, 0);
            acs[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][1], av[qq][0][1], acs[b0], 0, 0, 0);
            acs[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][1], av[qq][1][1], acs[b1], 0, 0, 0);
            acs[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][1], av[qq][0][0], acs[b0], 0, 0, 0);
            acs[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][1], av[qq][1][0], acs[b1], 0, 0, 0);
            acs[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[qq][0][1], acs[b0], 0, 0, 0);
            acs[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[qq][1][1], acs[b1], 0, 0, 0);
            acc[b0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[qq][0][0], acc[b0], 0, 0, 0);
            acc[b1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx[sI][0], av[qq][1][0], acc[b1], 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // one matrix instruction
                if (g < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // an operand read of the next pair
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);              // a vector-ALU instruction of the commit
                if (g == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // the pair's global load
                if (g >= 9) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // an LDS store of the commit
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        };
        // (compile-time pair indices through index sequences: a rolled loop over 72 pairs with every staging variant in its body exceeds the full-unroll budget)
        auto nothing = []() __attribute__((always_inline)) {};
        auto run_loads = [&](auto... I) __attribute__((always_inline)) { (pair(std::integral_constant<int, decltype(I)::value>{}, [&]() __attribute__((always_inline)) { load_one(q, decltype(I)::value); }), ...); };
        auto run_plain = [&](auto... I) __attribute__((always_inline)) { (pair(I, nothing), ...); };
        auto run_commit = [&](auto... I) __attribute__((always_inline)) {
            (pair(I, [&]() __attribute__((always_inline)) {
                 constexpr int kk = decltype(I)::value - (NP - 3 * SX), ci = kk / 3, sub = kk % 3;
                 if constexpr (sub == 0) commit_a(ci);
                 else if constexpr (sub == 1) commit_b(ci);
                 else commit_c(ci, wbuf);
             }),
             ...);
        };
        cs_seq<0, SX>(run_loads);
        pair(std::integral_constant<int, SX>{}, [&]() __attribute__((always_inline)) {
            if constexpr (has_ab) va = *reinterpret_cast<const float4*>(q.pa);
        });
        pair(std::integral_constant<int, SX + 1>{}, [&]() __attribute__((always_inline)) {
            if constexpr (has_ab) vb = *reinterpret_cast<const float4*>(q.pb);
        });
        cs_seq<SX + 2, NP - 3 * SX - SX - 2>(run_plain);
        cs_seq<NP - 3 * SX, 3 * SX>(run_commit);
#pragma unroll
        for (int b = 0; b < TH; ++b) acc[b] += acs[b];
    };

    // ---- epilogue of one item.  The filter is the matrix instruction's A operand (rows = channels), the pixels its B operand (columns): accumulator
    // register r of tile row b, lane (m16, kg) = pixel (row b, column m16), channel 16 wave + 4 kg + r -- four consecutive channels of one pixel, ONE 16-byte
    // store (a quarter of the store instructions of the pixel-major form: the epilogue's stores were issue-bound, and the next tile's loads queue behind them)
    const unsigned y_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * COUT) * 4u);
    auto epilogue = [&](const Item& I) __attribute__((always_inline)) {
        const int th_valid = min(TH, a.Ho - I.ty0), tw_valid = min(kTW, a.Wo - I.tx0);
        if (a.stats) {
            // the wave owns its 16 channels over the whole tile: sums of (x - c), (x - c)^2 with c = the tile's first pixel, reduced over the 16 pixel columns
            float cs[4], s1[4], s2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cs[r] = __shfl(acc[0][r], lane & 48);
                s1[r] = 0.f;
                s2[r] = 0.f;
            }
            const bool colok = m16 < tw_valid;
#pragma unroll
            for (int b = 0; b < TH; ++b) {
                const bool ok = colok && b < th_valid;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = ok ? acc[b][r] - cs[r] : 0.f;
                    s1[r] += d;
                    s2[r] = fmaf(d, d, s2[r]);
                }
            }
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s1[r] += __shfl_xor(s1[r], sh);
                    s2[r] += __shfl_xor(s2[r], sh);
                }
            if (m16 == 0) {
                const float cb = (float)(th_valid * tw_valid);
                float* st = a.stats + ((size_t)I.lin * COUT + 16 * wave + 4 * kg) * 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[3 * r + 0] = cs[r] + s1[r] / cb;
                    st[3 * r + 1] = fmaxf(s2[r] - s1[r] * s1[r] / cb, 0.f);
                    st[3 * r + 2] = cb;
                }
            }
        }
        float* yn = a.y + (size_t)I.n * a.Ho * a.Wo * COUT;
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yn)), 0, y_bytes, 0x00020000);
        const int lane_off = ((I.tx0 + m16) * COUT + 16 * wave + 4 * kg) * 4;
        const bool colok = m16 < tw_valid;
#pragma unroll
        for (int b = 0; b < TH; ++b) {
            const int row_off = (I.ty0 + b) * a.Wo * COUT * 4;
            const bool ok = colok && b < th_valid;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fs_u32x4, acc[b]), yr, ok ? (unsigned)(lane_off + row_off) : kOOB, 0, 0);
        }
        zero_acc();
    };

    if (my_items == 0) return;
    Item cur = decode(0);
    {   // the first tile: loads, wait, serial commit into buffer 0
        const Nxt q0 = next_of(cur, 1);
#pragma unroll
        for (int i = 0; i < SX; ++i) load_one(q0, i);
        if constexpr (has_ab) {
            va = *reinterpret_cast<const float4*>(q0.pa);
            vb = *reinterpret_cast<const float4*>(q0.pb);
        }
#pragma unroll
        for (int i = 0; i < SX; ++i) {
            commit_a(i);
            commit_b(i);
            commit_c(i, 0);
        }
    }
    __syncthreads();
    for (int it = 0; it < my_items; ++it) {
        const int live = it + 1 < my_items;
        const Item nxt = live ? decode(it + 1) : cur;
        const Nxt q = next_of(nxt, live);   // (no next tile: every load against an empty buffer -- zeros, no traffic -- into a patch nobody reads)
        const int rbuf = (it & 1) * PATCHB;
        sweep(rbuf, rbuf ^ PATCHB, q);
        FS_LDS_BARRIER();   // every wave is done reading this tile's patch and writing the next one's
        epilogue(cur);
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------------------------ host
namespace {
struct CsInst {
    int Cin, Cout, KS, stride, TH;
};
// 1: 16 -> 32, 3x3/2 (initconv_1; gradient of the 32 -> 16 resize-conv)      2: 32 -> 64, 2x2/1 (32 -> 16 resize-conv; gradient of initconv_1)
// 3: 32 -> 64, 3x3/2 (initconv_2; gradient of the 64 -> 32 resize-conv)      4: 64 -> 128, 2x2/1 (64 -> 32 resize-conv; gradient of initconv_2)
// 5: 64 -> 64, 3x3/1 (the residual convs and their input gradients when the grid is too small for the Winograd kernel: batch 4)
const CsInst kInst[5] = {{16, 32, 3, 2, 16}, {32, 64, 2, 1, 16}, {32, 64, 3, 2, 8}, {64, 128, 2, 1, 8}, {64, 64, 3, 1, 8}};
}  // namespace

static int cstream_instance(const ConvArgs& a) {
    for (int i = 0; i < 5; ++i)
        if (a.Cin == kInst[i].Cin && a.Cout == kInst[i].Cout && a.KH == kInst[i].KS && a.KW == kInst[i].KS && a.stride == kInst[i].stride)
            return i + 1;
    return 0;
}

// FS_CSTREAM_SPLIT (default 1): instances 1-4 on the bf16 matrix cores as six exact bf16-piece products (the X6 form of the kernel); 0: fp32 matrix instructions
static bool cstream_split_on() { return tune_int("FS_CSTREAM_SPLIT", 1) != 0; }

bool cstream_eligible(const ConvArgs& a) {
    const int inst = cstream_instance(a);
    if (!tune_int("FS_CSTREAM", 1) || !inst) return false;
    // bit i enables instance i+1.  Instance 5 (residual convs on small grids) is OFF by default: measured +1 % at batch 4
    // (20 launches 0.68 -> 0.63 ms) for 509 registers, and it would make the kernel choice of the residual convs depend on the
    // batch size (the data-parallel identity grads(batch) = sum grads(sample) then only holds to rounding-order noise)
    const bool r64x = inst == 5 && (a.res_x6 || tune_int("FS_CSTREAM_R64X_ALL", 0)) && cstream_split_on() && !a.add_src && !a.fin.counter && !a.shuffle;   // the forward residual convs, split-bf16 direct form (conv_r64x_kernel)
    if (!r64x && !((tune_int("FS_CSTREAM_MASK", 15) >> (inst - 1)) & 1)) return false;
    const bool plain = a.src_mode == SRC_PLAIN && a.dil_x <= 1 && !a.bias && !a.out_relu && !a.mask_src && !a.route_src &&
                       !a.pool_out && a.w_nstride == 0 && !a.w_wino && !a.w_wino2;
    if (!plain) return false;
    if (a.add_src && (a.shuffle || a.stats || a.add_pad < 0)) return false;   // (the residual-gradient addend: plain stores only)
    if (a.in_a && !a.in_b) return false;
    if (a.in_relu && !a.in_a) return false;   // (a ReLU on load only comes with its affine here)
    if (a.pad_t < 0 || a.pad_l < 0 || a.pad_t > 2 || a.pad_l > 2) return false;
    // measured at batch 4 (256x256: 100-500 tiles, at most two per workgroup): still ahead of the one-tile kernel -- the
    // resident filter and the cheap addresses count even without a second tile to prefetch; tiny launches stay with it
    const long tiles = (long)a.N * cdiv(a.Ho, kInst[inst - 1].TH) * cdiv(a.Wo, kTW);
    return tiles >= tune_int("FS_CSTREAM_MIN_TILES", 64);
}

void cstream_plan(const ConvArgs& a, ConvPlan* out) {
    ConvPlan p{};
    const int inst = cstream_instance(a);
    const int TH = inst ? kInst[inst - 1].TH : 16;
    p.variant = 7;
    p.BN = a.Cout;
    p.CC = a.Cin;
    p.TH = TH;
    p.TW = kTW;
    p.tiles_y = cdiv(a.Ho, TH);
    p.tiles_x = cdiv(a.Wo, kTW);
    p.PH = (TH - 1) * a.stride + a.KH;
    p.PW = (kTW - 1) * a.stride + a.KW;
    p.S = a.Cin + 1;
    p.ksplit = 1;
    int patch_floats = (p.PH * p.PW * p.S + 4 + 3) & ~3;
    if (inst == 5 && (a.res_x6 || tune_int("FS_CSTREAM_R64X_ALL", 0)) && cstream_split_on() && !a.add_src && !a.fin.counter && !a.shuffle) {   // conv_r64x_kernel: eight 8-channel planes of (pixels + 1) x 48 bytes, each rounded up to 256
        p.S = 48;
        patch_floats = 8 * ((((p.PH * p.PW + 1) * 48) + 255) & ~255) / 4;
        p.flat = 1;
        if (a.pad_t == 0 && a.pad_l == 0 && tune_int("FS_R64X_PIPE", 1)) {   // VALID: conv_r64p_kernel, the next tile's staging threaded into the sweep over a second patch buffer
            p.S = 49;
            patch_floats *= 2;
        }
    }
    if (inst >= 1 && inst <= 4 && cstream_split_on()) {   // the split-bf16 form (X6): [piece][Cin] bf16 + 16 bytes per pixel, row pitch 32 / 40 pixels, + the sink pixel
        p.S = 3 * a.Cin * 2 + 16;
        const int patch_b = (p.PH * (a.stride == 2 ? 40 : 32) + 1) * p.S;
        patch_floats = (patch_b + 15) / 16 * 4;
        p.flat = 1;   // (the launch reads this: the plan was made for the split form)
    }
    p.lds_bytes = 4 * (patch_floats + 2 * 12 * a.Cout);   // (statistics buffers: at most 4 wave records x 3 x Cout, twice)
    *out = p;
}

template <int TH, int WN, int NB, int CIN, int KS, int STRIDE, bool X6 = false>
static void cs_launch(const ConvArgs& a, unsigned grid, hipStream_t s) {
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(conv_stream_kernel<TH, WN, NB, CIN, KS, STRIDE, X6>));
    hipLaunchKernelGGL((conv_stream_kernel<TH, WN, NB, CIN, KS, STRIDE, X6>), dim3(grid), dim3(256), (size_t)a.p.lds_bytes, s, a);
}

int cstream_launch(const ConvArgs& a, hipStream_t s) {
    const ConvPlan& p = a.p;
    const long total = (long)a.N * p.tiles_y * p.tiles_x;
    const int wgs = tune_int("FS_CSTREAM_WGS", 256);
    const unsigned grid = (unsigned)(total < wgs ? total : wgs);
    if (p.flat) {   // the split-bf16 form
        switch (cstream_instance(a)) {
            case 1: cs_launch<16, 1, 1, 16, 3, 2, true>(a, grid, s); break;
            case 2: cs_launch<16, 1, 2, 32, 2, 1, true>(a, grid, s); break;
            case 3: cs_launch<8, 2, 1, 32, 3, 2, true>(a, grid, s); break;
            case 4: cs_launch<8, 4, 1, 64, 2, 1, true>(a, grid, s); break;
            case 5: {
                static BigLds lds_attr, lds_attr_p;
                if (a.p.S == 49 && (a.in_a != nullptr) == (a.in_relu != 0)) {   // the pipelined form (two patch buffers); on-load affine and ReLU come together
                    static BigLds lds_attr_q;
                    if (a.in_a) {
                        lds_attr_p.ensure(reinterpret_cast<const void*>(conv_r64p_kernel<8, true>));
                        hipLaunchKernelGGL((conv_r64p_kernel<8, true>), dim3(grid), dim3(256), (size_t)a.p.lds_bytes, s, a);
                    } else {
                        lds_attr_q.ensure(reinterpret_cast<const void*>(conv_r64p_kernel<8, false>));
                        hipLaunchKernelGGL((conv_r64p_kernel<8, false>), dim3(grid), dim3(256), (size_t)a.p.lds_bytes, s, a);
                    }
                } else {
                    lds_attr.ensure(reinterpret_cast<const void*>(conv_r64x_kernel<8>));
                    hipLaunchKernelGGL((conv_r64x_kernel<8>), dim3(grid), dim3(256), (size_t)a.p.lds_bytes, s, a);
                }
                break;
            }
            default: return -4;
        }
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    switch (cstream_instance(a)) {
        case 1: cs_launch<16, 1, 1, 16, 3, 2>(a, grid, s); break;   // 4 waves over the pixels; 72 filter registers
        case 2: cs_launch<16, 1, 2, 32, 2, 1>(a, grid, s); break;   // 4 waves over the pixels; 128
        case 3: cs_launch<8, 2, 1, 32, 3, 2>(a, grid, s); break;    // 2 x 2 waves; 144
        case 4: cs_launch<8, 4, 1, 64, 2, 1>(a, grid, s); break;    // 4 waves over the channels, 4 pixel blocks each; 128
        case 5: cs_launch<8, 2, 1, 64, 3, 1>(a, grid, s); break;    // 2 x 2 waves; 288 (what a lone wave's 512 registers allow)
        default: return -4;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
