// Streaming bf16 convolution for the mixed-precision INFERENCE path of the transform net (BASELINE config 5: 1080p, batch 8
// per GPU; reference im_transf_net.py:14-75) -- every layer behind the image layer: the two stride-2 convs, the ten residual
// convs, the two phase-collapsed resize-convs and the kw-folded 9x9 output layer.
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
__device__ __forceinline__ unsigned short bs_f2bf(float f) { return (unsigned short)(bs_pack2(f, 0.f) & 0xFFFFu); }
}  // namespace

// BN output channels per workgroup (32: four waves over the pixel blocks; 64: 2 x 2 waves); CIN input channels; KH x KW taps,
// horizontal tap spacing DILX; STRIDE 1 or 2.
template <int BN, int CIN, int KH, int KW, int STRIDE, int DILX>
__global__ __launch_bounds__(256) void conv_bstream_kernel(ConvBArgs a) {
    HIP_DYNAMIC_SHARED(float, smem_f)
    const ConvBPlan& p = a.p;
    constexpr int WN = BN / 32, WMW = 4 / WN, WM = 8 / WMW;
    constexpr int KC = CIN / 16, G = KH * KW, KSTEPS = G * KC;
    constexpr int PP = CIN + 8;
    constexpr int PH = (kBTH - 1) * STRIDE + KH, PW = (kBTW - 1) * STRIDE + (KW - 1) * DILX + 1, NPX = PH * PW;
    constexpr int G8 = CIN / 8, G8SH = G8 == 2 ? 1 : (G8 == 4 ? 2 : 3);
    constexpr int NGR = NPX * G8, SX = (NGR + 255) / 256;
    constexpr int PATCH_E = (NPX * PP + 8 + 7) & ~7;           // + 8 elements of sink behind the patch
    constexpr int REDF = WMW * 3 * BN;
    constexpr int OG = BN / 8, OGSH = OG == 4 ? 2 : 3;         // 16-byte output granules per staged pixel
    constexpr int SO = (256 * OG) / 256;                       // ... per thread and tile
    unsigned short* const patch = reinterpret_cast<unsigned short*>(smem_f);
    float* const red = reinterpret_cast<float*>(patch + PATCH_E);                  // [2][REDF]
    unsigned short* const stage = reinterpret_cast<unsigned short*>(red + 2 * REDF);  // [256][BN]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, kq = lane >> 5;
    const int mw = wave % WMW, nbw = wave / WMW;
    const int co0 = blockIdx.y * BN;
    auto fdiv = [](int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); };
    auto uniform_ptr = [](const void* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    };

    // ---- the filter, once, into registers: packed bf16 [tap][cout_pad][CIN]; lane (lm, kq) of channel block nbw holds, for
    // k-step j = (tap g, 16-channel group cs), the 8 input channels cs*16 + kq*8 .. +7 of output channel co0 + nbw*32 + lm
    bs_bf16x8 breg[KSTEPS];
    {
        const unsigned short* wl = a.w + ((size_t)(co0 + nbw * 32 + lm)) * CIN + kq * 8;
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j)
            breg[j] = __builtin_bit_cast(bs_bf16x8, *reinterpret_cast<const uint4*>(wl + (size_t)(j / KC) * p.cout_pad * CIN + (j % KC) * 16));
    }

    // ---- this lane's A-fragment bases: pixel t = (mw*WM + m)*32 + lm of the tile (row t >> 4, column t & 15)
    int laneA[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int t = (mw * WM + m) * 32 + lm;
        laneA[m] = (((t >> 4) * STRIDE) * PW + (t & 15) * STRIDE) * PP + kq * 8;
    }

    // ---- staging descriptors (tile-invariant): element e = tid + i*256 is granule (8 channels) g8 of patch pixel e / G8
    const int g8 = tid & (G8 - 1);
    int pq[SX], pdst[SX];
#pragma unroll
    for (int i = 0; i < SX; ++i) {
        const int e = tid + i * 256;
        pq[i] = -1;
        pdst[i] = NPX * PP;   // sink
        if (e < NGR) {
            const int pix = e >> G8SH;
            const int py = fdiv(pix, 1.0f / (float)PW), px = pix - py * PW;
            pq[i] = (py << 8) | px;
            pdst[i] = pix * PP + g8 * 8;
        }
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * CIN) * 2u);
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
    auto decode = [&](int it) {
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
    uint4 pv[SX];
    unsigned pok = 0;   // bit i: granule i came from inside the image (padding must stay 0 through the on-load affine)
    float va[8], vb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        va[k] = 1.f;
        vb[k] = 0.f;
    }
    auto issue = [&](const Item& I) {
        const int vy0 = I.ty0 * STRIDE - a.pad_t, vx0 = I.tx0 * STRIDE - a.pad_l;
        const unsigned short* xn = static_cast<const unsigned short*>(a.x) + (size_t)I.n * a.H * a.W * CIN;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(xn), 0, x_bytes, 0x00020000);
        pok = 0;
#pragma unroll
        for (int i = 0; i < SX; ++i) {
            const int sy = vy0 + (pq[i] >> 8), sx = vx0 + (pq[i] & 255);
            const bool ok = pq[i] >= 0 && (unsigned)sy < (unsigned)a.H && (unsigned)sx < (unsigned)a.W;
            pok |= ok ? (1u << i) : 0u;
            pv[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((sy * a.W + sx) * CIN + g8 * 8) * 2u : kOOB, 0, 0));
        }
        if (has_ab) {
            const float* pa = a.in_a + (size_t)I.n * a.in_nstride + g8 * 8;
            const float* pb = a.in_b + (size_t)I.n * a.in_nstride + g8 * 8;
            const float4 a0 = *reinterpret_cast<const float4*>(pa), a1 = *reinterpret_cast<const float4*>(pa + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(pb), b1 = *reinterpret_cast<const float4*>(pb + 4);
            va[0] = a0.x, va[1] = a0.y, va[2] = a0.z, va[3] = a0.w, va[4] = a1.x, va[5] = a1.y, va[6] = a1.z, va[7] = a1.w;
            vb[0] = b0.x, vb[1] = b0.y, vb[2] = b0.z, vb[3] = b0.w, vb[4] = b1.x, vb[5] = b1.y, vb[6] = b1.z, vb[7] = b1.w;
        }
    };
    auto relu1 = [](float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < SX; ++i) {
            uint4 v = pv[i];
            if (has_ab) {   // producer instance norm (+ ReLU) folded into the load: bf16 -> fp32, fma, -> bf16
                const unsigned okm = (pok >> i) & 1u ? 0xFFFFFFFFu : 0u;
                unsigned* w32 = reinterpret_cast<unsigned*>(&v);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float lo = fmaf(__uint_as_float(w32[k] << 16), va[2 * k], __uint_as_float(__float_as_uint(vb[2 * k]) & okm));
                    float hi = fmaf(__uint_as_float(w32[k] & 0xFFFF0000u), va[2 * k + 1], __uint_as_float(__float_as_uint(vb[2 * k + 1]) & okm));
                    if (in_relu) {
                        lo = relu1(lo);
                        hi = relu1(hi);
                    }
                    w32[k] = bs_pack2(lo, hi);
                }
            }
            *reinterpret_cast<uint4*>(patch + pdst[i]) = v;
        }
    };

    f32x16 acc[WM];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    };
    zero_acc();
    auto aoff = [&](int j) { return (((j / KC) / KW) * PW + ((j / KC) % KW) * DILX) * PP + (j % KC) * 16; };
    auto sweep = [&]() {
        constexpr int D = WM >= 4 ? 1 : 2;
        uint4 av[D + 1][WM];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int m = 0; m < WM; ++m)
                if (d < KSTEPS) av[d][m] = *reinterpret_cast<const uint4*>(patch + laneA[m] + aoff(d));
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
            if (j + D < KSTEPS) {
#pragma unroll
                for (int m = 0; m < WM; ++m) av[(j + D) % (D + 1)][m] = *reinterpret_cast<const uint4*>(patch + laneA[m] + aoff(j + D));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < WM; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bs_bf16x8, av[j % (D + 1)][m]), breg[j], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- epilogue, first half (before barrier B): statistics partials of this wave and the rounded tile into LDS.
    // Accumulator register r of lane (lm, kq), block m: tile pixel t = (mw*WM + m)*32 + (r & 3) + 8 (r >> 2) + 4 kq, i.e. row
    // 2 (mw*WM + m) + (r >> 3), column 4 kq + (r & 3) + 8 ((r >> 2) & 1); channel nbw*32 + lm.
    const int pyb = mw * WM * 2, pxb = kq * 4;
    auto epilogue_write = [&](const Item& I, float* rbuf) {
        const int th_valid = min(kBTH, a.Ho - I.ty0), tw_valid = min(kBTW, a.Wo - I.tx0);
        if (a.stats) {
            const float other = __shfl_xor(acc[0][0], 32);
            const float cs = kq ? other : acc[0][0];   // shift: the wave's own first pixel of the channel
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = pyb + 2 * m + (r >> 3) < th_valid && pxb + (r & 3) + 8 * ((r >> 2) & 1) < tw_valid;
                    const float d = ok ? acc[m][r] - cs : 0.f;
                    s1 += d;
                    s2 = fmaf(d, d, s2);
                }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lane < 32) {
                rbuf[(mw * 3 + 0) * BN + nbw * 32 + lane] = s1;
                rbuf[(mw * 3 + 1) * BN + nbw * 32 + lane] = s2;
                rbuf[(mw * 3 + 2) * BN + nbw * 32 + lane] = cs;
            }
        }
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = (mw * WM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                stage[t * BN + nbw * 32 + lm] = bs_f2bf(acc[m][r]);
            }
        zero_acc();
    };
    // ---- second half (after barrier B): 16-byte stores of whole pixel rows, every thread SO granules
    const int Cr = a.shuffle ? a.Cout >> 2 : a.Cout;
    const float inv_cr = 1.0f / (float)Cr;
    auto epilogue_store = [&](const Item& I) {
        const int th_valid = min(kBTH, a.Ho - I.ty0), tw_valid = min(kBTW, a.Wo - I.tx0);
        unsigned short* yb = static_cast<unsigned short*>(a.y) + (size_t)I.n * a.Ho * a.Wo * a.Cout;   // (same element count shuffled or not)
#pragma unroll
        for (int i = 0; i < SO; ++i) {
            const int e = tid + i * 256;
            const int pix = e >> OGSH, g = e & (OG - 1);
            const int py = pix >> 4, px = pix & 15;
            const int co = co0 + g * 8;
            if (py >= th_valid || px >= tw_valid || co >= a.Cout) continue;
            const int oy = I.ty0 + py, ox = I.tx0 + px;
            size_t o;
            if (a.shuffle) {
                const int q = fdiv(co, inv_cr), cof = co - q * Cr;
                o = ((size_t)(2 * oy + (q >> 1)) * (2 * a.Wo) + 2 * ox + (q & 1)) * Cr + cof;
            } else {
                o = ((size_t)oy * a.Wo + ox) * a.Cout + co;
            }
            *reinterpret_cast<uint4*>(yb + o) = *reinterpret_cast<const uint4*>(stage + pix * BN + g * 8);
        }
    };
    // merge of the per-wave records of one tile (Chan's update, fixed order) -> {mean, M2, count} of the tile
    auto finalize = [&](const Item& I, const float* rbuf) {
        if (co0 + tid >= a.Cout) return;
        const int th_valid = min(kBTH, a.Ho - I.ty0), tw_valid = min(kBTW, a.Wo - I.tx0);
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
        float* st = a.stats + ((size_t)I.lin * a.Cout + co0 + tid) * 3;
        st[0] = mean;
        st[1] = m2;
        st[2] = cnt;
    };

    // ---- the pipeline (one patch stage, as fs_cstream.hip)
    if (my_items == 0) return;
    if (tid < 8) patch[NPX * PP + tid] = 0;   // the sink (only ever rewritten with the zeros of out-of-range loads)
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
            issue(nxt);
        }
        sweep();
        FS_LDS_BARRIER();   // A: every wave is done reading the patch (and the previous tile's staged pixels have left)
        if (more) commit();
        epilogue_write(cur, red + (it & 1) * REDF);
        FS_LDS_BARRIER();   // B: next patch, this tile's statistics records and staged pixels visible
        epilogue_store(cur);
        prev = cur;
        cur = nxt;
    }
    if (a.stats && tid < BN) finalize(prev, red + ((my_items - 1) & 1) * REDF);
}

// ------------------------------------------------------------------------------------------------------------ host
namespace {
struct BsInst {
    int BN, Cin, KH, KW, stride, dil;
};
// 1: initconv_1 (16 -> 32, 3x3/2)   2: initconv_2 (32 -> 64, 3x3/2)   3: the residual convs (64 -> 64, 3x3)
// 4: 64 -> 32 resize-conv (2x2 taps, 128 virtual channels: two channel halves per tile)   5: 32 -> 16 resize-conv (2x2, 64 virtual)
// 6: the kw-folded output layer (9 x 2 taps, spacing 5, 16 virtual channels of a 32-wide block)
const BsInst kBs[6] = {{32, 16, 3, 3, 2, 1}, {64, 32, 3, 3, 2, 1}, {64, 64, 3, 3, 1, 1}, {64, 64, 2, 2, 1, 1}, {64, 32, 2, 2, 1, 1}, {32, 16, 9, 2, 1, 5}};
}  // namespace

int bstream_instance(const ConvBArgs& a) {
    if (!tune_int("FS_BSTREAM", 1) || a.x_f32 || a.y_f32 || a.src_mode != SRC_PLAIN) return 0;
    const int dil = a.dil_x > 0 ? a.dil_x : 1;
    const int bn = a.Cout > 32 ? 64 : 32;
    for (int i = 0; i < 6; ++i)
        if (a.Cin == kBs[i].Cin && a.KH == kBs[i].KH && a.KW == kBs[i].KW && a.stride == kBs[i].stride && dil == kBs[i].dil && bn == kBs[i].BN) {
            if (!((tune_int("FS_BSTREAM_MASK", 63) >> i) & 1)) return 0;
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
    p.c4 = 0;
    p.CC = a.Cin;
    p.PP = a.Cin + 8;
    p.TH = kBTH;
    p.TW = kBTW;
    p.tiles_y = cdiv(a.Ho, kBTH);
    p.tiles_x = cdiv(a.Wo, kBTW);
    p.PH = (kBTH - 1) * I.stride + I.KH;
    p.PW = (kBTW - 1) * I.stride + (I.KW - 1) * I.dil + 1;
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
    const int wgs = tune_int("FS_BSTREAM_WGS", 256);
    const int ny = p.cout_pad / p.BN;
    long gx = wgs / ny;   // one persistent workgroup per CU over both grid dimensions
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
        default: return -4;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
