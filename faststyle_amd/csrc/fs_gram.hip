// Gram matrices of the style loss, second generation (reference utils.py:66-83:  G[n] = F[n]^T F[n] / (h*w*c) with
// F[n] = the [h*w, c] feature map of sample n; tapped at conv1_2, conv2_2, conv3_3, conv4_3).
//
// Per sample a [C x HW] x [HW x C] product with a tiny result and a long reduction over the pixels.  What the first-generation
// kernel (fs_wgrad.hip, per_sample) loses is the same thing as everywhere on this path: stage -> wait -> multiply -> store per
// tile with nothing overlapped.  Here a workgroup owns ONE (sample, 128 x 128-channel output tile, pixel range) item for its
// whole life and streams the pixels through a single LDS stage, 64 pixels at a time, with the next 64 pixels' loads in
// flight during the sweep (registers -> one 16-byte LDS write per 16-byte load: the operand needs no transform).
//   * A and B operands come from the SAME feature map: rows of the output tile are the channels of group I, columns those of
//     group J; on the diagonal (I == J) one staged tile serves both and only the 10 of 16 32x32 blocks with j >= i are
//     multiplied -- the result is symmetric, the reduction kernel mirrors it;
//   * the four waves split the tile's BLOCKS, not its pixels: <= 4 blocks (64 accumulator registers) per wave, no
//     cross-wave reduction, one partial slab per workgroup; two workgroups fit a CU;
//   * v_mfma_f32_32x32x2_f32, K = a pair of pixels; partial slabs are summed in a fixed order (deterministic, no atomics).
#include "fs_kernels.h"

#include <cstdio>

#include <type_traits>

namespace fs {

namespace {
constexpr unsigned kOOB = 0x80000000u;
// pixels per staged tile: 64 x 128 channels, or 256 x 64 channels (C = 64 multiplies one block per wave and pixel pair:
// the longer tile gives the same matrix work between two barriers) -- 16-byte loads per thread, operand and tile: 8 / 16

struct GramArgs {
    const float* F;   // [N][HW][C]
    float* slabs;     // [N][pairs][splits][CG*CG]
    int N, HW, C;
    int CG;           // channels per group: 64 (C = 64) or 128
    int groups;       // C / CG
    int pairs;        // groups*(groups+1)/2: output tiles (I <= J)
    int splits;       // pixel ranges per (sample, tile)
};
}  // namespace

template <int CG>
__global__ __launch_bounds__(256, 2) void gram_stream_kernel(GramArgs a) {   // two workgroups per CU (<= 256 registers)
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, kq = lane >> 5;
    constexpr int S = CG + 4, c4n = CG >> 2, c4sh = c4n == 16 ? 4 : 5;
    constexpr int kGP = CG == 64 ? 256 : 64, kGL = kGP * c4n / 256;
    float* const tA = smem;
    float* const tB = smem + kGP * S;

    // ---- the item of this workgroup: ((n * pairs + pair) * splits + split); pair -> (I, J), I <= J, row-major upper triangle
    int lin = (int)blockIdx.x;
    const int split = lin % a.splits;
    lin /= a.splits;
    int pair = lin % a.pairs;
    const int n = lin / a.pairs;
    int I = 0;
    while (pair >= a.groups - I) {
        pair -= a.groups - I;
        ++I;
    }
    const int J = I + pair;
    const bool diag = I == J;
    const int p0 = (int)((long long)a.HW * split / a.splits), p1 = (int)((long long)a.HW * (split + 1) / a.splits);

    // ---- the blocks of this wave: (ib[k], jb[k]), k < nb.  128-channel tile off the diagonal: row block = wave, all four column
    // blocks; on the diagonal the ten blocks with j >= i are dealt 3, 3, 2, 2; 64-channel tile (C = 64): one block per wave.
    int ib[4], jb[4], nb;
    if (CG == 64) {
        nb = 1;
        ib[0] = wave >> 1;
        jb[0] = wave & 1;
        ib[1] = ib[2] = ib[3] = jb[1] = jb[2] = jb[3] = 0;
    } else if (!diag) {
        nb = 4;
        for (int k = 0; k < 4; ++k) {
            ib[k] = wave;
            jb[k] = k;
        }
    } else {
        // the ten (i, j), two bits each, in the order (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
        constexpr unsigned TI = (1u << 8) | (1u << 10) | (1u << 12) | (2u << 14) | (2u << 16) | (3u << 18);
        constexpr unsigned TJ = (1u << 2) | (2u << 4) | (3u << 6) | (1u << 8) | (2u << 10) | (3u << 12) | (2u << 14) | (3u << 16) | (3u << 18);
        const int beg = wave == 0 ? 0 : (wave == 1 ? 3 : (wave == 2 ? 6 : 8));
        nb = wave < 2 ? 3 : 2;
        for (int k = 0; k < 4; ++k) {
            const int kk = k < nb ? k : 0;   // (k >= nb: a redundant copy of the wave's first block, multiplied but never stored)
            ib[k] = (int)((TI >> (2 * (beg + kk))) & 3u);
            jb[k] = (int)((TJ >> (2 * (beg + kk))) & 3u);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ib[k] = __builtin_amdgcn_readfirstlane(ib[k]);
        jb[k] = __builtin_amdgcn_readfirstlane(jb[k]);
    }
    nb = __builtin_amdgcn_readfirstlane(nb);

    // ---- staging: element e = tid + i*256 of a 64-pixel x CG-channel tile: pixel e / (CG/4), channel quad e % (CG/4)
    const unsigned f_bytes = __builtin_amdgcn_readfirstlane((unsigned)((size_t)a.HW * a.C * 4));
    const float* Fn = a.F + (size_t)n * a.HW * a.C;
    {
        const unsigned long long v = reinterpret_cast<unsigned long long>(Fn);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        Fn = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    }
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Fn), 0, f_bytes, 0x00020000);
    const int nel = kGP * c4n;                       // elements per operand tile: 1024 (CG = 64) or 2048
    int epix[kGL], edst[kGL];
    unsigned ecol[kGL];
#pragma unroll
    for (int i = 0; i < kGL; ++i) {
        const int e = tid + i * 256;
        const int pix = e >> c4sh, c4 = e & (c4n - 1);
        epix[i] = e < nel ? pix : 0x40000000;        // (beyond the tile: never inside a pixel range -> no load, no write)
        edst[i] = pix * S + c4 * 4;
        ecol[i] = (unsigned)(c4 * 4) * 4u;
    }
    float4 va[kGL], vb[kGL];
    auto issue = [&](int pbase) {   // pixels [pbase, pbase + 64) of the range; beyond p1: zeros (out-of-range offset)
#pragma unroll
        for (int i = 0; i < kGL; ++i) {
            const int px = pbase + epix[i];
            const bool ok = px < p1;
            const unsigned rowb = (unsigned)px * (unsigned)a.C * 4u;
            va[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(fr, ok ? rowb + (unsigned)(I * CG) * 4u + ecol[i] : kOOB, 0, 0));
            if (!diag)
                vb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(fr, ok ? rowb + (unsigned)(J * CG) * 4u + ecol[i] : kOOB, 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < kGL; ++i)
            if (epix[i] < kGP) {
                *reinterpret_cast<float4*>(tA + edst[i]) = va[i];
                if (!diag) *reinterpret_cast<float4*>(tB + edst[i]) = vb[i];
            }
    };

    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const int boff = diag ? 0 : kGP * S;   // (an integer offset: selecting between two LDS pointers makes them generic pointers)
    // operand offsets of the wave's blocks (lane lm = channel inside the block, lane half kq = pixel of the pair)
    int oa[4], ob[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        oa[k] = kq * S + ib[k] * 32 + lm;
        ob[k] = boff + kq * S + jb[k] * 32 + lm;
    }
    // NBK blocks per wave, compile-time: a sweep without a branch between its matrix instructions (a wave-uniform `if` around
    // each MFMA keeps the compiler from hoisting the LDS reads of the next pixel pair over it: measured 57 TFLOP/s).  The
    // diagonal deal is 3, 3, 2, 2: the two waves with two blocks multiply a third, redundant one (block k = 2 repeats their
    // first) that is never stored.
    auto sweep = [&](auto NBKT) {
        constexpr int NBK = decltype(NBKT)::value;
        // fully unrolled, every operand address a lane base + a compile-time offset; explicit software pipeline pinned with
        // sched_barrier: the operands of pixel pair j + D are read before the matrix instructions of pair j
        constexpr int D = NBK >= 3 ? 2 : 3;
        float av[D + 1][NBK], bv[D + 1][NBK];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int k = 0; k < NBK; ++k) {
                av[d][k] = smem[oa[k] + 2 * d * S];
                bv[d][k] = smem[ob[k] + 2 * d * S];
            }
#pragma unroll
        for (int j = 0; j < kGP / 2; ++j) {
            if (j + D < kGP / 2) {
#pragma unroll
                for (int k = 0; k < NBK; ++k) {
                    av[(j + D) % (D + 1)][k] = smem[oa[k] + 2 * (j + D) * S];
                    bv[(j + D) % (D + 1)][k] = smem[ob[k] + 2 * (j + D) * S];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NBK; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j % (D + 1)][k], bv[j % (D + 1)][k], acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- the pipeline over the 64-pixel tiles of the range
    const int ntiles = (p1 - p0 + kGP - 1) / kGP;
    if (ntiles > 0) {
        issue(p0);
        commit();
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            const bool more = t + 1 < ntiles;
            if (more) issue(p0 + (t + 1) * kGP);
            if constexpr (CG == 64) {
                sweep(std::integral_constant<int, 1>{});
            } else {
                if (diag)
                    sweep(std::integral_constant<int, 3>{});
                else
                    sweep(std::integral_constant<int, 4>{});
            }
            __syncthreads();   // every wave is done reading the stage
            if (more) commit();
            __syncthreads();
        }
    }
    // ---- the partial slab of this workgroup: [CG][CG], row = channel of group I, column = channel of group J
    float* slab = a.slabs + (size_t)blockIdx.x * CG * CG;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= nb) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ib[k] * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
            slab[row * CG + jb[k] * 32 + lm] = acc[k][r];
        }
    }
}

// Round 6: the 128-channel tiles in six exact bf16-piece products (the arithmetic of fs_wino6.hip / conv_s16x_kernel).  v_mfma_f32_32x32x16_bf16 takes its 8
// k values per lane from consecutive PIXELS of one channel, the feature map is pixel-major: the commit TRANSPOSES on the way into LDS -- a thread loads 8
// consecutive pixels x 4 channels (eight 16-byte loads, the channel quad of a lane is contiguous across lanes), splits the 32 values and writes, per channel
// and piece, 8 pixels = one 16-byte LDS store into [piece][channel][32 pixels] (row pitch 80 bytes: 5 slots of 16, odd -- the 16 channels of a
// ds_read_b128 lane group fall on 16 distinct slots).  32-pixel tiles (two k-steps between barriers; 61 KB for both operands: two workgroups per CU), threads
// 0..127 stage the row group, 128..255 the column group.  The six products of a step accumulate smallest-first in ONE accumulator per block (64 + 60 fragment
// registers leave no room for a second set under 256).
typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
__host__ __device__ __forceinline__ void gs_split(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    h = u & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, h);
    m = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, m);
    l = __builtin_bit_cast(unsigned, r2);
}
__global__ __launch_bounds__(256, 2) void gram_streamx_kernel(GramArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    char* const lds = reinterpret_cast<char*>(smem);
    constexpr int CG = 128, TP = 32;                       // channels per group, pixels per staged tile
    constexpr int RB = TP * 2 + 16;                        // bytes per channel row of a piece plane
    constexpr int PLB = CG * RB;                           // one piece plane: 10240 bytes
    constexpr int OPB = 3 * PLB;                           // one operand: 30720 bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, kq = lane >> 5;

    int lin = (int)blockIdx.x;
    const int split = lin % a.splits;
    lin /= a.splits;
    int pair = lin % a.pairs;
    const int n = lin / a.pairs;
    int I = 0;
    while (pair >= a.groups - I) {
        pair -= a.groups - I;
        ++I;
    }
    const int J = I + pair;
    const bool diag = I == J;
    const int p0 = (int)((long long)a.HW * split / a.splits), p1 = (int)((long long)a.HW * (split + 1) / a.splits);

    int ib[4], jb[4], nb;
    if (!diag) {
        nb = 4;
        for (int k = 0; k < 4; ++k) {
            ib[k] = wave;
            jb[k] = k;
        }
    } else {
        constexpr unsigned TI = (1u << 8) | (1u << 10) | (1u << 12) | (2u << 14) | (2u << 16) | (3u << 18);
        constexpr unsigned TJ = (1u << 2) | (2u << 4) | (3u << 6) | (1u << 8) | (2u << 10) | (3u << 12) | (2u << 14) | (3u << 16) | (3u << 18);
        const int beg = wave == 0 ? 0 : (wave == 1 ? 3 : (wave == 2 ? 6 : 8));
        nb = wave < 2 ? 3 : 2;
        for (int k = 0; k < 4; ++k) {
            const int kk = k < nb ? k : 0;
            ib[k] = (int)((TI >> (2 * (beg + kk))) & 3u);
            jb[k] = (int)((TJ >> (2 * (beg + kk))) & 3u);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ib[k] = __builtin_amdgcn_readfirstlane(ib[k]);
        jb[k] = __builtin_amdgcn_readfirstlane(jb[k]);
    }
    nb = __builtin_amdgcn_readfirstlane(nb);

    // ---- staging: thread (operand op = tid >> 7, pixel octet o = (tid >> 5) & 3, channel quad c4 = tid & 31) loads pixels 8 o .. + 7, channels 4 c4 .. + 3
    const unsigned f_bytes = __builtin_amdgcn_readfirstlane((unsigned)((size_t)a.HW * a.C * 4));
    const float* Fn = a.F + (size_t)n * a.HW * a.C;
    {
        const unsigned long long v = reinterpret_cast<unsigned long long>(Fn);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        Fn = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    }
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Fn), 0, f_bytes, 0x00020000);
    const int op = tid >> 7, oct = (tid >> 5) & 3, c4 = tid & 31;
    const bool stager = op == 0 || !diag;                  // (on the diagonal one staged tile serves both operands)
    const unsigned colb = (unsigned)(((op ? J : I) * CG + c4 * 4) * 4);
    const int wdst = op * OPB + (c4 * 4) * RB + oct * 16;   // + e RB (channel) + piece PLB
    float4 pv[8];
    auto issue = [&](int pbase) {   // pixels [pbase, pbase + 32) of the range; beyond p1: zeros (out-of-range offset)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int px = pbase + oct * 8 + i;
            const bool ok = stager && px < p1;
            pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(fr, ok ? (unsigned)px * (unsigned)a.C * 4u + colb : kOOB, 0, 0));
        }
    };
    auto commit = [&]() {
        if (!stager) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned h[8], m[8], l[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = e == 0 ? pv[i].x : (e == 1 ? pv[i].y : (e == 2 ? pv[i].z : pv[i].w));
                gs_split(v, h[i], m[i], l[i]);
            }
            char* d = lds + wdst + e * RB;
            *reinterpret_cast<uint4*>(d) = make_uint4((h[0] >> 16) | h[1], (h[2] >> 16) | h[3], (h[4] >> 16) | h[5], (h[6] >> 16) | h[7]);
            *reinterpret_cast<uint4*>(d + PLB) = make_uint4((m[0] >> 16) | m[1], (m[2] >> 16) | m[3], (m[4] >> 16) | m[5], (m[6] >> 16) | m[7]);
            *reinterpret_cast<uint4*>(d + 2 * PLB) = make_uint4((l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u), (l[4] >> 16) | (l[5] & 0xffff0000u),
                                                                (l[6] >> 16) | (l[7] & 0xffff0000u));
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const int boff = diag ? 0 : OPB;
    int oa[4], ob[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        oa[k] = (ib[k] * 32 + lm) * RB + kq * 16;
        ob[k] = boff + (jb[k] * 32 + lm) * RB + kq * 16;
    }
    auto sweep = [&](auto NBKT, auto SHAREA) {
        constexpr int NBK = decltype(NBKT)::value;
        constexpr bool shareA = decltype(SHAREA)::value;   // off the diagonal the wave's four blocks have ONE row block: its fragments are read once
        constexpr int NA = shareA ? 1 : NBK;
#pragma unroll
        for (int j = 0; j < TP / 16; ++j) {
            gs_bf16x8 av[NA][3], bv[NBK][3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
                for (int k = 0; k < NA; ++k) av[k][pc] = __builtin_bit_cast(gs_bf16x8, *reinterpret_cast<const uint4*>(lds + oa[k] + pc * PLB + j * 32));
#pragma unroll
                for (int k = 0; k < NBK; ++k) bv[k][pc] = __builtin_bit_cast(gs_bf16x8, *reinterpret_cast<const uint4*>(lds + ob[k] + pc * PLB + j * 32));
            }
            __builtin_amdgcn_sched_barrier(0);
            // the six products, smallest first, block after block inside a product: consecutive matrix instructions never share an accumulator
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int k = 0; k < NBK; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[shareA ? 0 : k][PA[t]], bv[k][PB[t]], acc[k], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int ntiles = (p1 - p0 + TP - 1) / TP;
    if (ntiles > 0) {
        issue(p0);
        commit();
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            const bool more = t + 1 < ntiles;
            if (more) issue(p0 + (t + 1) * TP);
            if (diag)
                sweep(std::integral_constant<int, 3>{}, std::false_type{});
            else
                sweep(std::integral_constant<int, 4>{}, std::true_type{});
            __syncthreads();   // every wave is done reading the stage
            if (more) commit();
            __syncthreads();
        }
    }
    // A 32 x 32 block ON the diagonal holds both of its halves, computed separately -- and in the split arithmetic element (i, j) and element (j, i) add the
    // same six products in different orders (h l <-> l h, h m <-> m h): equal to rounding, not bit for bit as with ONE fp32 product.  The lower triangle takes
    // the upper one's values (through LDS, free after the last sweep; every wave of a diagonal tile owns exactly one such block): G stays exactly symmetric.
    if (diag) {
        float* const tr = smem + wave * (32 * 33);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nb && ib[k] == jb[k]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * kq) * 33 + lm] = acc[k][r];
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nb && ib[k] == jb[k]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * kq;
                    const float t = tr[lm * 33 + i];
                    acc[k][r] = i <= lm ? acc[k][r] : t;
                }
            }
    }
    float* slab = a.slabs + (size_t)blockIdx.x * CG * CG;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= nb) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ib[k] * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
            slab[row * CG + jb[k] * 32 + lm] = acc[k][r];
        }
    }
}

// G[n][(I*CG + r)*C + J*CG + c] = scale * sum over the pixel ranges, and its mirror image.  grid (CG*CG/1024, pairs, N); a thread
// owns 4 consecutive columns of one row.  On a diagonal tile the blocks below the diagonal were never multiplied: their
// elements arrive as the mirror of the blocks above.
__global__ __launch_bounds__(256) void gram_reduce_kernel(GramArgs a, float scale, float* __restrict__ G) {
    const int CG = a.CG;
    const int e4 = (int)blockIdx.x * 256 + (int)threadIdx.x;   // float4 index inside the tile
    if (e4 * 4 >= CG * CG) return;
    const int r = (e4 * 4) / CG, c = (e4 * 4) - r * CG;
    int pair = (int)blockIdx.y, I = 0;
    while (pair >= a.groups - I) {
        pair -= a.groups - I;
        ++I;
    }
    const int J = I + pair, n = (int)blockIdx.z;
    if (I == J && CG == 128 && (r >> 5) > (c >> 5)) return;
    const float* p = a.slabs + (((size_t)n * a.pairs + blockIdx.y) * a.splits) * CG * CG + (size_t)r * CG + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < a.splits; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * CG * CG);
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
    }
    s.x *= scale;
    s.y *= scale;
    s.z *= scale;
    s.w *= scale;
    float* Gn = G + (size_t)n * a.C * a.C;
    const int gr = I * CG + r, gc = J * CG + c;
    *reinterpret_cast<float4*>(Gn + (size_t)gr * a.C + gc) = s;
    const bool mirror = I != J || (CG == 128 && (r >> 5) < (c >> 5));   // (blocks ON the diagonal hold both halves already)
    if (mirror) {
        Gn[(size_t)(gc + 0) * a.C + gr] = s.x;
        Gn[(size_t)(gc + 1) * a.C + gr] = s.y;
        Gn[(size_t)(gc + 2) * a.C + gr] = s.z;
        Gn[(size_t)(gc + 3) * a.C + gr] = s.w;
    }
}

// The same reduction for SEVERAL layers in one launch, with the style-loss terms of losses.py:61-64 on the way (round 5: a train step had a
// reduce, a squared-difference and a sum launch per style layer -- twelve launches of a few microseconds of work each):
//   G = scale * sum of slabs (+ mirror image), S = gscale * (G - Gt), partial[block] = sum over the block's elements of (G - Gt)^2 with every
//   element of the full C x C matrix counted once (a mirrored element counts twice).
// grid (16, sum of the jobs' pairs, N); blocks beyond a job's tile size return.
namespace {
struct GramFinishArgs {
    struct J {
        GramArgs a;
        const float* Gt;
        float* G;
        float* S;
        float* partial;
        float scale, gscale;
        int pair0;   // first blockIdx.y of the job
    } j[4];
    int n;
};
}  // namespace
__global__ __launch_bounds__(256) void gram_finish_kernel(GramFinishArgs f) {
    __shared__ float sh[4];
    __shared__ float tile[32][33];
    int q = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < f.n && (int)blockIdx.y >= f.j[k].pair0) q = k;
    const GramFinishArgs::J& J_ = f.j[q];
    const GramArgs& a = J_.a;
    const int CG = a.CG;
    const int bx = (CG * CG / 4 + 255) / 256;   // 32 x 32 sub-blocks of the CG x CG tile, one per workgroup
    if ((int)blockIdx.x >= bx) return;   // (uniform per block)
    const int pair_id = (int)blockIdx.y - J_.pair0;
    // (round 5) a workgroup owns a 32 x 32 sub-block: rows of 128 contiguous bytes for the direct half, and -- through a transposing LDS tile --
    // rows of 128 contiguous bytes of the MIRRORED half too (the element-wise mirror wrote and read single floats C apart: 110 us per batch-32 step)
    const int sbn = CG >> 5, sr = (int)blockIdx.x / sbn, sc = (int)blockIdx.x - sr * sbn;
    const int tr = (int)threadIdx.x >> 3, tc = ((int)threadIdx.x & 7) * 4;
    const int r = sr * 32 + tr, c = sc * 32 + tc;
    int pair = pair_id, I = 0;
    while (pair >= a.groups - I) {
        pair -= a.groups - I;
        ++I;
    }
    const int J = I + pair, n = (int)blockIdx.z;
    const bool active = !(I == J && CG == 128 && sr > sc);                 // (block-uniform; diagonal 128-tiles hold their upper sub-blocks only)
    const bool mirror = I != J || (CG == 128 && sr < sc);                  // (blocks ON the diagonal hold both halves already)
    float acc = 0.f;
    if (active) {
        const float* p = a.slabs + (((size_t)n * a.pairs + pair_id) * a.splits) * CG * CG + (size_t)r * CG + c;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < a.splits; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * CG * CG);
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
            s.w += v.w;
        }
        s.x *= J_.scale;
        s.y *= J_.scale;
        s.z *= J_.scale;
        s.w *= J_.scale;
        const size_t cc = (size_t)a.C * a.C;
        float* Gn = J_.G ? J_.G + (size_t)n * cc : nullptr;   // (G itself is optional: the training step reads S only)
        float* Sn = J_.S + (size_t)n * cc;
        const int gr = I * CG + r, gc = J * CG + c;
        const float4 t = *reinterpret_cast<const float4*>(J_.Gt + (size_t)gr * a.C + gc);
        const float d[4] = {s.x - t.x, s.y - t.y, s.z - t.z, s.w - t.w};
        const float g = J_.gscale;
        if (Gn) *reinterpret_cast<float4*>(Gn + (size_t)gr * a.C + gc) = s;
        *reinterpret_cast<float4*>(Sn + (size_t)gr * a.C + gc) = make_float4(g * d[0], g * d[1], g * d[2], g * d[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fmaf(d[e], d[e], acc);
        if (mirror) {
            tile[tr][tc + 0] = s.x;
            tile[tr][tc + 1] = s.y;
            tile[tr][tc + 2] = s.z;
            tile[tr][tc + 3] = s.w;
            __syncthreads();   // (block-uniform branch)
            // mirrored row = a column of the sub-block: element (J CG + sc 32 + tr, I CG + sr 32 + tc ..) = sub-block elements (tc .. tc + 3, tr)
            const float sv[4] = {tile[tc + 0][tr], tile[tc + 1][tr], tile[tc + 2][tr], tile[tc + 3][tr]};
            const size_t mo = (size_t)(J * CG + sc * 32 + tr) * a.C + (size_t)(I * CG + sr * 32 + tc);
            // (the target is symmetric up to rounding, the product of another kernel: the mirrored element takes ITS target)
            const float4 tm = *reinterpret_cast<const float4*>(J_.Gt + mo);
            const float dm[4] = {sv[0] - tm.x, sv[1] - tm.y, sv[2] - tm.z, sv[3] - tm.w};
            if (Gn) *reinterpret_cast<float4*>(Gn + mo) = make_float4(sv[0], sv[1], sv[2], sv[3]);
            *reinterpret_cast<float4*>(Sn + mo) = make_float4(g * dm[0], g * dm[1], g * dm[2], g * dm[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = fmaf(dm[e], dm[e], acc);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) J_.partial[((size_t)n * a.pairs + pair_id) * bx + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------------------------------------------------------
// Gradient of the style loss through a Gram matrix (the adjoint of utils.py:76-82 behind train.py:203):
//     dF[n] = F[n] . S[n]   (+ an optional addend: the content-loss gradient of a layer that carries both terms),
// S[n] = 4 w (G[n] - G_target) / (c^2 h w c), symmetric [C x C] -- a 1x1 convolution with per-sample filters.  Streaming form
// (cf. fs_cstream.hip): a workgroup owns a contiguous range of pixel tiles of ONE sample (and, for C = 256, one half of the
// output channels), keeps its slice of S[n] in REGISTERS (K/2 x NB values per lane), and runs tile after tile through one LDS
// stage with the next tile's loads in flight during the sweep.
//   C =  64: tile 256 pixels, waves 4 x 1 over (pixel blocks x channel blocks), 2 x 2 blocks per wave,  64 filter registers
//   C = 128: tile 256 pixels, waves 1 x 4, 8 x 1 blocks per wave,                                        64
//   C = 256: tile 128 pixels, waves 1 x 4, 4 x 1 blocks per wave, two workgroup groups (128 channels each), 128
namespace {
struct GramBwdArgs {
    const float* F;     // [N][HW][C]
    const float* S;     // [N][C][C]
    const float* add;   // optional [N][HW][C]
    float* dF;          // [N][HW][C]
    int N, HW, C;
    int wpg;            // workgroups per (sample, channel half)
    const float* above; // RT: [N][H/2][W/2][C], the gradient of max_pool(F)
    int W;              // RT: map width (H = HW / W)
    const float* content;   // RT, optional [N][HW][C]: the content features -- the addend is cscale * (F - content) (the content-loss gradient, formed
    float cscale;           // here instead of read), and cpartial[workgroup] = the workgroup's sum of (F - content)^2 (the loss's partial sums)
    float* cpartial;
};
}  // namespace

// RT (round 5): the kernel also does what vgg_bwd_route did in a pass of its own behind it -- the result is
//     d_pre = (dF + the max-pool gradient `above` routed to the FIRST maximum of each 2x2 window of F) * (F > 0)
// (TF MaxPoolGrad + ReluGrad; the sums in vgg_bwd_route_kernel's order, so the two paths agree bit for bit).  F is this kernel's own A operand:
// a tile is two map rows x TPX/2 columns, laid into the LDS stage so that 32-pixel block 2b holds row 0 and block 2b + 1 row 1 of the same 32
// columns -- the four pixels of a window are then the accumulator registers (m, r), (m, r + 1), (m + 1, r), (m + 1, r + 1) of ONE lane, and their
// F values four LDS reads of the stage the sweep just used.  Needs an even H and W a multiple of TPX/2 (the VGG maps of 256 x 256 inputs).
template <int C, int WN, int TPX, bool RT = false>
__global__ __launch_bounds__(256) void gram_bwd_kernel(GramBwdArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int NH = C > 128 ? C / 128 : 1;              // channel halves (workgroup groups)
    constexpr int CW = C / NH;                             // output channels of one workgroup
    constexpr int WMW = 4 / WN, NB = CW / 32 / WN, WM = TPX / 32 / WMW;
    constexpr int S = C + 1, C4 = C / 4, KSTEPS = C / 2;
    constexpr int C4SH = C4 == 16 ? 4 : (C4 == 32 ? 5 : 6);
    constexpr int SX = TPX * C4 / 256;                     // 16-byte loads per thread and tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, kq = lane >> 5;
    const int mw = wave % WMW, nbw = wave / WMW;
    // ---- this workgroup: sample n, channel half nh, tiles [t_beg, t_end) of the sample
    int lin = (int)blockIdx.x;
    const int wi = lin % a.wpg;
    lin /= a.wpg;
    const int nh = lin % NH, n = lin / NH;
    constexpr int TW = TPX / 2;                            // RT: tile columns
    static_assert(!RT || ((WM % 2) == 0 && TW % 32 == 0), "RT tiles: row pairs of whole 32-column blocks");
    const int tpr = RT ? a.W / TW : 1;                     // RT: tiles per row pair
    const int tiles = RT ? (a.HW / a.W / 2) * tpr : (a.HW + TPX - 1) / TPX;
    const int t_beg = (int)((long long)tiles * wi / a.wpg), t_end = (int)((long long)tiles * (wi + 1) / a.wpg);
    if (t_beg >= t_end) {
        if (RT && a.cpartial && tid == 0) a.cpartial[blockIdx.x] = 0.f;
        return;
    }
    const int co0 = nh * CW + nbw * NB * 32;               // first output channel of this wave
    // RT: first pixel of tile t = rows 2 ty, 2 ty + 1, columns tx TW ..
    auto tile_p0 = [&](int t) {
        const int ty = t / tpr, tx = t - ty * tpr;
        return 2 * ty * a.W + tx * TW;
    };

    // ---- S[n] rows k = 2j + kq, columns co0 + nn*32 + lm: resident for the workgroup's lifetime
    const float* Sn = a.S + (size_t)n * C * C;
    float breg[KSTEPS][NB];
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j)
#pragma unroll
        for (int nn = 0; nn < NB; ++nn) breg[j][nn] = Sn[(2 * j + kq) * C + co0 + nn * 32 + lm];

    // (the row blocks of a wave span more than 64 KB: one base register per 64 KB, each behind an opaque copy, keeps every read at
    // "register + 16-bit immediate" -- folded onto ONE base, half of the reads got a v_add each between the matrix instructions)
    constexpr int MPB = (65536 / (32 * S * 4)) > 0 ? (65536 / (32 * S * 4)) : 1;   // row blocks per base register
    constexpr int NBASE = (WM + MPB - 1) / MPB;
    int laneB[NBASE];
#pragma unroll
    for (int g = 0; g < NBASE; ++g) {
        laneB[g] = ((mw * WM + g * MPB) * 32 + lm) * S + kq;
        if (g > 0) FS_OPAQUE(laneB[g]);
    }
    int laneA[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) laneA[m] = laneB[m / MPB] + (m % MPB) * 32 * S;

    const float* Fn = a.F + (size_t)n * a.HW * C;
    {
        const unsigned long long v = reinterpret_cast<unsigned long long>(Fn);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        Fn = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    }
    const unsigned f_bytes = __builtin_amdgcn_readfirstlane((unsigned)((size_t)a.HW * C * 4));
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Fn), 0, f_bytes, 0x00020000);
    // staging: element e = tid + i*256 = (pixel e / C4, channel quad e % C4): consecutive threads read consecutive 16 bytes
    float4 pv[SX];
    auto issue = [&](int t) {
        if constexpr (RT) {
            // pass i fills stage rows i P .. i P + P - 1 (P = 256 / C4 pixels, P divides 32): one map row, 32-column block (i P) >> 6, columns from
            // (i P) & 31 -- the thread's part of the address is tid * 16 as in the plain form, the pass's part is a scalar; whole tiles only
            constexpr int P = 256 / C4;
            const unsigned base = (unsigned)tile_p0(t) * (unsigned)(C * 4), row1 = (unsigned)a.W * (unsigned)(C * 4);
#pragma unroll
            for (int i = 0; i < SX; ++i) {
                const int qi = i * P;
                const unsigned soff = base + (((qi >> 5) & 1) ? row1 : 0u) + (unsigned)(((qi >> 6) * 32 + (qi & 31)) * C * 4);
                pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(fr, (unsigned)tid * 16u, __builtin_amdgcn_readfirstlane(soff), 0));
            }
            return;
        }
        const unsigned base = (unsigned)(t * TPX) * (unsigned)(C * 4);
#pragma unroll
        for (int i = 0; i < SX; ++i) {   // (pixels beyond the map: out-of-range offset -> zeros)
            const unsigned off = base + (unsigned)(tid + i * 256) * 16u;
            pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(fr, off < f_bytes ? off : kOOB, 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < SX; ++i) {
            const int e = tid + i * 256;
            float* d = smem + (e >> C4SH) * S + (e & (C4 - 1)) * 4;   // (RT: the same stage rows -- issue() chose the pixels to suit)
            d[0] = pv[i].x;
            d[1] = pv[i].y;
            d[2] = pv[i].z;
            d[3] = pv[i].w;
        }
    };
    f32x16 acc[WM][NB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int nn = 0; nn < NB; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;
    };
    zero_acc();
    // explicit software pipeline, pinned with sched_barrier: the A operands of pixel-pair step q + D are read before the matrix
    // instructions of step q.  (Left to itself the compiler, short of registers, emits read -> wait -> two DEPENDENT MFMAs
    // per operand: measured 2.3x the matrix time.)
    auto sweep = [&]() {
        constexpr int D = WM >= 4 ? 1 : 2;
        float av[D + 1][WM];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int m = 0; m < WM; ++m) av[d][m] = smem[laneA[m] + 2 * d];
#pragma unroll
        for (int q = 0; q < KSTEPS; ++q) {
            if (q + D < KSTEPS) {
#pragma unroll
                for (int m = 0; m < WM; ++m) av[(q + D) % (D + 1)][m] = smem[laneA[m] + 2 * (q + D)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nn = 0; nn < NB; ++nn)
#pragma unroll
                for (int m = 0; m < WM; ++m) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q % (D + 1)][m], breg[q][nn], acc[m][nn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    float* dFn = a.dF + (size_t)n * a.HW * C;
    {
        const unsigned long long v = reinterpret_cast<unsigned long long>(dFn);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        dFn = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
    }
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(dFn, 0, f_bytes, 0x00020000);
    const float* addn = a.add ? a.add + (size_t)n * a.HW * C : nullptr;
    const float* contn = RT && a.content ? a.content + (size_t)n * a.HW * C : nullptr;
    float csum = 0.f;
    // (RT: every global access of the routing pass is a buffer operation "lane offset register + SCALAR offset": the per-element parts are
    // wave-uniform, and as 64-bit lane addresses they were hoisted out of the tile loop into ~100 spilled registers)
    auto rsrc_of = [&](const float* p, unsigned bytes) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t cr = rsrc_of(RT ? (contn ? contn : (addn ? addn : Fn)) : Fn, f_bytes);                     // content features or addend
    const __amdgpu_buffer_rsrc_t ar = rsrc_of(RT ? a.above + (size_t)n * (a.HW / 4) * C : Fn, RT ? f_bytes / 4 : f_bytes);   // pooled gradient
    auto ldf = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, __builtin_amdgcn_readfirstlane(soff), 0));
    };
    // ---- RT: the pooled gradient of the tile's windows (loaded beside the next tile, before the sweep), the routing pass over the accumulators
    // (after the sweep, while the stage still holds F), and the store
    constexpr int NW = RT ? (WM / 2) * NB * 8 : 1;
    float da[NW];
    auto issue_above = [&](int t) {
        const int ty = t / tpr, tx = t - ty * tpr;
        // window (mp, nn, wd) of this lane: column (tx TW + (mw WM / 2 + mp) 32 + pi) / 2 of pooled row ty, pi = 2 (wd & 1) + 8 (wd >> 1) + 4 kq
        const unsigned base = ((unsigned)(ty * (a.W >> 1) + ((tx * TW + (mw * WM / 2) * 32 + 4 * kq) >> 1)) * (unsigned)C + (unsigned)(co0 + lm)) * 4u;
#pragma unroll
        for (int mp = 0; mp < WM / 2; ++mp)
#pragma unroll
            for (int nn = 0; nn < NB; ++nn)
#pragma unroll
                for (int wd = 0; wd < 8; ++wd) da[(mp * NB + nn) * 8 + wd] = ldf(ar, base, (unsigned)((mp * 16 + (wd & 1) + 4 * (wd >> 1)) * C + nn * 32) * 4u);
    };
    auto route = [&](int t) {
        const int p0 = tile_p0(t);
        const float* frow = smem + (mw * WM * 32 + 4 * kq) * S + co0 + lm;   // F of this lane's pixels: stage row (mw WM + m) 32 + pi
        const unsigned lane_off = ((unsigned)(p0 + (mw * WM / 2) * 32 + 4 * kq) * (unsigned)C + (unsigned)(co0 + lm)) * 4u;
        const unsigned row1 = (unsigned)a.W * (unsigned)(C * 4);
        constexpr int CH = 16;
        if (contn) {   // the content term formed here: cscale * (F - content) added, (F - content)^2 summed -- sqdiff_kernel's products, rounded as there
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int nn = 0; nn < NB; ++nn)
#pragma unroll
                    for (int h = 0; h < 16; h += CH) {   // (C = 256: eight at a time -- registers)
                        float fc[CH], ff[CH];
#pragma unroll
                        for (int k = 0; k < CH; ++k) {
                            const int r = h + k;
                            fc[k] = ldf(cr, lane_off, (m & 1) * row1 + (unsigned)(((m >> 1) * 32 + (r & 3) + 8 * (r >> 2)) * C + nn * 32) * 4u);
                            ff[k] = frow[(m * 32 + (r & 3) + 8 * (r >> 2)) * S + nn * 32];
                        }
#pragma unroll
                        for (int k = 0; k < CH; ++k) {
                            const float d = ff[k] - fc[k];
                            csum = fmaf(d, d, csum);
                            acc[m][nn][h + k] += __fmul_rn(a.cscale, d);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
        } else if (addn) {   // (sixteen loads at a time, as the plain epilogue's)
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int nn = 0; nn < NB; ++nn) {
                    float ad[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ad[r] = ldf(cr, lane_off, (m & 1) * row1 + (unsigned)(((m >> 1) * 32 + (r & 3) + 8 * (r >> 2)) * C + nn * 32) * 4u);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][nn][r] += ad[r];
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        // groups of two windows (eight F values), the reads of group g + 1 in flight behind the arithmetic of group g: with one wave per SIMD
        // nothing else hides the LDS latency (all 128 reads first: registers; each group waited for on its own: 2.3x the routing time)
        constexpr int NG = (WM / 2) * NB * 4;
        float fb[2][8];
        auto read_f = [&](int g, float (&f)[8]) {
            const int mp = g / (NB * 4), nn = (g >> 2) % NB;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int wd = 2 * (g & 3) + (k >> 2), e = k & 3, r0 = 2 * wd, pi = (r0 & 3) + 8 * (r0 >> 2);
                f[k] = frow[((2 * mp + (e >> 1)) * 32 + pi + (e & 1)) * S + nn * 32];
            }
        };
        read_f(0, fb[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) read_f(g + 1, fb[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int mp = g / (NB * 4), nn = (g >> 2) % NB;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int wd = 2 * (g & 3) + h, r0 = 2 * wd;
                const float* f = &fb[g & 1][4 * h];
                const float dav = da[(mp * NB + nn) * 8 + wd];
                // the first maximum of the window in the order (0,0) (0,1) (1,0) (1,1)
                const bool w0 = f[0] >= f[1] && f[0] >= f[2] && f[0] >= f[3];
                const bool w1 = f[1] > f[0] && f[1] >= f[2] && f[1] >= f[3];
                const bool w2 = f[2] > f[0] && f[2] > f[1] && f[2] >= f[3];
                const bool w3 = f[3] > f[0] && f[3] > f[1] && f[3] > f[2];
                const bool win[4] = {w0, w1, w2, w3};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gv = acc[2 * mp + (e >> 1)][nn][r0 + (e & 1)];
                    const float v = win[e] ? gv + dav : gv;
                    acc[2 * mp + (e >> 1)][nn][r0 + (e & 1)] = f[e] > 0.f ? v : 0.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto store_rt = [&](int t) {
        const unsigned lane_off = ((unsigned)(tile_p0(t) + (mw * WM / 2) * 32 + 4 * kq) * (unsigned)C + (unsigned)(co0 + lm)) * 4u;
        const unsigned row1 = (unsigned)a.W * (unsigned)(C * 4);
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int nn = 0; nn < NB; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned soff = (m & 1) * row1 + (unsigned)(((m >> 1) * 32 + (r & 3) + 8 * (r >> 2)) * C + nn * 32) * 4u;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[m][nn][r]), yr, lane_off, __builtin_amdgcn_readfirstlane(soff), 0);
                }
        zero_acc();
    };
    auto epilogue = [&](int t) {
        // accumulator register r of lane (lm, kq): pixel (mw*WM + m)*32 + (r & 3) + 8 (r >> 2) + 4 kq of the tile, channel co0 + nn*32 + lm.
        // Byte offset = lane part + compile-time part; pixels beyond the map get the out-of-range offset and are dropped.
        const unsigned lane_off = ((unsigned)(t * TPX + mw * WM * 32 + 4 * kq) * (unsigned)C + (unsigned)(co0 + lm)) * 4u;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int nn = 0; nn < NB; ++nn) {
                float ad[16];
                if (addn) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned off = lane_off + (unsigned)((m * 32 + (r & 3) + 8 * (r >> 2)) * C + nn * 32) * 4u;
                        ad[r] = off < f_bytes ? addn[off >> 2] : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned off = lane_off + (unsigned)((m * 32 + (r & 3) + 8 * (r >> 2)) * C + nn * 32) * 4u;
                    float v = acc[m][nn][r];
                    if (addn) v += ad[r];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, off < f_bytes ? off : kOOB, 0, 0);
                }
            }
        zero_acc();
    };
    issue(t_beg);
    commit();
    __syncthreads();
    for (int t = t_beg; t < t_end; ++t) {
        const bool more = t + 1 < t_end;
        if (more) issue(t + 1);
        if constexpr (RT) issue_above(t);
        sweep();
        if constexpr (RT) route(t);
        FS_LDS_BARRIER();
        if (more) commit();
        if constexpr (RT) store_rt(t);
        else epilogue(t);
        FS_LDS_BARRIER();   // (LDS only: the tile's stores drain during the next sweep instead of being waited for here)
    }
    if (RT && a.cpartial) {   // the workgroup's partial sum of the content loss: waves in a fixed order (the stage is free: the loop ended on a barrier)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) csum += __shfl_xor(csum, o);
        if (lane == 0) smem[wave] = csum;
        __syncthreads();
        if (tid == 0) a.cpartial[blockIdx.x] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    }
}

// ------------------------------------------------------------------------------------------------------------ host
static bool gram2_shape(int N, int HW, int C, GramArgs* out) {
    if (!(C == 64 || (C % 128 == 0 && C <= 1024)) || N < 1 || HW < 1) return false;
    if ((size_t)HW * C * 4 >= 0x80000000ull) return false;   // (32-bit byte offsets inside one sample)
    GramArgs a{};
    a.N = N;
    a.HW = HW;
    a.C = C;
    a.CG = C == 64 ? 64 : 128;
    a.groups = C / a.CG;
    a.pairs = a.groups * (a.groups + 1) / 2;
    // pixel ranges: enough items for ~3 waves of workgroups over the chip, at least ~4 staged tiles each
    int splits = 1;
    const int target = tune_int("FS_GRAM2_ITEMS", 768);
    const int min_px = tune_int("FS_GRAM2_MIN_TILES", 4) * (a.CG == 64 ? 256 : 64);   // (tests lower it: several ranges on small maps)
    while ((long)N * a.pairs * splits < target && HW / (splits * 2) >= (min_px > 1 ? min_px : 1)) splits *= 2;
    a.splits = splits;
    *out = a;
    return true;
}

bool gram2_eligible(int N, int HW, int C) {
    GramArgs a;
    return tune_int("FS_GRAM2", 1) != 0 && gram2_shape(N, HW, C, &a);
}

size_t gram2_slab_floats(int N, int HW, int C) {
    GramArgs a;
    if (!gram2_shape(N, HW, C, &a)) return 0;
    return (size_t)N * a.pairs * a.splits * a.CG * a.CG;
}

int gram2_finish_partials(int N, int C) {
    GramArgs a;
    if (!gram2_shape(N, 1, C, &a)) return 0;   // (pairs and the tile size do not depend on HW)
    return N * a.pairs * ((a.CG * a.CG / 4 + 255) / 256);
}

int gram2_finish_batch(const GramFinishJob* jobs, int n, int N, hipStream_t s) {
    if (n < 1 || n > 4) return -1;
    GramFinishArgs f{};
    f.n = n;
    int pairs = 0;
    for (int k = 0; k < n; ++k) {
        GramFinishArgs::J& j = f.j[k];
        if (!gram2_shape(N, jobs[k].HW, jobs[k].C, &j.a)) return -1;
        j.a.slabs = const_cast<float*>(jobs[k].slabs);
        j.Gt = jobs[k].Gt;
        j.G = jobs[k].G;
        j.S = jobs[k].S;
        j.partial = jobs[k].partial;
        j.scale = jobs[k].scale;
        j.gscale = jobs[k].gscale;
        j.pair0 = pairs;
        pairs += j.a.pairs;
    }
    hipLaunchKernelGGL(gram_finish_kernel, dim3(16, (unsigned)pairs, (unsigned)N), dim3(256), 0, s, f);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

static int gram2_stream_impl(const GramArgs& a, int N, int HW, hipStream_t s);

int gram2_stream(const float* F, float* slabs, int N, int HW, int C, hipStream_t s) {
    GramArgs a;
    if (!gram2_shape(N, HW, C, &a)) return -1;
    a.F = F;
    a.slabs = slabs;
    return gram2_stream_impl(a, N, HW, s);
}

// G[n] = scale * F[n]^T F[n]; slabs: gram2_slab_floats(N, HW, C) floats of scratch
int gram2_launch(const float* F, float* G, float* slabs, int N, int HW, int C, float scale, hipStream_t s) {
    GramArgs a;
    if (!gram2_shape(N, HW, C, &a)) return -1;
    a.F = F;
    a.slabs = slabs;
    if (const int rc = gram2_stream_impl(a, N, HW, s)) return rc;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)cdiv(a.CG * a.CG / 4, 256), (unsigned)a.pairs, (unsigned)N), dim3(256), 0, s, a, scale, G);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

static int gram2_stream_impl(const GramArgs& a, int N, int HW, hipStream_t s) {
    const size_t lds = a.CG == 64 ? (size_t)256 * 68 * sizeof(float) : (size_t)2 * 64 * 132 * sizeof(float);
    Profiler* prof = Profiler::current();
    // FLOPs EXECUTED: diagonal 128-channel tiles multiply 10 of their 16 blocks
    const double blocks = a.CG == 64 ? 4.0 * 1 : (16.0 * (a.pairs - a.groups) + 10.0 * a.groups);
    if (prof) prof->begin(PF_GRAM_STREAM, 2.0 * N * (double)HW * 32.0 * 32.0 * blocks, s);
    if (a.CG == 64) {
        static BigLds lds_attr;
        lds_attr.ensure(reinterpret_cast<const void*>(gram_stream_kernel<64>));
        hipLaunchKernelGGL(gram_stream_kernel<64>, dim3((unsigned)(N * a.pairs * a.splits)), dim3(256), lds, s, a);
    } else if (tune_int("FS_GRAM_SPLIT", 1)) {   // six exact bf16-piece products (gram_streamx_kernel): two operands x three piece planes of 128 channels x 80 bytes
        hipLaunchKernelGGL(gram_streamx_kernel, dim3((unsigned)(N * a.pairs * a.splits)), dim3(256), (size_t)(2 * 3 * 128 * 80), s, a);
    } else {
        static BigLds lds_attr;
        lds_attr.ensure(reinterpret_cast<const void*>(gram_stream_kernel<128>));
        hipLaunchKernelGGL(gram_stream_kernel<128>, dim3((unsigned)(N * a.pairs * a.splits)), dim3(256), lds, s, a);
    }
    if (prof) prof->end(s);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}


bool gram_bwd2_eligible(int N, int HW, int C) {
    if (!tune_int("FS_GRAM_BWD2", 1) || N < 1 || HW < 1) return false;
    if (!(C == 64 || C == 128 || C == 256)) return false;
    return (size_t)HW * C * 4 < 0x7F000000ull;   // 32-bit byte offsets inside one sample, with room for a tile of overshoot
}

// S[n][i][j] = scale * (dG[n][i][j] + dG[n][j][i]): the symmetrised, scaled upstream gradient of a Gram matrix (fs_gram_bwd)
__global__ __launch_bounds__(256) void gram_symmetrize_kernel(const float* __restrict__ dG, float* __restrict__ S, int C, float scale, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t cc = (size_t)C * C;
    const size_t n = i / cc, r = i - n * cc;
    const int row = (int)(r / C), col = (int)(r - (size_t)row * C);
    S[i] = scale * (dG[i] + dG[n * cc + (size_t)col * C + row]);
}

int gram_symmetrize(const float* dG, float* S, int N, int C, float scale, hipStream_t s) {
    const size_t total = (size_t)N * C * C;
    hipLaunchKernelGGL(gram_symmetrize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dG, S, C, scale, total);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// the routed form (gram_bwd_kernel<.., RT = true>): whole tiles of two rows x TPX/2 columns
bool gram_bwd2_route_eligible(int N, int H, int W, int C) {
    if (!tune_int("FS_GRAM_ROUTE_FUSED", 1) || H < 2 || W < 2 || !gram_bwd2_eligible(N, H * W, C)) return false;
    const int TW = C == 64 ? 128 : C == 128 ? 64 : 32;
    return !(H & 1) && W % TW == 0;
}

// workgroups of the routed launch (= the partial sums a content term leaves)
int gram_bwd2_route_grid(int N, int H, int W, int C) {
    const int NH = C > 128 ? C / 128 : 1, TPX = C == 64 ? 256 : C == 128 ? 128 : 64;
    const int tiles = (H / 2) * (W / (TPX / 2));
    int wpg = tune_int("FS_GRAM_BWD2_WGS", 256) / (N * NH);
    if (wpg < 1) wpg = 1;
    if (wpg > tiles) wpg = tiles;
    return N * NH * wpg;
}

// dF[n] = F[n] S[n] (+ add[n]);  above != nullptr (gram_bwd2_route_eligible(N, HW / W, W, C)): dF = (that + the max-pool gradient `above`
// ([N][H/2][W/2][C]) routed through F) * (F > 0)
int gram_bwd2_launch(const float* F, const float* S, const float* add, float* dF, int N, int HW, int C, hipStream_t s, const float* above, int W,
                     const float* content, float cscale, float* cpartial) {
    if (!gram_bwd2_eligible(N, HW, C)) return -1;
    if (above && (W < 1 || HW % W || !gram_bwd2_route_eligible(N, HW / W, W, C))) return -1;
    if (content && (!above || add || !cpartial)) return -1;
    GramBwdArgs a{};
    a.above = above;
    a.W = W;
    a.content = content;
    a.cscale = cscale;
    a.cpartial = cpartial;
    a.F = F;
    a.S = S;
    a.add = add;
    a.dF = dF;
    a.N = N;
    a.HW = HW;
    a.C = C;
    const int NH = C > 128 ? C / 128 : 1, TPX = above ? (C == 64 ? 256 : C == 128 ? 128 : 64) : (C == 256 ? 128 : 256);   // (routed: half the pixels for C >= 128 -- the plain tile sizes spill with the routing pass's registers)
    const int tiles = above ? (HW / W / 2) * (W / (TPX / 2)) : cdiv(HW, TPX);
    int wpg = tune_int("FS_GRAM_BWD2_WGS", 256) / (N * NH);
    if (wpg < 1) wpg = 1;
    if (wpg > tiles) wpg = tiles;
    a.wpg = wpg;
    const unsigned grid = (unsigned)(N * NH * wpg);
    const size_t lds = (size_t)TPX * (C + 1) * sizeof(float);
    if (tune_int("FS_CONV_DEBUG", 0))
        fprintf(stderr, "gram_bwd2: N %d HW %d (W %d) C %d tile %d px%s%s, %d tiles / sample, %u workgroups\n", N, HW, W, C, TPX, above ? " + pool routing + mask" : "",
                content ? " + content term" : (add ? " + addend" : ""), tiles, grid);
    Profiler* prof = Profiler::current();
    if (prof) prof->begin(PF_GRAM_BWD, 2.0 * N * (double)HW * C * C, s);
    if (above) {
        static BigLds l64, l128, l256;
        if (C == 64) {
            l64.ensure(reinterpret_cast<const void*>(gram_bwd_kernel<64, 1, 256, true>));
            hipLaunchKernelGGL((gram_bwd_kernel<64, 1, 256, true>), dim3(grid), dim3(256), lds, s, a);
        } else if (C == 128) {
            l128.ensure(reinterpret_cast<const void*>(gram_bwd_kernel<128, 4, 128, true>));
            hipLaunchKernelGGL((gram_bwd_kernel<128, 4, 128, true>), dim3(grid), dim3(256), lds, s, a);
        } else {
            l256.ensure(reinterpret_cast<const void*>(gram_bwd_kernel<256, 4, 64, true>));
            hipLaunchKernelGGL((gram_bwd_kernel<256, 4, 64, true>), dim3(grid), dim3(256), lds, s, a);
        }
    } else if (C == 64) {
        static BigLds lds_attr;
        lds_attr.ensure(reinterpret_cast<const void*>(gram_bwd_kernel<64, 1, 256>));
        hipLaunchKernelGGL((gram_bwd_kernel<64, 1, 256>), dim3(grid), dim3(256), lds, s, a);
    } else if (C == 128) {
        static BigLds lds_attr;
        lds_attr.ensure(reinterpret_cast<const void*>(gram_bwd_kernel<128, 4, 256>));
        hipLaunchKernelGGL((gram_bwd_kernel<128, 4, 256>), dim3(grid), dim3(256), lds, s, a);
    } else {
        static BigLds lds_attr;
        lds_attr.ensure(reinterpret_cast<const void*>(gram_bwd_kernel<256, 4, 128>));
        hipLaunchKernelGGL((gram_bwd_kernel<256, 4, 128>), dim3(grid), dim3(256), lds, s, a);
    }
    if (prof) prof->end(s);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
