// Third-generation path for the deep VGG16 3x3 convs (round 6; reference libs/vgg16.py:106-173: conv3_x / conv4_x, and their input gradients behind
// train.py:203): Winograd F(4x4,3x3) whose 36 Winograd-domain GEMMs run on the bf16 matrix cores as SIX EXACT PRODUCTS of bf16 pieces with fp32
// accumulation (U = Uh + Um + Ul, V = Vh + Vm + Vl, every piece 8 mantissa bits of the fp32 value: the split is exact; the six products
// Uh Vh, Uh Vm, Um Vh, Uh Vl, Ul Vh, Um Vm keep every term >= 2^-16 of the product, the three dropped ones are <= 2^-24).  Measured error against
// the float64 oracle: BELOW the fp32 F(4x4) kernel's (tools/bf16x3_error.py, tests/test_kernels_parity.py::test_winograd_f4x4_split_bf16_*).
//
// Why three launches instead of fs_wino4t.hip's one (DESIGN.md section 4 has the arithmetic): v_mfma_f32_32x32x16_bf16 multiplies 2.67x faster than the
// fp32 instruction sequence it replaces, but its operands are 1.5x the bytes (three 2-byte pieces).  A fused item is bounded by the register file
// (tiles x channels x 36 positions <= ~73 k accumulators per CU, i.e. 32 tiles x 64 channels): at that size the filter operand alone is 64 B/clk/CU from
// the L2 at the matrix rate (the fp32 kernel streams 11-22), and one K = 32 stage of V for 32 tiles is 221 KB of LDS.  Per POSITION the products are
// plain GEMMs [tiles x Cin] x [Cin x Cout] that tile 128 x 128 with both operands through LDS (16 B/clk/CU from the L2 per operand side) -- so:
//   K1 wino6_input_kernel   x -> V = B^T d B, fp32 [36][Cin/32][tiles][32]              (streaming; one thread = one tile x 4 channels)
//   K2 wino6_gemm_kernel    M[pos] = V[pos] U[pos]: 128 x 128 x 32 stages, V split into its three pieces on the way into LDS (the split costs
//                           ~90 vector-ALU instructions per thread and stage in the shadow of 48 matrix instructions), U pre-split by wt_wino6
//   K3 wino6_output_kernel  y = A^T M A + bias / ReLU / 2x2 max-pool / consumer ReLU mask (the epilogue forms of fs_wino4t.hip: 0, 3, 4)
// The price is V and M through HBM / the Infinity Cache (2.25x the input + 2.25x the output, fp32), which is why only the layers whose reduction is
// deep enough to pay for it take this path (FS_WINO6_MINCC: Cin * Cout >= 512 * 256, i.e. conv4_x both directions) and only under FS_WINO_V=6.
#include "fs_wino4.h"

namespace fs {

typedef __bf16 w6_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kW6TM = 128;                    // tiles per workgroup (GEMM M)
constexpr int kW6TN = 128;                    // output channels per workgroup (GEMM N)
constexpr int kW6K = 32;                      // input channels per stage
constexpr int kW6Pitch = 80;                  // bytes per LDS row of one piece: 64 B of k + 16 B of pad -- the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte bank groups
constexpr int kW6PieceB = 128 * kW6Pitch;     // one piece of one operand: 10240 B
constexpr int kW6StageB = 6 * kW6PieceB;      // A (V) three pieces + B (U) three pieces: 61440 B; two stages = 120 KB
#ifndef FS_W6_ABL
#define FS_W6_ABL 0   // timing builds (results wrong): 1 no split arithmetic, 2 no global loads in the loop, 4 no LDS stores, 8 no barrier, 16 no matrix instructions, 32 no operand reads
#endif

// fp32 -> three bf16 pieces by truncation: h = top 8 significant bits, m = the next 8 of the (exact) remainder, l = what is left (<= 8 bits): x = h + m + l exactly
__host__ __device__ __forceinline__ void w6_split(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    h = u & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, h);
    m = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, m);
    l = __builtin_bit_cast(unsigned, r2);
}

// U6[pos][ci/32][piece][co][ci%32] (bf16 bits): one workgroup stage of the GEMM (128 output channels x 32 input channels) is three contiguous 8 KB runs, one per
// piece, each in the row order of its LDS image (eight consecutive lanes store two rows: conflict-free 16-byte LDS stores).
// float64 transform rounded once to fp32 (as every F(4x4) filter layout of this library), then split exactly.
__global__ __launch_bounds__(256) void wt_wino6_kernel(const float* __restrict__ w, unsigned short* __restrict__ U, int Cin, int Cout) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t cc = (size_t)Cin * Cout;
    if (i >= cc) return;
    const int ci = (int)(i / Cout), co = (int)(i - (size_t)ci * Cout);
    double o[36];
    wino4_filter_transform(w, cc, i, o);
    const int KB = Cin >> 5;
#pragma unroll
    for (int pos = 0; pos < 36; ++pos) {
        unsigned h, m, l;
        w6_split((float)o[pos], h, m, l);
        unsigned short* d = U + ((((size_t)pos * KB + (ci >> 5)) * 3) * Cout + co) * 32 + (ci & 31);
        d[0] = (unsigned short)(h >> 16);
        d[(size_t)Cout * 32] = (unsigned short)(m >> 16);
        d[(size_t)Cout * 64] = (unsigned short)(l >> 16);
    }
}

int wt_wino6(const float* w, unsigned short* U, int Cin, int Cout, hipStream_t s) {
    if (Cin % kW6K || Cout % kW6TN) return -1;
    const size_t cc = (size_t)Cin * Cout;
    hipLaunchKernelGGL(wt_wino6_kernel, dim3((unsigned)((cc + 255) / 256)), dim3(256), 0, s, w, U, Cin, Cout);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

struct W6Args {
    const float* x;            // [N,H,W,Cin]
    float* y;                  // [N,Ho,Wo,Cout]
    const float* bias;
    const float* mask_src;
    float* pool_out;
    float* V;                  // [36][Cin/32][Tpad][32]
    const unsigned short* U;   // wt_wino6
    float* M;                  // [36][Tpad][Cout]
    int N, H, W, Cin, Ho, Wo, Cout, pad;
    int th, tw;                // tiles per sample
    int t0, T, Tpad;           // this launch's tile range [t0, t0 + T) of the N * th * tw tiles; Tpad = T rounded up to 128
    int out_relu, y_keep_n;
};

#define W6_BT4(d0, d1, d2, d3, d4, d5, t0, t1, t2, t3, t4, t5)   \
    do {                                                         \
        const auto a_ = d4 - 4.f * d2, b_ = d3 - 4.f * d1;       \
        const auto c_ = d4 - d2, e_ = d3 - d1;                   \
        const auto f_ = 4.f * d0 - 5.f * d2 + d4;                \
        const auto g_ = 4.f * d1 - 5.f * d3 + d5;                \
        t0 = f_; /* (outputs may alias inputs) */                \
        t1 = a_ + b_;                                            \
        t2 = a_ - b_;                                            \
        t3 = c_ + 2.f * e_;                                      \
        t4 = c_ - 2.f * e_;                                      \
        t5 = g_;                                                 \
    } while (0)
#define W6_AT4(m0, m1, m2, m3, m4, m5, y0, y1, y2, y3)   \
    do {                                                 \
        const auto p_ = m1 + m2, q_ = m1 - m2;           \
        const auto r_ = m3 + m4, s_ = m3 - m4;           \
        y0 = m0 + p_ + r_;                               \
        y1 = q_ + 2.f * s_;                              \
        y2 = p_ + 4.f * r_;                              \
        y3 = q_ + 8.f * s_ + m5;                         \
    } while (0)

// K1: input transform.  One thread = one tile x four channels: 36 16-byte loads (zero outside the image), B^T d B, 36 16-byte stores.
__global__ __launch_bounds__(256, 3) void wino6_input_kernel(W6Args a) {   // (<= 168 registers: a wave fits on a SIMD beside a resident GEMM wave of an independent launch)
    const int cq_n = a.Cin >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int tl = (int)(idx / cq_n), cq = (int)(idx - (long)tl * cq_n);
    if (tl >= a.T) return;
    const int t = a.t0 + tl;
    const int per = a.th * a.tw;
    const int n = t / per, r = t - n * per, ty = r / a.tw, tx = r - ty * a.tw;
    const int c = cq * 4;
    const float* xs = a.x + (size_t)n * a.H * a.W * a.Cin + c;
    const int y0 = 4 * ty - a.pad, x0 = 4 * tx - a.pad;
    f32x4 d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int yy = y0 + i, xx = x0 + j;
            const bool in = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            const float4 v = in ? *reinterpret_cast<const float4*>(xs + ((size_t)yy * a.W + xx) * a.Cin) : make_float4(0.f, 0.f, 0.f, 0.f);
            d[i][j] = f32x4{v.x, v.y, v.z, v.w};
        }
#pragma unroll
    for (int j = 0; j < 6; ++j) W6_BT4(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
    const int KB = a.Cin >> 5;
    float* vb = a.V + ((size_t)(c >> 5) * a.Tpad + tl) * 32 + (c & 31);
    const size_t pstride = (size_t)KB * a.Tpad * 32;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f32x4 o0, o1, o2, o3, o4, o5;
        W6_BT4(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], o0, o1, o2, o3, o4, o5);
        const f32x4 o[6] = {o0, o1, o2, o3, o4, o5};
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<float4*>(vb + (size_t)(i * 6 + j) * pstride) = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
    }
}

// K3: output transform + epilogue.  One thread = one tile x TWO output channels (8-byte accesses: 36 + 16 + 16 two-float values are ~110 registers, a wave
// fits on a SIMD beside a resident GEMM wave of an independent launch; with four channels per thread the kernel needs 200+ or spills).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int EPI>
__global__ __launch_bounds__(256, 3) void wino6_output_kernel(W6Args a) {
    const int cq_n = a.Cout >> 1;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int tl = (int)(idx / cq_n), cq = (int)(idx - (long)tl * cq_n);
    if (tl >= a.T) return;
    const int t = a.t0 + tl;
    const int per = a.th * a.tw;
    const int n = t / per, r = t - n * per, ty = r / a.tw, tx = r - ty * a.tw;
    const int c = cq * 2;
    const float* mb = a.M + (size_t)tl * a.Cout + c;
    const size_t pstride = (size_t)a.Tpad * a.Cout;
    f32x2 s[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x2 m[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float2 v = *reinterpret_cast<const float2*>(mb + (size_t)(i * 6 + j) * pstride);
            m[i] = f32x2{v.x, v.y};
        }
        W6_AT4(m[0], m[1], m[2], m[3], m[4], m[5], s[0][j], s[1][j], s[2][j], s[3][j]);
    }
    f32x2 o[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) W6_AT4(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], o[i][0], o[i][1], o[i][2], o[i][3]);
    const int oy = 4 * ty, ox = 4 * tx;
    f32x2 bs = f32x2{0.f, 0.f};
    if (EPI == 3 && a.bias) {
        const float2 b2 = *reinterpret_cast<const float2*>(a.bias + c);
        bs = f32x2{b2.x, b2.y};
    }
    const bool keep_y = !(EPI == 3 && a.pool_out && a.y_keep_n > 0 && n >= a.y_keep_n);
    float* yb = a.y + (size_t)n * a.Ho * a.Wo * a.Cout + c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 v = o[i][j];
            const bool in = oy + i < a.Ho && ox + j < a.Wo;
            const size_t off = ((size_t)(oy + i) * a.Wo + (ox + j)) * a.Cout;
            if (EPI == 3) {
                v = v + bs;
                if (a.out_relu) {
                    v[0] = fmaxf(v[0], 0.f);
                    v[1] = fmaxf(v[1], 0.f);
                }
                o[i][j] = v;
            }
            if (EPI == 4 && in) {
                const float2 k2 = *reinterpret_cast<const float2*>(a.mask_src + (size_t)n * a.Ho * a.Wo * a.Cout + c + off);
                v[0] = k2.x > 0.f ? v[0] : 0.f;
                v[1] = k2.y > 0.f ? v[1] : 0.f;
            }
            if (in && keep_y) *reinterpret_cast<float2*>(yb + off) = make_float2(v[0], v[1]);
        }
    if (EPI == 3 && a.pool_out) {   // 2x2/2 max-pool of the stored values: the tile's four windows (Ho, Wo even)
        const int Hp = a.Ho >> 1, Wp = a.Wo >> 1;
        float* pb = a.pool_out + (size_t)n * Hp * Wp * a.Cout + c;
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                if (oy + 2 * wy >= a.Ho || ox + 2 * wx >= a.Wo) continue;
                f32x2 q;
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    q[e] = fmaxf(fmaxf(o[2 * wy][2 * wx][e], o[2 * wy][2 * wx + 1][e]), fmaxf(o[2 * wy + 1][2 * wx][e], o[2 * wy + 1][2 * wx + 1][e]));
                *reinterpret_cast<float2*>(pb + ((size_t)((oy >> 1) + wy) * Wp + ((ox >> 1) + wx)) * a.Cout) = make_float2(q[0], q[1]);
            }
    }
}

// 8 fp32 -> the three bf16 pieces, packed for one 16-byte LDS store each (element e of a piece in the low / high half of word e / 2)
__device__ __forceinline__ void w6_split8(const float4& lo, const float4& hi, uint4& H, uint4& Mi, uint4& L) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w6_split(v[i], h[i], m[i], l[i]);
    H = make_uint4((h[0] >> 16) | h[1], (h[2] >> 16) | h[3], (h[4] >> 16) | h[5], (h[6] >> 16) | h[7]);
    Mi = make_uint4((m[0] >> 16) | m[1], (m[2] >> 16) | m[3], (m[4] >> 16) | m[5], (m[6] >> 16) | m[7]);
    L = make_uint4((l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u), (l[4] >> 16) | (l[5] & 0xffff0000u), (l[6] >> 16) | (l[7] & 0xffff0000u));
}

// K2: the 36 GEMMs.  Workgroup = (position, 128 tiles, 128 output channels), blocks of v_mfma_f32_32x32x16_bf16 (A = V: rows = tiles; B = U: columns = output
// channels; a lane of the result holds ONE output channel of 16 tiles -- the 32 lanes of a half-wave store 128 contiguous bytes of M[pos][tile][:]).  Stages of
// 32 input channels, two LDS stages, global loads three stages ahead.  NW = 4 (default): four waves 2 x 2, each 64 x 64 (one wave per SIMD); NW = 8 (FS_WINO6_WAVES=8, measured level or slower: its operand reads per matrix instruction are 1.5x): eight
// waves 2 x 4, each 64 x 32 -- TWO waves per SIMD, so that one wave's staging instructions (split arithmetic, LDS stores, operand reads, global loads: more
// issue cycles per stage than the matrix instructions themselves, profiles/r06_micro_split_bf16_pipeline.txt) issue while the other's matrix instructions run.
template <int NW>
__global__ __launch_bounds__(64 * NW) void wino6_gemm_kernel(W6Args a) {
    HIP_DYNAMIC_SHARED(float, smem)
    char* lds = reinterpret_cast<char*>(smem);
    constexpr int NT = 64 * NW;          // threads
    constexpr int WNB = NW == 4 ? 2 : 1; // 32-channel blocks per wave
    constexpr int AU = 512 / NT;         // A staging units (8 channels of one tile row: two quads) per thread and stage
    constexpr int BQ = 1536 / NT;        // B quads per thread and stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = NW == 4 ? wave >> 1 : wave >> 2, wn = NW == 4 ? wave & 1 : wave & 3;
    const int KB = a.Cin >> 5, MB = a.Tpad >> 7, NB = a.Cout >> 7;
    // XCD-aware order: workgroup b runs on XCD b mod 8; virtual index (b mod 8) G/8 + b/8 hands every XCD a contiguous range of (position, tile block,
    // channel block) with the channel blocks of a tile block adjacent: its 32 concurrent workgroups share 8 V tile blocks and the position's U in one L2
    const long G = (long)gridDim.x, b = (long)blockIdx.x;
    const long v = (G % 8 == 0) ? (b % 8) * (G / 8) + b / 8 : b;
    const int nb = (int)(v % NB), mb = (int)((v / NB) % MB), pos = (int)(v / ((long)NB * MB));
    const float* Ag = a.V + ((size_t)pos * KB * a.Tpad + (size_t)mb * 128) * 32;
    const size_t a_kstride = (size_t)a.Tpad * 32, b_pstride = (size_t)a.Cout * 32, b_kstride = 3 * b_pstride;
    const unsigned short* Bg = a.U + (size_t)pos * KB * b_kstride + (size_t)nb * 128 * 32;

    // staging: A 128 rows x 32 floats = 512 units of two quads, unit u = tid + NT i: row u / 4, 8-channel group u % 4; B 3 pieces x 512 quads, quad index
    // q = tid + NT j over [piece][512].  Two register sets: the global loads of stage s + 3 are issued in stage s and consumed (split, stored to LDS) in the
    // first half of stage s + 2 -- a stage and a half of matrix instructions between issue and use.  No conditional loads / stores (the lesson of fs_wino4.hip:
    // a staged value that is a phi of "loaded" and "not loaded" ends up in scratch memory): a stage beyond the last is loaded from the last one's addresses and
    // stored to the LDS buffer nobody reads.
    struct Regs {
        float4 a[AU][2];
        uint4 b[BQ];
    };
    auto load_regs = [&](Regs& R, int kb) __attribute__((always_inline)) {
        kb = kb < KB ? kb : KB - 1;
        const float* ap = Ag + (size_t)kb * a_kstride;
#pragma unroll
        for (int i = 0; i < AU; ++i) {
            const int u = tid + NT * i;
            R.a[i][0] = *reinterpret_cast<const float4*>(ap + (size_t)u * 8);
            R.a[i][1] = *reinterpret_cast<const float4*>(ap + (size_t)u * 8 + 4);
        }
        const unsigned short* bp = Bg + (size_t)kb * b_kstride;
#pragma unroll
        for (int j = 0; j < BQ; ++j) {
            const int q = tid + NT * j;   // (piece q / 512 is a compile-time constant per j: NT divides 512)
            R.b[j] = *reinterpret_cast<const uint4*>(bp + (size_t)(q >> 9) * b_pstride + (size_t)(q & 511) * 8);
        }
    };
    auto write_lds = [&](const Regs& R, int st) __attribute__((always_inline)) {
        char* base = lds + st * kW6StageB;
#pragma unroll
        for (int i = 0; i < AU; ++i) {
            const int u = tid + NT * i;
            uint4 H, Mi, L;
            if (FS_W6_ABL & 1) {
                H = __builtin_bit_cast(uint4, R.a[i][0]);
                Mi = __builtin_bit_cast(uint4, R.a[i][1]);
                L = H;
            } else
                w6_split8(R.a[i][0], R.a[i][1], H, Mi, L);
            if ((FS_W6_ABL & 4) && (H.x ^ Mi.y ^ L.z ^ H.w ^ Mi.x ^ L.y ^ H.z ^ Mi.w ^ L.x ^ H.y ^ Mi.z ^ L.w) != 0x9e3779b9u) continue;
            char* p = base + (u >> 2) * kW6Pitch + (u & 3) * 16;
            *reinterpret_cast<uint4*>(p) = H;
            *reinterpret_cast<uint4*>(p + kW6PieceB) = Mi;
            *reinterpret_cast<uint4*>(p + 2 * kW6PieceB) = L;
        }
#pragma unroll
        for (int j = 0; j < BQ; ++j) {
            const int q = tid + NT * j, r = q & 511;
            if ((FS_W6_ABL & 4) && (R.b[j].x ^ R.b[j].y ^ R.b[j].z ^ R.b[j].w) != 0x9e3779b9u) continue;
            *reinterpret_cast<uint4*>(base + (3 + (q >> 9)) * kW6PieceB + (r >> 2) * kW6Pitch + (r & 3) * 16) = R.b[j];
        }
    };

    // two accumulators per block: the leading product Uh Vh alone, and the five small ones (<= 2^-8 of it) together -- a rounding of the small sum is 2^-8 of a
    // rounding of the large one, so the result carries ONE full-size fp32 rounding per 16-channel block of the reduction; they meet once, in the store
    f32x16 acc[2][WNB], acl[2][WNB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WNB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = acl[i][j][e] = 0.f;

    const int a_lane = (64 * wm + (lane & 31)) * kW6Pitch + (lane >> 5) * 16;
    const int b_lane = 3 * kW6PieceB + (32 * WNB * wn + (lane & 31)) * kW6Pitch + (lane >> 5) * 16;

    // operand fragments of one half stage (16 of the 32 input channels): 2 tile blocks x 3 pieces of V, WNB channel blocks x 3 pieces of U
    struct Frag {
        w6_bf16x8 a[2][3], b[WNB][3];
    };
    auto read_frag = [&](Frag& F, int st, int ks) __attribute__((always_inline)) {
        const char* sa = lds + st * kW6StageB + a_lane + ks * 32;
        const char* sb = lds + st * kW6StageB + b_lane + ks * 32;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
                F.a[blk][p] = (FS_W6_ABL & 32) ? __builtin_bit_cast(w6_bf16x8, make_uint4(lane + p, blk, ks, 1))
                                               : __builtin_bit_cast(w6_bf16x8, *reinterpret_cast<const uint4*>(sa + p * kW6PieceB + blk * 32 * kW6Pitch));
#pragma unroll
            for (int blk = 0; blk < WNB; ++blk)
                F.b[blk][p] = (FS_W6_ABL & 32) ? __builtin_bit_cast(w6_bf16x8, make_uint4(lane, blk + p, ks, 2))
                                               : __builtin_bit_cast(w6_bf16x8, *reinterpret_cast<const uint4*>(sb + p * kW6PieceB + blk * 32 * kW6Pitch));
        }
    };
    // the matrix instructions of a half stage, product by product over the blocks (a block's accumulator is touched every 2 WNB-th instruction;
    // pieces: 0 = h, 1 = m, 2 = l; smallest terms first)
    auto mfma_half = [&](const Frag& F) __attribute__((always_inline)) {
        if (FS_W6_ABL & 16) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                asm volatile("" ::"v"(F.a[0][p]), "v"(F.a[1][p]), "v"(F.b[0][p]));
                if (WNB == 2) asm volatile("" ::"v"(F.b[WNB - 1][p]));
            }
            return;
        }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < WNB; ++j) {
                    if (t < 5) acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][PA[t]], F.b[j][PB[t]], acl[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[i][PA[t]], F.b[j][PB[t]], acc[i][j], 0, 0, 0);
                }
    };
    // One stage = two halves of matrix instructions, each with its operands read from LDS during the half BEFORE it (a wave issues in order: an operand
    // read waited for in front of its matrix instruction stalls the matrix pipe -- measured 100 of 245 us on conv4_2):
    //   first half : products of (stage s, channels 0..15) | reads of (s, 16..31) | split of register set R (= stage s + 1) into the other LDS buffer
    //   barrier    : buffer st ^ 1 complete; every wave has its reads of buffer st behind it (so the next stage may overwrite st)
    //   second half: products of (s, 16..31) | reads of (s + 1, 0..15) from the other buffer | global loads of stage s + 3 into R
    // The scheduler is asked to thread the reads / vector-ALU / LDS-store / global-load work between the matrix instructions.
    constexpr int NM = 12 * WNB;   // matrix instructions per half
    Frag F0, F1;
    auto stage = [&](int st, Regs& R, int kb_load) __attribute__((always_inline)) {
        mfma_half(F0);
        read_frag(F1, st, 1);
        write_lds(R, st ^ 1);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int g = 0; g < NM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one matrix instruction
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // an LDS read
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);   // vector-ALU instructions of the split
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // an LDS write
        }
        // nothing crosses the half's end: the groups above ask for more vector-ALU instructions than a half holds, and with both stages in one basic
        // block the scheduler otherwise fills them with the NEXT stage's split arithmetic -- which reads the other register set and drags its
        // s_waitcnt vmcnt up here (the loads issued half a stage ago)
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (!(FS_W6_ABL & 8)) __syncthreads();
        mfma_half(F1);
        read_frag(F0, st ^ 1, 0);
        if (!(FS_W6_ABL & 2)) load_regs(R, kb_load);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int g = 0; g < NM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a global load
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
    };

    Regs R0, R1;
    load_regs(R0, 0);
    write_lds(R0, 0);
    load_regs(R0, 1);
    load_regs(R1, 2);
    __syncthreads();
    read_frag(F0, 0, 0);
    // The prologue's loads have landed before the loop: the loop header's only pending set is then the back edge's, where set R0 is older than set R1
    // and stage 0 can wait for R0 alone.  The wait is the BUILTIN (the wait-count insertion pass reads it and clears its pending set; it cannot read an
    // inline-assembly s_waitcnt) between two empty assembly statements with a memory clobber (the scheduler otherwise sinks the prologue's loads BELOW the
    // builtin and interleaves the two sets -- after which every iteration's stage 0 ended its first half with s_waitcnt vmcnt(1)).
    FS_WAIT_VMEM_FENCED();
    // The loop body is two UNCONDITIONAL stages (an odd last stage runs after the loop) on purpose: with `if (kb + 1 < KB) stage(1, ...)` inside, the loop
    // header had a predecessor on which stage 0's own global loads were the youngest outstanding ones (the path that skips stage 1 -- never taken before the
    // exit, but the wait-count insertion merges it), so every iteration's stage 0 began with s_waitcnt vmcnt(6) and ended its first half with vmcnt(0): the
    // loads issued half a stage earlier (set R1) had to land inside half a stage of matrix instructions.  With one back edge and a drained preheader the
    // waits are exact -- vmcnt(17) ... vmcnt(10): each set is waited for a stage and a half after its issue, with the other set's ten loads still in
    // flight (conv4_2 at batch 32: 409 -> 355 us, same lease; profiles/r06_ab_wino6_exact_waits.txt).
    const int KB2 = KB & ~1;
    for (int kb = 0; kb < KB2; kb += 2) {
        stage(0, R0, kb + 3);   // stage kb: buffer 0; R0 holds stage kb + 1
        stage(1, R1, kb + 4);   // stage kb + 1: buffer 1; R1 holds stage kb + 2
    }
    if (KB & 1) stage(0, R0, KB + 2);   // (stage KB - 1: buffer 0; its staging half writes a stage nobody reads)
    // M[pos][tile][cout]: register r of block (i, j) = tile 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), channel 32 WNB wn + 32 j + (lane & 31)
    float* mp = a.M + ((size_t)pos * a.Tpad + (size_t)mb * 128 + 64 * wm + 4 * (lane >> 5)) * a.Cout + (size_t)nb * 128 + 32 * WNB * wn + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WNB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) mp[(size_t)(32 * i + (r & 3) + 8 * (r >> 2)) * a.Cout + 32 * j] = acc[i][j][r] + acl[i][j][r];
}

// ------------------------------------------------------------------------------------------------------------------------------------ host side
static int wino6_epi(const ConvArgs& a) {
    const bool act = a.bias || a.out_relu || a.pool_out;
    if (a.stats || a.add_src || a.inb_rec || a.in_a) return -1;
    if (a.mask_src) return act ? -1 : 4;
    return act ? 3 : 0;
}

size_t wino6_filter_floats(int Cin, int Cout) { return ((size_t)36 * Cin * Cout * 3 / 2 + 63) & ~(size_t)63; }   // 3 bf16 pieces per element, in floats

// workspace (floats) one launch wants for ALL its tiles in one pass; with less it runs in tile chunks (>= 128 tiles)
size_t wino6_ws_floats(int N, int Ho, int Wo, int Cin, int Cout) {
    const size_t T = (size_t)N * cdiv(Ho, 4) * cdiv(Wo, 4), Tpad = (T + 127) & ~(size_t)127;
    return 36 * Tpad * ((size_t)Cin + Cout);
}

bool wino6_eligible(const ConvArgs& a) {
    if (!a.w_wino6 || !a.w6_ws) return false;
    const bool pad_ok = a.pad_t == a.pad_l && a.pad_t >= 0 && a.pad_t <= 2 && a.Ho == a.H + 2 * a.pad_t - 2 && a.Wo == a.W + 2 * a.pad_l - 2;
    const long T = (long)a.N * cdiv(a.Ho, 4) * cdiv(a.Wo, 4);
    return a.KH == 3 && a.KW == 3 && a.stride == 1 && pad_ok && a.src_mode == SRC_PLAIN && a.Cin % kW6K == 0 && a.Cout % kW6TN == 0 && !a.shuffle && wino6_epi(a) >= 0 &&
           !a.route_src && a.w_nstride == 0 && a.dil_x <= 1 && !a.fin.counter && a.Ho > 0 && a.Wo > 0 && (!a.pool_out || (!(a.Ho & 1) && !(a.Wo & 1))) &&
           (long)a.Cin * a.Cout >= (long)tune_int("FS_WINO6_MINCC", 512 * 256) && T >= tune_int("FS_WINO6_MINTILES", 256) && T < (1L << 24) && 36.0 * (double)((T + 127) & ~127L) * (a.Cin > a.Cout ? a.Cin : a.Cout) * 4.0 < 4294967296.0 &&
           a.w6_ws_floats >= (size_t)36 * 128 * ((size_t)a.Cin + a.Cout);
}

void wino6_plan(const ConvArgs& a, ConvPlan* out) {
    ConvPlan p{};
    p.variant = 12;
    p.BN = kW6TN;
    p.CC = kW6K;
    p.TH = p.TW = 4;
    p.tiles_y = cdiv(a.Ho, 4);
    p.tiles_x = cdiv(a.Wo, 4);
    p.lds_bytes = 2 * kW6StageB;
    p.ksplit = 1;
    *out = p;
}

int wino6_launch(const ConvArgs& a, hipStream_t s) {
    const int epi = wino6_epi(a);
    if (epi < 0 || !wino6_eligible(a)) return -7;
    W6Args w{};
    w.x = a.x;
    w.y = a.y;
    w.bias = a.bias;
    w.mask_src = a.mask_src;
    w.pool_out = a.pool_out;
    w.U = a.w_wino6;
    w.N = a.N;
    w.H = a.H;
    w.W = a.W;
    w.Cin = a.Cin;
    w.Ho = a.Ho;
    w.Wo = a.Wo;
    w.Cout = a.Cout;
    w.pad = a.pad_t;
    w.th = cdiv(a.Ho, 4);
    w.tw = cdiv(a.Wo, 4);
    w.out_relu = a.out_relu;
    w.y_keep_n = a.y_keep_n;
    const long Tall = (long)a.N * w.th * w.tw;
    // tiles per pass: what the workspace holds (multiples of 128), optionally capped (FS_WINO6_CHUNK: a chunk whose V and M stay in the 256 MB Infinity Cache)
    long cap = (long)(a.w6_ws_floats / ((size_t)36 * ((size_t)a.Cin + a.Cout))) & ~127L;
    const int knob = tune_int("FS_WINO6_CHUNK", 0);
    if (knob >= 128 && (knob & ~127) < cap) cap = knob & ~127;
    if (cap < 128) return -7;
    const int nw = tune_int("FS_WINO6_WAVES", 4) == 8 ? 8 : 4;   // 8 measured level or slower (profiles/r06_ab_split_bf16_pipeline.txt): kept as an experiment
    static BigLds lds_attr4, lds_attr8;
    if (nw == 4) lds_attr4.ensure(reinterpret_cast<const void*>(wino6_gemm_kernel<4>));
    else lds_attr8.ensure(reinterpret_cast<const void*>(wino6_gemm_kernel<8>));
    auto k1 = [&](const W6Args& c, hipStream_t st) {
        const long n1 = (long)c.T * (a.Cin / 4);
        hipLaunchKernelGGL(wino6_input_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, c);
    };
    auto k2 = [&](const W6Args& c, hipStream_t st) {
        const dim3 g2((unsigned)(36 * (c.Tpad / 128) * (a.Cout / 128)));
        if (nw == 4) hipLaunchKernelGGL(wino6_gemm_kernel<4>, g2, dim3(256), (size_t)(2 * kW6StageB), st, c);
        else hipLaunchKernelGGL(wino6_gemm_kernel<8>, g2, dim3(512), (size_t)(2 * kW6StageB), st, c);
    };
    auto k3 = [&](const W6Args& c, hipStream_t st) {
        const long n3 = (long)c.T * (a.Cout / 2);
        const dim3 g3((unsigned)((n3 + 255) / 256));
        if (epi == 0) hipLaunchKernelGGL(wino6_output_kernel<0>, g3, dim3(256), 0, st, c);
        else if (epi == 3) hipLaunchKernelGGL(wino6_output_kernel<3>, g3, dim3(256), 0, st, c);
        else hipLaunchKernelGGL(wino6_output_kernel<4>, g3, dim3(256), 0, st, c);
    };
    auto chunk = [&](long t0, long T, float* base) {
        W6Args c = w;
        c.t0 = (int)t0;
        c.T = (int)T;
        c.Tpad = (c.T + 127) & ~127;
        c.V = base;
        c.M = base + (size_t)36 * c.Tpad * a.Cin;
        return c;
    };
    // Two chunks software-pipelined over two streams (a.w6_side + three events; FS_WINO6_PIPE=0 disables): the transforms are HBM-bound, the GEMM is not, and
    // their waves fit on a SIMD beside a GEMM wave --   s:    K1a | K2a        | K2b        | K3b
    //                                                   side:      | K1b        | K3a        |          (joined before returning)
    const long Ta = ((Tall / 2) + 127) & ~127L;
    if (a.w6_side && a.w6_ev && !Profiler::current() && tune_int("FS_WINO6_PIPE", 1) && Tall - Ta >= 128 && cap >= Ta + (((Tall - Ta) + 127) & ~127L)) {
        const W6Args ca = chunk(0, Ta, a.w6_ws);
        const W6Args cb = chunk(Ta, Tall - Ta, a.w6_ws + (size_t)36 * ca.Tpad * ((size_t)a.Cin + a.Cout));
        hipStream_t t = a.w6_side;
        k1(ca, s);
        if (hipEventRecord(a.w6_ev[0], s) != hipSuccess || hipStreamWaitEvent(t, a.w6_ev[0], 0) != hipSuccess) return -20;
        k1(cb, t);
        if (hipEventRecord(a.w6_ev[1], t) != hipSuccess) return -20;
        k2(ca, s);
        if (hipEventRecord(a.w6_ev[0], s) != hipSuccess || hipStreamWaitEvent(t, a.w6_ev[0], 0) != hipSuccess) return -20;
        k3(ca, t);
        if (hipEventRecord(a.w6_ev[2], t) != hipSuccess) return -20;
        if (hipStreamWaitEvent(s, a.w6_ev[1], 0) != hipSuccess) return -20;
        k2(cb, s);
        k3(cb, s);
        if (hipStreamWaitEvent(s, a.w6_ev[2], 0) != hipSuccess) return -20;
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    for (long t0 = 0; t0 < Tall; t0 += cap) {
        const W6Args c = chunk(t0, Tall - t0 < cap ? Tall - t0 : cap, a.w6_ws);
        k1(c, s);
        k2(c, s);
        k3(c, s);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
