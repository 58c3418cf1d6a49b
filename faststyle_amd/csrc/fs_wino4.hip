// Winograd F(4x4, 3x3) convolution on the fp32 matrix cores (round 4): the 3x3 stride-1 SAME convolutions of VGG16
// (reference libs/vgg16.py:36-220: conv1_2 ... conv4_3 forward, and their input gradients in the training step).
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A        (Lavin & Gray, interpolation points 0, +-1, +-2, inf; cross-correlation form)
//
// A 6x6 input tile (stride 4) yields a 4x4 output tile from 36 element-wise products instead of 144 multiply-adds: summed
// over the input channels, 36 independent GEMMs [tiles x Cin] x [Cin x Cout] with 4x less matrix work than the direct form
// and 1.78x less than F(2x2,3x3) (fs_wino.hip / fs_wino2.hip).  fp32 throughout; the transforms multiply by 2, 4, 5, 8 and
// the filter transform by 1/4, 1/6, 1/12, 1/24, so the rounding error is about ten times that of F(2x2) -- measured
// ~1e-5 of the output's magnitude at 512 input channels against 1e-6 (tests/test_kernels_parity.py holds it to 5e-5 against the
// float64 oracle; the north-star budget is 1e-3).
//
// Mapping (one wave per SIMD, 256 threads, persistent workgroups over a strided item list -- the recipe of fs_wino2.hip):
//   * item = 32 tiles (4 x 8 tiles = 16 x 32 output pixels) x 64 output channels x all 36 positions = 73,728 accumulators,
//     288 registers per lane: a wave owns 16 tiles x 32 channels x 36 positions as 72 blocks of v_mfma_f32_16x16x4_f32 with
//     the CHANNELS on the matrix instruction's row side -- a lane then holds, for ONE tile, four consecutive channels of
//     every position, so the output transform A^T M A runs in registers and leaves as 16-byte stores;
//   * the input channels are walked in chunks of 4 (= the K of one matrix instruction): per chunk and wave 72 matrix
//     instructions (2304 cycles), one 8-byte LDS read (both channel blocks of a position: U is laid out [pos][half][k][m][2])
//     and one 4-byte read (V [pos][half][k][tile]) per pair of them, addresses = one lane-constant base + an immediate;
//   * the two LDS stages hold, per chunk, the transformed filter U (36 KB, a straight 16-byte copy of the pre-transformed
//     filter in HBM, fs::wt_wino4), the transformed input V (18 KB) and the raw 18 x 34 x 4 patch (channel-planar, plane pitch
//     = 1 mod 32: the transform's reads and the loaders' writes are conflict-free);
//   * WAVE SPECIALISATION for the staging of the next chunk, threaded through the 72 matrix-instruction slots of the sweep:
//     waves 0-1 transform (one 6x6 block per thread: 36 LDS reads, 144 vector instructions bunched into four gaps, 36 LDS
//     writes), waves 2-3 load (the 36 KB filter chunk: 18 x 16-byte loads + LDS writes per thread, the patch of the
//     chunk after next: 5 loads + 20 LDS writes) -- both about 1.1k cycles beside the 2.3k of matrix instructions, because
//     beside the fp32 matrix instruction every vector-ALU / vector-memory instruction costs its issue time (tools/mfma_overlap.hip);
//   * epilogue per item: bias + ReLU (+ the 2x2 max-pool of the tile's four windows) for the forward, the consumer's
//     ReLU mask for the input gradients; split-K (raw partials) where the launch cannot fill the chip.
#include "fs_kernels.h"

#include <cstdlib>
#include <type_traits>

namespace fs {

namespace {
constexpr int kTY = 4, kTX = 8;                  // tiles per block: rows x columns (a tile = 4 x 4 output pixels)
constexpr int kNT = kTY * kTX;                   // 32 tiles
constexpr int kBH = 4 * kTY, kBW = 4 * kTX;      // 16 x 32 output pixels per block
constexpr int kPH = kBH + 2, kPW = kBW + 2;      // 18 x 34 input patch
constexpr int kPP = kPH * kPW;                   // 612 patch pixels
constexpr int kCC = 4;                           // input channels per chunk
constexpr int kBN = 64;                          // output channels per item
constexpr int kPlane = 641;                      // patch plane pitch in floats (= 1 mod 32; 612 pixels + 29 floats of sink)
constexpr int kUF = 36 * kBN * kCC;              // 9216 floats
constexpr int kVF = 36 * kNT * kCC;              // 4608
constexpr int kPatchF = 4 * kPlane + 4;          // 2568
constexpr int kStageF = kUF + kVF + kPatchF;     // 16392 floats = 65,568 bytes per stage
constexpr unsigned kOOB = 0x80000000u;
}  // namespace

// U4[pos][ci/4][co/64][half = (co/32)%2][k = ci%4][m = co%16][mb = (co/16)%2] = (G g G^T)[pos], g = w[:, :, ci, co]  (w HWIO)
// -- the LDS image of a (chunk, channel block) is 36 contiguous 1 KB pieces.  Computed in float64, rounded once.
__global__ __launch_bounds__(256) void wt_wino4_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t cc = (size_t)Cin * Cout;
    if (i >= cc) return;
    const int ci = (int)(i / Cout), co = (int)(i - (size_t)ci * Cout);
    double g[3][3], t[6][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) g[kh][kw] = (double)w[(size_t)(kh * 3 + kw) * cc + i];
    auto row6 = [](double a, double b, double c, double (&o)[6]) {   // G [a b c]^T
        o[0] = a / 4.0;
        o[1] = -(a + b + c) / 6.0;
        o[2] = -(a - b + c) / 6.0;
        o[3] = a / 24.0 + b / 12.0 + c / 6.0;
        o[4] = a / 24.0 - b / 12.0 + c / 6.0;
        o[5] = c;
    };
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        double o[6];
        row6(g[0][kw], g[1][kw], g[2][kw], o);
#pragma unroll
        for (int r = 0; r < 6; ++r) t[r][kw] = o[r];
    }
    const int c64 = co & 63;
    float* dst = U + ((size_t)(ci >> 2) * (Cout >> 6) + (co >> 6)) * 256 + (c64 >> 5) * 128 + (ci & 3) * 32 + (c64 & 15) * 2 + ((c64 >> 4) & 1);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double o[6];
        row6(t[r][0], t[r][1], t[r][2], o);
#pragma unroll
        for (int q = 0; q < 6; ++q) dst[(size_t)(r * 6 + q) * cc] = (float)o[q];
    }
}

int wt_wino4(const float* w, float* U, int Cin, int Cout, hipStream_t s) {
    if (Cin % kCC || Cout % kBN) return -1;
    const size_t cc = (size_t)Cin * Cout;
    hipLaunchKernelGGL(wt_wino4_kernel, dim3((unsigned)((cc + 255) / 256)), dim3(256), 0, s, w, U, Cin, Cout);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

#ifdef FS_WINO4_TRACE
// debug build only (tools/w4_trace.py): per-workgroup phase cycle counts of the last launch
__device__ long long g_wino4_trace[4096 * 8];
extern "C" int fs_debug_wino4_trace(long long* out, int n_wg) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino4_trace), sizeof(long long) * 8 * (size_t)n_wg, 0, hipMemcpyDeviceToHost);
}
#define FS_W4_NOW() ((long long)__builtin_readcyclecounter())
#endif
#ifndef FS_W4_ABL
#define FS_W4_ABL 0   /* timing experiments (results wrong): 1 no input transform, 2 no filter loads / commit, 4 no patch loads / commit, 8 no operand reads */
#endif

// 36 positions x 2 channel blocks x 4 registers = 288 accumulator registers, but the accumulator file holds 256 and the
// compiler's matrix-instruction form takes its C/D operand from that file only (asked for more, it funnels EVERY accumulator
// through one quad with v_accvgpr copies).  So positions 0..31 use the builtin (256 AGPRs), positions 32..35 an
// inline-assembly v_mfma with C/D in ordinary vector registers (legal on gfx90a+).  No software wait states are needed: an
// accumulator is next read 72 matrix instructions later, or in the epilogue behind a barrier.
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_W4_MFMA_V(accq, av, bv) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accq) : "v"(av), "v"(bv))
#else
#define FS_W4_MFMA_V(accq, av, bv) (accq) = __builtin_amdgcn_mfma_f32_16x16x4f32((av), (bv), (accq), 0, 0, 0)
#endif
// B^T x for one 6-vector (input transform, one dimension): 12 instructions
#define FS_W4_BT(d0, d1, d2, d3, d4, d5, t0, t1, t2, t3, t4, t5) \
    do {                                                         \
        const float a_ = fmaf(-4.f, d2, d4);                     \
        const float b_ = fmaf(-4.f, d1, d3);                     \
        const float c_ = d4 - d2;                                \
        const float e_ = d3 - d1;                                \
        t0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));                  \
        t1 = a_ + b_;                                            \
        t2 = a_ - b_;                                            \
        t3 = fmaf(2.f, e_, c_);                                  \
        t4 = fmaf(-2.f, e_, c_);                                 \
        t5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));                  \
    } while (0)
// A^T m for one 6-vector (output transform, one dimension): 10 instructions
#define FS_W4_AT(m0, m1, m2, m3, m4, m5, y0, y1, y2, y3) \
    do {                                                 \
        const float p_ = m1 + m2, q_ = m1 - m2;          \
        const float r_ = m3 + m4, s_ = m3 - m4;          \
        y0 = m0 + p_ + r_;                               \
        y1 = fmaf(2.f, s_, q_);                          \
        y2 = fmaf(4.f, r_, p_);                          \
        y3 = fmaf(8.f, s_, q_) + m5;                     \
    } while (0)

__global__ __launch_bounds__(256) void wino4_conv_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
#ifdef FS_WINO4_TRACE
    const long long tr_t0 = FS_W4_NOW();
    long long tr_sweep = 0, tr_bar = 0, tr_epi = 0, tr_pro = 0;
#endif
    const ConvPlan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int th = wave & 1, chh = wave >> 1;   // matrix role: tiles 16 th .., channels 32 chh .. of the item
    auto fdiv = [](int x, float inv_d) __attribute__((always_inline)) { return (int)(((float)x + 0.5f) * inv_d); };
    auto uniform_ptr = [](const float* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };

    // ---- the item list of this workgroup: item = ((n * blocks + block) * ncob + channel block) * ksplit + z.  Workgroup b runs on
    // XCD b % 8 (observed; for speed only): virtual index v = (b % 8) * (G / 8) + b / 8 gives every XCD a contiguous range of
    // items, so the channel blocks of one pixel block -- sharers of its input patch -- meet in one XCD's L2.
    const int blocks = p.tiles_y * p.tiles_x;
    const int ncob = a.Cout / kBN;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int nchunks_all = a.Cin / kCC;
    const int total_items = a.N * blocks * ncob * ks;
    const int G = (int)gridDim.x;
    const int vb = (G & 7) ? (int)blockIdx.x : (((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3));
    const int my_items = (vb < total_items) ? (total_items - 1 - vb) / G + 1 : 0;
    if (my_items == 0) return;
    const float inv_ks = 1.0f / (float)ks, inv_ncob = 1.0f / (float)ncob, inv_blocks = 1.0f / (float)blocks, inv_tx = 1.0f / (float)p.tiles_x;
    struct Item {
        int n, oy0, ox0, cob, cbeg, cend, z;
    };
    auto decode = [&](int it) __attribute__((always_inline)) {
        Item r;
        const int lin = vb + it * G;
        const int t1 = fdiv(lin, inv_ks);
        r.z = lin - t1 * ks;
        const int t2 = fdiv(t1, inv_ncob);
        r.cob = t1 - t2 * ncob;
        r.n = fdiv(t2, inv_blocks);
        const int br = t2 - r.n * blocks;
        const int byi = fdiv(br, inv_tx);
        r.oy0 = byi * kBH;
        r.ox0 = (br - byi * p.tiles_x) * kBW;
        r.cbeg = ks > 1 ? r.z * nchunks_all / ks : 0;
        r.cend = ks > 1 ? (r.z + 1) * nchunks_all / ks : nchunks_all;
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.oy0 = __builtin_amdgcn_readfirstlane(r.oy0);
        r.ox0 = __builtin_amdgcn_readfirstlane(r.ox0);
        r.cob = __builtin_amdgcn_readfirstlane(r.cob);
        r.cbeg = __builtin_amdgcn_readfirstlane(r.cbeg);
        r.cend = __builtin_amdgcn_readfirstlane(r.cend);
        r.z = __builtin_amdgcn_readfirstlane(r.z);
        return r;
    };
    // a cursor over the (item, chunk) steps of this workgroup
    struct Cursor {
        Item I;
        int it, chunk, live;
    };
    auto cursor_begin = [&]() __attribute__((always_inline)) {
        Cursor c;
        c.I = decode(0);
        c.it = 0;
        c.chunk = c.I.cbeg;
        c.live = 1;
        return c;
    };
    auto cursor_next = [&](Cursor& c) __attribute__((always_inline)) {   // returns 1 when the cursor moved to a new item
        if (!c.live) return 0;
        if (++c.chunk < c.I.cend) return 0;
        if (++c.it >= my_items) {
            c.live = 0;
            return 0;
        }
        c.I = decode(c.it);
        c.chunk = c.I.cbeg;
        return 1;
    };

    // ---- staging of the next steps: every wave does a quarter of it with the SAME straight-line instruction stream (no role
    // branches, no conditional loads: a step that does not exist is loaded through the out-of-range offset -- zeros, no memory
    // traffic -- and prepared into a stage nobody reads; conditionals around the slices turn every staged value into a phi
    // of "loaded" and "old", which the register allocator resolves with copies and a wait after every load).
    // filter: quad i of the thread = float4 e = tid + 256 i of the chunk's 2304 (position e >> 6 = (tid >> 6) + 4 i)
    float4 uv[9];
    // patch: pixel e = tid + 256 i of the 18 x 34 patch, one float4 = the chunk's 4 channels (e >= 612: plane padding)
    float4 pv[3];
    int ppy[3], ppx[3], pdst[3];
    unsigned gvo[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = tid + 256 * i;
        ppy[i] = e < kPP ? e / kPW : -4096;
        ppx[i] = e < kPP ? e - (e / kPW) * kPW : 0;
        pdst[i] = e < kPP ? e : kPP + (e - kPP) % (kPlane - kPP);
        gvo[i] = kOOB;
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * a.Cin) * 4u);
    const unsigned u_bytes = __builtin_amdgcn_readfirstlane((unsigned)(36 * a.Cin * a.Cout) * 4u);
    const unsigned u_pos4 = __builtin_amdgcn_readfirstlane((unsigned)(4 * a.Cin * a.Cout) * 4u);   // byte stride of four position planes
    const float* ub = uniform_ptr(a.w_wino4);
    const unsigned uvo = (unsigned)(((tid >> 6) * a.Cin * a.Cout + (tid & 63) * 4) * 4);
    unsigned uvo_eff = uvo;   // kOOB while the step the filter loads are for does not exist
    auto patch_offsets = [&](const Item& I, int live) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int sy = I.oy0 - 1 + ppy[i], sx = I.ox0 - 1 + ppx[i];
            const bool ok = live && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
            gvo[i] = ok ? (unsigned)((sy * a.W + sx) * a.Cin) * 4u : kOOB;
        }
    };
    auto issue_patch_one = [&](const Item& I, int chunk, int i) __attribute__((always_inline)) {
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * a.Cin);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, x_bytes, 0x00020000);
        pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, gvo[i], chunk * kCC * 4, 0));
    };
    auto commit_patch_one = [&](float* patch, int i) __attribute__((always_inline)) {
        float* d = patch + pdst[i];
        d[0] = pv[i].x;
        d[kPlane] = pv[i].y;
        d[2 * kPlane] = pv[i].z;
        d[3 * kPlane] = pv[i].w;
    };
    auto issue_filter_one = [&](const Item& I, int chunk, int i) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, u_bytes, 0x00020000);
        const unsigned so = (unsigned)((chunk * ncob + I.cob) * 1024) + (unsigned)i * u_pos4;
        uv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ur, uvo_eff, so, 0));
    };
    auto commit_filter_one = [&](float* Ul, int i) __attribute__((always_inline)) { *reinterpret_cast<float4*>(Ul + (tid + 256 * i) * 4) = uv[i]; };
    // input transform V = B^T d B of the 32 tiles x 4 channels of a chunk, split over PAIRS of lanes: wave w owns tile row w
    // (8 tiles x 4 channels); lane = (half h, channel c, tile column tx).  Half h does B^T d for columns 3h .. 3h+2 (18 reads of
    // the patch, 36 instructions), the halves exchange through the wave's own scratch T[36][32] in LDS (same wave: program
    // order, no barrier), half h does (.) B for rows 3h .. 3h+2 (18 reads of T, 36 instructions, 18 writes of V).
    const int h_t = lane >> 5, c_t = (lane >> 3) & 3, tx_t = lane & 7;
    const int tsrc = c_t * kPlane + (4 * wave) * kPW + 4 * tx_t + 3 * h_t;            // patch offset of the lane's three columns
    const int tT = 2 * kStageF + wave * (36 * 32) + (lane & 31);                      // the lane's slot in its wave's scratch (floats from smem)
    const int tdst = (wave >> 1) * 64 + c_t * 16 + (wave & 1) * 8 + tx_t;             // V offset of (tile 8 wave + tx, channel c)
    float td[18], tt[18];
    auto transform_read = [&](const float* patch, int k0, int k1) __attribute__((always_inline)) {   // k = i * 3 + jj: d[i][3h + jj]
#pragma unroll
        for (int k = k0; k < k1; ++k) td[k] = patch[tsrc + (k / 3) * kPW + (k % 3)];
    };
    auto transform_rows = [&]() __attribute__((always_inline)) {   // t[:, j] = B^T d[:, j] for the lane's three columns: tt[i * 3 + jj]
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
            FS_W4_BT(td[jj], td[3 + jj], td[6 + jj], td[9 + jj], td[12 + jj], td[15 + jj], tt[jj], tt[3 + jj], tt[6 + jj], tt[9 + jj], tt[12 + jj], tt[15 + jj]);
    };
    auto transform_xwrite = [&](int k0, int k1) __attribute__((always_inline)) {   // t[i][3h + jj] -> T[i * 6 + 3h + jj][lane & 31]
#pragma unroll
        for (int k = k0; k < k1; ++k) smem[tT + ((k / 3) * 6 + 3 * h_t + (k % 3)) * 32] = tt[k];
    };
    auto transform_xread = [&](int k0, int k1) __attribute__((always_inline)) {    // t[3h + ii][j] -> td[ii * 6 + j]
#pragma unroll
        for (int k = k0; k < k1; ++k) td[k] = smem[tT + ((3 * h_t + k / 6) * 6 + (k % 6)) * 32];
    };
    auto transform_cols = [&]() __attribute__((always_inline)) {   // V[i][:] = t[i][:] B for the lane's three rows: tt[ii * 6 + j]
#pragma unroll
        for (int ii = 0; ii < 3; ++ii)
            FS_W4_BT(td[6 * ii], td[6 * ii + 1], td[6 * ii + 2], td[6 * ii + 3], td[6 * ii + 4], td[6 * ii + 5], tt[6 * ii], tt[6 * ii + 1], tt[6 * ii + 2],
                     tt[6 * ii + 3], tt[6 * ii + 4], tt[6 * ii + 5]);
    };
    auto transform_write = [&](float* Vl, int k0, int k1) __attribute__((always_inline)) {   // position (3h + ii) * 6 + j
#pragma unroll
        for (int k = k0; k < k1; ++k) Vl[tdst + (18 * h_t + k) * (kNT * kCC)] = tt[k];
    };

    f32x4 acc[32][2];    // positions 0..31: accumulator file
    f32x4 accv[4][2];    // positions 32..35: ordinary vector registers (FS_W4_MFMA_V)
#define FS_W4_ACC(pos, mb, r) ((pos) < 32 ? acc[(pos) & 31][mb][r] : accv[(pos) & 3][mb][r])
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int pos = 0; pos < 32; ++pos)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[pos][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pos = 0; pos < 4; ++pos)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) accv[pos][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    Cursor CU = cursor_begin();   // filter cursor: step q+1 during sweep q
    Cursor CP = cursor_begin();   // patch cursor: the step whose patch loads are issued next / were issued last

    // One slice of the next steps' preparation per matrix-instruction slot (72 per sweep), identical for every wave:
    //    0-8   global loads of the 9 filter quads of step q+1
    //   10-15  LDS reads of the patch of step q+1 (the lane's 6 x 3 inputs), three per slot
    //   18     B^T d, 36 vector instructions in ONE gap (beside the fp32 matrix instruction every vector instruction costs its
    //          issue time, the first of a gap more: bunch them)
    //   20-25  the halves' exchange: LDS writes of t        36-41 LDS reads of t
    //   27-29  LDS writes of the patch of step q+2 (its loads went out during the previous sweep)
    //   31-33  global loads of the patch of step q+3
    //   44     (.) B, 36 vector instructions               46-54 LDS writes of V, two per slot
    //   56-64  LDS writes of the filter quads
    auto slice = [&](int sl, float* Un, float* Vn, const float* Pn, float* Pc) __attribute__((always_inline)) {
        if (sl < 9) {
            if (!(FS_W4_ABL & 2)) issue_filter_one(CU.I, CU.chunk, sl);
        } else if (sl >= 10 && sl < 16) {
            if (!(FS_W4_ABL & 1)) transform_read(Pn, 3 * (sl - 10), 3 * (sl - 10) + 3);
        } else if (sl == 18) {
            if (!(FS_W4_ABL & 1)) transform_rows();
        } else if (sl >= 20 && sl < 26) {
            if (!(FS_W4_ABL & 1)) transform_xwrite(3 * (sl - 20), 3 * (sl - 20) + 3);
        } else if (sl >= 27 && sl < 30) {
            if (!(FS_W4_ABL & 4)) commit_patch_one(Pc, sl - 27);
        } else if (sl >= 31 && sl < 34) {
            if (!(FS_W4_ABL & 4)) issue_patch_one(CP.I, CP.chunk, sl - 31);
        } else if (sl >= 36 && sl < 42) {
            if (!(FS_W4_ABL & 1)) transform_xread(3 * (sl - 36), 3 * (sl - 36) + 3);
        } else if (sl == 44) {
            if (!(FS_W4_ABL & 1)) transform_cols();
        } else if (sl >= 46 && sl < 55) {
            if (!(FS_W4_ABL & 1)) transform_write(Vn, 2 * (sl - 46), 2 * (sl - 46) + 2);
        } else if (sl >= 56 && sl < 65) {
            if (!(FS_W4_ABL & 2)) commit_filter_one(Un, sl - 56);
        }
    };
    auto sweep = [&](const float* Uc, const float* Vc, float* Un, float* Vn, const float* Pn, float* Pc) __attribute__((always_inline)) {
        const float* pa = Uc + chh * 128 + lane * 2;
        const float* pb = Vc + th * 64 + lane;
        f32x2 A[3];
        float B[3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            A[h] = *reinterpret_cast<const f32x2*>(pa + h * (kBN * kCC));
            B[h] = pb[h * (kNT * kCC)];
        }
        fs_static_for<0, 36>([&](auto POS) __attribute__((always_inline)) {
            constexpr int pos = decltype(POS)::value;
            constexpr int c = pos % 3, n2 = (pos + 2) % 3;
            if constexpr (pos < 32) acc[pos & 31][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c].x, B[c], acc[pos & 31][0], 0, 0, 0);
            else FS_W4_MFMA_V(accv[pos & 3][0], A[c].x, B[c]);
            __builtin_amdgcn_sched_barrier(0);
            if (pos + 2 < 36 && !(FS_W4_ABL & 8)) {   // operands two positions ahead
                A[n2] = *reinterpret_cast<const f32x2*>(pa + (pos + 2) * (kBN * kCC));
                B[n2] = pb[(pos + 2) * (kNT * kCC)];
            }
            slice(2 * pos, Un, Vn, Pn, Pc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (pos < 32) acc[pos & 31][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c].y, B[c], acc[pos & 31][1], 0, 0, 0);
            else FS_W4_MFMA_V(accv[pos & 3][1], A[c].y, B[c]);
            __builtin_amdgcn_sched_barrier(0);
            slice(2 * pos + 1, Un, Vn, Pn, Pc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- epilogue of one item.  Lane (j = lane & 15, g = lane >> 4) of wave (th, chh) holds tile 16 th + j and, per channel
    // block mb, the four channels co0 + 32 chh + 16 mb + 4 g .. + 3 of all 36 positions: output transform in registers, one
    // 16-byte store per pixel and channel block.  Pixels outside the image (edge blocks) carry the out-of-range offset: loads
    // return 0, stores are dropped by the hardware range check.
    auto epilogue = [&](const Item& I) __attribute__((always_inline)) {
        const int tl = 16 * th + (lane & 15);
        const int oy = I.oy0 + 4 * (tl >> 3), ox = I.ox0 + 4 * (tl & 7);
        const int co = I.cob * kBN + chh * 32 + 4 * (lane >> 4);
        const float* yb = a.y + ((size_t)I.n + (ks > 1 ? (size_t)I.z * a.N : 0)) * a.Ho * a.Wo * a.Cout;
        const unsigned img_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * a.Cout) * 4u);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yb)), 0, img_bytes, 0x00020000);
        const int rowp = a.Wo * a.Cout;
        const int obase = (oy * a.Wo + ox) * a.Cout + co;
        unsigned voff[16];
#pragma unroll
        for (int px = 0; px < 16; ++px) {
            const int py = px >> 2, pxx = px & 3;
            const bool ok = oy + py < a.Ho && ox + pxx < a.Wo;
            voff[px] = ok ? (unsigned)(obase + py * rowp + pxx * a.Cout) * 4u : kOOB;
        }
        const float* msn = a.mask_src ? uniform_ptr(a.mask_src + (size_t)I.n * a.Ho * a.Wo * a.Cout) : nullptr;
        const bool relu_out = a.out_relu != 0;
        float* pon = a.pool_out ? a.pool_out + (size_t)I.n * (a.Ho >> 1) * (a.Wo >> 1) * a.Cout : nullptr;
        // Per channel block: issue its 16 mask loads, transform (~600 vector instructions: covers their latency), apply + store.
        // Register budget outside the accumulator file: 64 mask + 64 outputs + 24 intermediates + 16 offsets.
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            float4 mk[16];
            if (msn) {
                const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(msn), 0, img_bytes, 0x00020000);
#pragma unroll
                for (int px = 0; px < 16; ++px) mk[px] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(mr, voff[px], mb * 64, 0));
            }
            float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) bs = *reinterpret_cast<const float4*>(a.bias + co + 16 * mb);
            __builtin_amdgcn_sched_barrier(0);
            float o[16][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s[4][6];   // A^T M: rows 0..3, columns 0..5
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    FS_W4_AT(FS_W4_ACC(q, mb, r), FS_W4_ACC(6 + q, mb, r), FS_W4_ACC(12 + q, mb, r), FS_W4_ACC(18 + q, mb, r), FS_W4_ACC(24 + q, mb, r),
                             FS_W4_ACC(30 + q, mb, r), s[0][q], s[1][q], s[2][q], s[3][q]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    FS_W4_AT(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], o[4 * i][r], o[4 * i + 1][r], o[4 * i + 2][r], o[4 * i + 3][r]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float bsv[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
            for (int px = 0; px < 16; ++px) {
                const float mv[4] = {mk[px].x, mk[px].y, mk[px].z, mk[px].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = o[px][r] + bsv[r];
                    v = relu_out ? fmaxf(v, 0.f) : v;
                    if (msn) v = mv[r] > 0.f ? v : 0.f;
                    o[px][r] = v;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fs_u32x4, make_float4(o[px][0], o[px][1], o[px][2], o[px][3])), yr, voff[px], mb * 64, 0);
            }
            if (pon) {   // 2x2/2 max-pool: the tile's four windows (tiles sit on multiples of four; Ho, Wo even)
#pragma unroll
                for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                    for (int wx = 0; wx < 2; ++wx) {
                        const int p00 = (2 * wy) * 4 + 2 * wx;
                        float m4[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) m4[r] = fmaxf(fmaxf(o[p00][r], o[p00 + 1][r]), fmaxf(o[p00 + 4][r], o[p00 + 5][r]));
                        const int qy = (oy >> 1) + 2 * 0 + wy, qx = (ox >> 1) + wx;
                        if (2 * qy < a.Ho && 2 * qx < a.Wo)
                            *reinterpret_cast<float4*>(pon + ((size_t)qy * (a.Wo >> 1) + qx) * a.Cout + co + 16 * mb) = make_float4(m4[0], m4[1], m4[2], m4[3]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        zero_acc();
    };

    // ---- prologue: step 0 complete in stage 0 (patch, V, U), the patch of step 1 in stage 1, the patch of step 2 in registers
    float* const U0 = smem;
    float* const V0 = smem + kUF;
    float* const P0 = smem + kUF + kVF;
    patch_offsets(CP.I, 1);
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_patch_one(CP.I, CP.chunk, i);
#pragma unroll
    for (int i = 0; i < 9; ++i) issue_filter_one(CU.I, CU.chunk, i);
#pragma unroll
    for (int i = 0; i < 3; ++i) commit_patch_one(P0, i);
#pragma unroll
    for (int i = 0; i < 9; ++i) commit_filter_one(U0, i);
    if (cursor_next(CP) || !CP.live) patch_offsets(CP.I, CP.live);
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_patch_one(CP.I, CP.chunk, i);
#pragma unroll
    for (int i = 0; i < 3; ++i) commit_patch_one(P0 + kStageF, i);
    if (cursor_next(CP) || !CP.live) patch_offsets(CP.I, CP.live);
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_patch_one(CP.I, CP.chunk, i);
    cursor_next(CU);   // the filter cursor now points at step 1
    __syncthreads();
    transform_read(P0, 0, 18);
    transform_rows();
    transform_xwrite(0, 18);
    __builtin_amdgcn_wave_barrier();   // (the halves of a wave exchange through LDS: lockstep on the GPU, a fiber hand-off in the emulator)
    transform_xread(0, 18);
    transform_cols();
    transform_write(V0, 0, 18);
    __syncthreads();
    FS_WAIT_VMEM();
#ifdef FS_WINO4_TRACE
    tr_pro = FS_W4_NOW() - tr_t0;
#endif

    // ---- the flat pipeline over (item, chunk) steps: step q multiplies out of stage q & 1 while step q+1 is prepared into the
    // other stage (U, V), the patch of step q+2 lands in this stage's patch area and the patch loads of step q+3 go out
    int q = 0;
    for (int it = 0; it < my_items; ++it) {
        const Item cur_it = decode(it);
        for (int chunk = cur_it.cbeg; chunk < cur_it.cend; ++chunk, ++q) {
            uvo_eff = CU.live ? uvo : kOOB;
            if (cursor_next(CP) || !CP.live) patch_offsets(CP.I, CP.live);   // (scalar work; the offsets change once per item)
#ifdef FS_WINO4_TRACE
            const long long q0 = FS_W4_NOW();
#endif
            const int o0 = (q & 1) ? kStageF : 0, o1 = kStageF - o0;
            sweep(smem + o0, smem + o0 + kUF, smem + o1, smem + o1 + kUF, smem + o1 + kUF + kVF, smem + o0 + kUF + kVF);
            cursor_next(CU);
#ifdef FS_WINO4_TRACE
            const long long q1 = FS_W4_NOW();
#endif
            FS_LDS_BARRIER();
#ifdef FS_WINO4_TRACE
            const long long q2 = FS_W4_NOW();
            tr_sweep += q1 - q0;
            tr_bar += q2 - q1;
#endif
        }
#ifdef FS_WINO4_TRACE
        const long long e0 = FS_W4_NOW();
#endif
#ifdef FS_W4_NOEPI
        { float sacc = 0.f;
#pragma unroll
          for (int pos = 0; pos < 36; ++pos) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
              for (int r = 0; r < 4; ++r) sacc += FS_W4_ACC(pos, mb, r); } }
          a.y[tid] = sacc; zero_acc(); }
#else
        epilogue(cur_it);
#endif
#ifdef FS_WINO4_TRACE
        tr_epi += FS_W4_NOW() - e0;
#endif
    }
#ifdef FS_WINO4_TRACE
    if (tid == 0 && blockIdx.x < 4096) {
        long long* t = g_wino4_trace + (size_t)blockIdx.x * 8;
        t[0] = tr_t0;
        t[1] = tr_pro;
        t[2] = tr_sweep;
        t[3] = tr_bar;
        t[4] = tr_epi;
        t[5] = q;
        t[6] = FS_W4_NOW();
        t[7] = my_items;
    }
#endif
}

bool wino4_eligible(const ConvArgs& a) {
    // the VGG16 form only: 3x3 stride 1 SAME, plain source, bias / ReLU / pool or consumer mask in the epilogue
    const bool pad_ok = a.pad_t == 1 && a.pad_l == 1 && a.Ho == a.H && a.Wo == a.W;
    return a.w_wino4 && tune_int("FS_WINO_V", 4) >= 4 && a.KH == 3 && a.KW == 3 && a.stride == 1 && pad_ok && a.src_mode == SRC_PLAIN &&
           a.Cin % kCC == 0 && a.Cout % kBN == 0 && !a.shuffle && !a.add_src && !a.in_a && !a.stats && !a.route_src && a.w_nstride == 0 &&
           a.dil_x <= 1 && (!a.pool_out || (!(a.Ho & 1) && !(a.Wo & 1)));
}

void wino4_plan(const ConvArgs& a, ConvPlan* out) {
    ConvPlan p{};
    p.variant = 10;
    p.BN = kBN;
    p.CC = kCC;
    p.TH = kBH;
    p.TW = kBW;
    p.tiles_y = cdiv(a.Ho, kBH);
    p.tiles_x = cdiv(a.Wo, kBW);
    p.lds_bytes = 4 * (2 * kStageF + 4 * 36 * 32);   // two stages + the waves' transform exchange scratch
    p.ksplit = 1;
    const long items = (long)a.N * p.tiles_y * p.tiles_x * (a.Cout / kBN);
    const int nchunks = a.Cin / kCC;
    const int max_ks = tune_int("FS_WINO_KSPLIT", 4);
    if (a.split_ws && !a.pool_out) {
        int ks = 1;
        while (ks < max_ks && items * ks < 256 && nchunks / (ks * 2) >= 16 && (size_t)(ks * 2) * a.N * a.Ho * a.Wo * a.Cout <= a.split_ws_floats) ks *= 2;
        p.ksplit = ks;
    }
    *out = p;
}

int wino4_launch(const ConvArgs& a, hipStream_t s) {
    const ConvPlan& p = a.p;
    const long items = (long)a.N * p.tiles_y * p.tiles_x * (a.Cout / kBN) * (p.ksplit > 1 ? p.ksplit : 1);
    const int wgs = tune_int("FS_WINO4_WGS", 256);
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(wino4_conv_kernel));
    const long grid = items < wgs ? items : wgs;
    hipLaunchKernelGGL(wino4_conv_kernel, dim3((unsigned)grid), dim3(256), (size_t)p.lds_bytes, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
