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
//   * the staging of the next chunk is threaded through the 72 matrix-instruction slots of the sweep as straight-line slices,
//     the SAME instruction stream in every wave (no role branches, no conditional loads): 9 filter quads global -> registers
//     -> LDS, the input transform of one 6x6 block per lane PAIR (18 LDS reads, 72 vector instructions bunched into two gaps,
//     9 lane-half swaps, 18 LDS writes), 3 patch quads of the chunk after next -- about 1.2k cycles beside the 2.3k of matrix
//     instructions, because beside the fp32 matrix instruction every vector-ALU / vector-memory instruction costs its issue
//     time (tools/mfma_overlap.hip);
//   * epilogue per item: bias + ReLU (+ the 2x2 max-pool of the tile's four windows) for the forward, the consumer's
//     ReLU mask for the input gradients; split-K (raw partials) where the launch cannot fill the chip.
#include "fs_wino4.h"

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
#ifndef FS_W4_NA
#define FS_W4_NA 32
#endif
#ifndef FS_W4_MASK_EARLY
#define FS_W4_MASK_EARLY 8
#endif
constexpr int kMaskEarly = FS_W4_MASK_EARLY;      // consumer-mask loads issued in front of the output transform (the rest behind it)
constexpr int kNA = FS_W4_NA;                    // positions whose accumulators live in the accumulator file (the other 36 - kNA: vector registers)
}  // namespace

// U4[pos][ci/4][co/64][half = (co/32)%2][k = ci%4][m = co%16][mb = (co/16)%2] = (G g G^T)[pos], g = w[:, :, ci, co]  (w HWIO)
// -- the LDS image of a (chunk, channel block) is 36 contiguous 1 KB pieces.  Computed in float64, rounded once.
__global__ __launch_bounds__(256) void wt_wino4_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t cc = (size_t)Cin * Cout;
    if (i >= cc) return;
    const int ci = (int)(i / Cout), co = (int)(i - (size_t)ci * Cout);
    double o[36];
    wino4_filter_transform(w, cc, i, o);
    const int c64 = co & 63;
    float* dst = U + ((size_t)(ci >> 2) * (Cout >> 6) + (co >> 6)) * 256 + (c64 >> 5) * 128 + (ci & 3) * 32 + (c64 & 15) * 2 + ((c64 >> 4) & 1);
#pragma unroll
    for (int k = 0; k < 36; ++k) dst[(size_t)k * cc] = (float)o[k];
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

// EPI: the epilogue form, a compile-time constant (a run-time choice costs a select per stored element): 0 raw (split-K partials,
// the input gradient in front of a max-pool), 1 bias + ReLU (+ the fused 2x2 max-pool) -- the forward convs, 2 the consumer's
// ReLU mask -- the input gradients
template <int EPI>
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
    int pdst[3];
    unsigned gvo[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = tid + 256 * i;
        pdst[i] = e < kPP ? e : kPP + (e - kPP) % (kPlane - kPP);
        gvo[i] = kOOB;
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * a.Cin) * 4u);
    const unsigned u_bytes = __builtin_amdgcn_readfirstlane((unsigned)(36 * a.Cin * a.Cout) * 4u);
    const unsigned u_pos4 = __builtin_amdgcn_readfirstlane((unsigned)(4 * a.Cin * a.Cout) * 4u);   // byte stride of four position planes
    const float* ub = uniform_ptr(a.w_wino4);
    const unsigned uvo = (unsigned)(((tid >> 6) * a.Cin * a.Cout + (tid & 63) * 4) * 4);
    unsigned uvo_eff = uvo;   // kOOB while the step the filter loads are for does not exist
    auto patch_offsets = [&](const Item& I, int live) __attribute__((always_inline)) {   // (once per item: the pixel coordinates are recomputed, not kept)
        int t_ = tid;
        FS_W4_PIN(t_);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = t_ + 256 * i;
            const int py = (int)(((float)e + 0.5f) * (1.0f / (float)kPW)), px = e - py * kPW;
            const int sy = I.oy0 - 1 + py, sx = I.ox0 - 1 + px;
            const bool ok = live && e < kPP && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
            gvo[i] = ok ? (unsigned)((sy * a.W + sx) * a.Cin) * 4u : kOOB;
        }
    };
    auto issue_patch_one = [&](const Item& I, int chunk, int i) __attribute__((always_inline)) {
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * a.Cin);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, x_bytes, 0x00020000);
        pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, gvo[i], chunk * kCC * 4, 0));
    };
    auto commit_patch_one = [&](int a_pc, int i) __attribute__((always_inline)) {   // a_pc: address of pixel pdst[i] in plane 0
        FS_W4_LDS(float, a_pc) = pv[i].x;
        FS_W4_LDS(float, a_pc + kPlane * 4) = pv[i].y;
        FS_W4_LDS(float, a_pc + 2 * kPlane * 4) = pv[i].z;
        FS_W4_LDS(float, a_pc + 3 * kPlane * 4) = pv[i].w;
    };
    auto issue_filter_one = [&](const Item& I, int chunk, int i) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, u_bytes, 0x00020000);
        const unsigned so = (unsigned)((chunk * ncob + I.cob) * 1024) + (unsigned)i * u_pos4;
        uv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ur, uvo_eff, so, 0));
    };
    auto commit_filter_one = [&](int a_un, int i) __attribute__((always_inline)) { FS_W4_LDS(float4, a_un + i * 4096) = uv[i]; };   // a_un: address of float4 `tid`
    // input transform V = B^T d B of the 32 tiles x 4 channels of a chunk, split over PAIRS of lanes: wave w owns tile row w
    // (8 tiles x 4 channels); lane = (half h, channel c, tile column tx).  Half h does B^T d for columns 3h .. 3h+2 (18 reads of
    // the patch, 36 instructions); the halves trade nine registers each (v_permlane32_swap: rows 0..2 of the upper half's
    // columns against rows 3..5 of the lower half's) and half h does (.) B for rows 3h .. 3h+2 (36 instructions, 18 writes of V).
    const int h_t = lane >> 5, c_t = (lane >> 3) & 3, tx_t = lane & 7;
    const int tsrc = c_t * kPlane + (4 * wave) * kPW + 4 * tx_t + 3 * h_t;            // patch offset of the lane's three columns
    const int tdst = (wave >> 1) * 64 + c_t * 16 + (wave & 1) * 8 + tx_t;             // V offset of (tile 8 wave + tx, channel c)
    float td[18], tt[18];
    auto transform_read = [&](int a_pn, int k0, int k1) __attribute__((always_inline)) {   // k = i * 3 + jj: d[i][3h + jj]; a_pn: address of d[0][3h]
#pragma unroll
        for (int k = k0; k < k1; ++k) td[k] = FS_W4_LDS(float, a_pn + ((k / 3) * kPW + (k % 3)) * 4);
    };
    auto transform_rows = [&]() __attribute__((always_inline)) {   // t[:, j] = B^T d[:, j] for the lane's three columns: tt[i * 3 + jj]
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
            FS_W4_BT(td[jj], td[3 + jj], td[6 + jj], td[9 + jj], td[12 + jj], td[15 + jj], tt[jj], tt[3 + jj], tt[6 + jj], tt[9 + jj], tt[12 + jj], tt[15 + jj]);
    };
    auto transform_swap = [&]() __attribute__((always_inline)) {   // -> td[ii * 6 + j] = t[3h + ii][j]
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                float y = tt[i * 3 + jj], x = tt[(i + 3) * 3 + jj];   // lower half keeps y, gets the upper half's y into x; upper half keeps x, gets the lower half's x into y
                FS_W4_SWAP(y, x);
                td[6 * i + jj] = y;
                td[6 * i + 3 + jj] = x;
            }
    };
    auto transform_cols = [&]() __attribute__((always_inline)) {   // V[i][:] = t[i][:] B for the lane's three rows: tt[ii * 6 + j]
#pragma unroll
        for (int ii = 0; ii < 3; ++ii)
            FS_W4_BT(td[6 * ii], td[6 * ii + 1], td[6 * ii + 2], td[6 * ii + 3], td[6 * ii + 4], td[6 * ii + 5], tt[6 * ii], tt[6 * ii + 1], tt[6 * ii + 2],
                     tt[6 * ii + 3], tt[6 * ii + 4], tt[6 * ii + 5]);
    };
    auto transform_write = [&](int a_vn, int k0, int k1) __attribute__((always_inline)) {   // position (3h + ii) * 6 + j; a_vn: address of position 18 h
#pragma unroll
        for (int k = k0; k < k1; ++k) FS_W4_LDS(float, a_vn + k * (kNT * kCC * 4)) = tt[k];
    };

    f32x4 acc[kNA][2];         // positions 0 .. kNA-1: accumulator file
    f32x4 accv[36 - kNA][2];   // the rest: ordinary vector registers (FS_W4_MFMA_V)
// (accumulator-file elements leave through a volatile v_accvgpr_read exactly where the output transform consumes them: left to
// itself the scheduler hoists hundreds of these reads to the top of the epilogue, and the register allocator answers by
// spilling the sweep's loop invariants to scratch memory -- reloaded before every sweep behind `s_waitcnt vmcnt(0)`)
#ifdef FS_W4_PLAIN_ACC
#define FS_W4_ACC(pos, mb, r) ((pos) < kNA ? acc[(pos) < kNA ? (pos) : 0][mb][r] : accv[(pos) >= kNA ? (pos) - kNA : 0][mb][r])
#else
#define FS_W4_ACC(pos, mb, r) ((pos) < kNA ? FS_ACC_READ(acc[(pos) < kNA ? (pos) : 0][mb][r]) : accv[(pos) >= kNA ? (pos) - kNA : 0][mb][r])
#endif
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int pos = 0; pos < kNA; ++pos)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[pos][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pos = 0; pos < 36 - kNA; ++pos)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) accv[pos][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    FS_W4_MFMA_SETTLE();

    Cursor CU = cursor_begin();   // filter cursor: step q+1 during sweep q
    Cursor CP = cursor_begin();   // patch cursor: the step whose patch loads are issued next / were issued last

    // One slice of the next steps' preparation per matrix-instruction slot (72 per sweep), identical for every wave:
    //    0-8   global loads of the 9 filter quads of step q+1
    //   10-15  LDS reads of the patch of step q+1 (the lane's 6 x 3 inputs), three per slot
    //   18     B^T d, 36 vector instructions in ONE gap (beside the fp32 matrix instruction every vector instruction costs its
    //          issue time, the first of a gap more: bunch them)
    //   22     the halves' exchange (9 swaps) + (.) B (36 vector instructions)
    //   24-32  LDS writes of V, two per slot
    //   34-36  LDS writes of the patch of step q+2 (its loads went out during the previous sweep)
    //   38-40  global loads of the patch of step q+3
    //   56-64  LDS writes of the filter quads
    struct Addr {
        int pa, pb;         // matrix operands of the current stage: U (+ half, lane), V (+ half, lane)
        int un, vn, pn;     // next stage: the thread's filter quad 0, its V position 18 h, its patch block d[0][3h]
        int pc[3];          // this stage's patch area: the thread's three pixels
    };
    auto stage_addrs = [&](int o0, int o1) __attribute__((always_inline)) {   // o0 / o1: float offsets of the current / the other stage
        Addr A;
        A.pa = FS_W4_ADDR(smem + o0 + chh * 128 + lane * 2);
        A.pb = FS_W4_ADDR(smem + o0 + kUF + th * 64 + lane);
        A.un = FS_W4_ADDR(smem + o1 + tid * 4);
        A.vn = FS_W4_ADDR(smem + o1 + kUF + tdst + 18 * h_t * (kNT * kCC));
        A.pn = FS_W4_ADDR(smem + o1 + kUF + kVF + tsrc);
#pragma unroll
        for (int i = 0; i < 3; ++i) A.pc[i] = FS_W4_ADDR(smem + o0 + kUF + kVF + pdst[i]);
        FS_W4_PIN(A.pa);
        FS_W4_PIN(A.pb);
        FS_W4_PIN(A.un);
        FS_W4_PIN(A.vn);
        FS_W4_PIN(A.pn);
#pragma unroll
        for (int i = 0; i < 3; ++i) FS_W4_PIN(A.pc[i]);
        return A;
    };
    auto slice = [&](int sl, const Addr& AD) __attribute__((always_inline)) {
        if (sl < 9) {
            if (!(FS_W4_ABL & 2)) issue_filter_one(CU.I, CU.chunk, sl);
        } else if (sl >= 10 && sl < 16) {
            if (!(FS_W4_ABL & 1)) transform_read(AD.pn, 3 * (sl - 10), 3 * (sl - 10) + 3);
        } else if (sl == 18) {
            if (!(FS_W4_ABL & 1)) transform_rows();
        } else if (sl == 22) {
            if (!(FS_W4_ABL & 1)) {
                transform_swap();
                transform_cols();
            }
        } else if (sl >= 24 && sl < 33) {
            if (!(FS_W4_ABL & 1)) transform_write(AD.vn, 2 * (sl - 24), 2 * (sl - 24) + 2);
        } else if (sl >= 34 && sl < 37) {
            if (!(FS_W4_ABL & 4)) commit_patch_one(AD.pc[sl - 34], sl - 34);
        } else if (sl >= 38 && sl < 41) {
            if (!(FS_W4_ABL & 4)) issue_patch_one(CP.I, CP.chunk, sl - 38);
        } else if (sl >= 56 && sl < 65) {
            if (!(FS_W4_ABL & 2)) commit_filter_one(AD.un, sl - 56);
        }
    };
    auto sweep = [&](const Addr& AD) __attribute__((always_inline)) {
        f32x2 A[3];
        float B[3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            A[h] = FS_W4_LDS(f32x2, AD.pa + h * (kBN * kCC * 4));
            B[h] = FS_W4_LDS(float, AD.pb + h * (kNT * kCC * 4));
        }
        fs_static_for<0, 36>([&](auto POS) __attribute__((always_inline)) {
            constexpr int pos = decltype(POS)::value;
            constexpr int c = pos % 3, n2 = (pos + 2) % 3;
            if constexpr (pos < kNA) acc[pos < kNA ? pos : 0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c].x, B[c], acc[pos < kNA ? pos : 0][0], 0, 0, 0);
            else FS_W4_MFMA_V(accv[pos >= kNA ? pos - kNA : 0][0], A[c].x, B[c]);
            __builtin_amdgcn_sched_barrier(0);
            if (pos + 2 < 36 && !(FS_W4_ABL & 8)) {   // operands two positions ahead
                A[n2] = FS_W4_LDS(f32x2, AD.pa + (pos + 2) * (kBN * kCC * 4));
                B[n2] = FS_W4_LDS(float, AD.pb + (pos + 2) * (kNT * kCC * 4));
            }
            slice(2 * pos, AD);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (pos < kNA) acc[pos < kNA ? pos : 0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c].y, B[c], acc[pos < kNA ? pos : 0][1], 0, 0, 0);
            else FS_W4_MFMA_V(accv[pos >= kNA ? pos - kNA : 0][1], A[c].y, B[c]);
            __builtin_amdgcn_sched_barrier(0);
            slice(2 * pos + 1, AD);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- epilogue of one item.  Lane (j = lane & 15, g = lane >> 4) of wave (th, chh) holds tile 16 th + j and, per channel
    // block mb, the four channels co0 + 32 chh + 16 mb + 4 g .. + 3 of all 36 positions: output transform in registers, one
    // 16-byte store per pixel and channel block.  Pixels outside the image (edge blocks) carry the out-of-range offset: loads
    // return 0, stores are dropped by the hardware range check.
    // FULL: the item's 16 x 32-pixel block lies inside the image (every VGG16 layer of a 256 x 256 batch): one lane base register,
    // the pixel offsets are scalars.  Edge blocks swap the base for the out-of-range offset per pixel.
    auto epilogue_body = [&](auto FULLT, const Item& I) __attribute__((always_inline)) {
        constexpr bool full = decltype(FULLT)::value;
        // (an opaque copy of the lane index: everything derived from it is computed HERE, once per item -- hoisted out of the item
        // loop such values sit in registers across every sweep and push the staging state into scratch memory)
        int ln = lane;
        FS_W4_PIN(ln);
        const int tl = 16 * th + (ln & 15);
        const int oy = I.oy0 + 4 * (tl >> 3), ox = I.ox0 + 4 * (tl & 7);
        const int co = I.cob * kBN + chh * 32 + 4 * (ln >> 4);
        const float* yb = a.y + ((size_t)I.n + (ks > 1 ? (size_t)I.z * a.N : 0)) * a.Ho * a.Wo * a.Cout;
        const unsigned img_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * a.Cout) * 4u);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yb)), 0, img_bytes, 0x00020000);
        const unsigned rowp4 = __builtin_amdgcn_readfirstlane((unsigned)(a.Wo * a.Cout) * 4u), col4 = __builtin_amdgcn_readfirstlane((unsigned)a.Cout * 4u);
        const unsigned obase = (unsigned)((oy * a.Wo + ox) * a.Cout + co) * 4u;
        const int ry = a.Ho - oy, cx = a.Wo - ox;   // valid rows / columns of the lane's tile (edge blocks)
        auto voff = [&](int px) __attribute__((always_inline)) { return (full || ((px >> 2) < ry && (px & 3) < cx)) ? obase : kOOB; };
        auto soff = [&](int px, int mb) __attribute__((always_inline)) { return (unsigned)(px >> 2) * rowp4 + (unsigned)(px & 3) * col4 + (unsigned)mb * 64u; };
        const float* msn = EPI == 2 ? uniform_ptr(a.mask_src + (size_t)I.n * a.Ho * a.Wo * a.Cout) : nullptr;
        const bool pool = EPI == 1 && a.pool_out != nullptr;
        auto relu1 = [](float x) __attribute__((always_inline)) {   // ONE v_max_f32 (fmaxf comes with a canonicalising second instruction)
#if defined(__HIP_DEVICE_COMPILE__)
            float r;
            asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
            return r;
#else
            return x > 0.f ? x : 0.f;
#endif
        };
        // Per channel block: mask loads of pixels 0..7 | output transform (~600 vector instructions: covers their latency) | mask
        // loads of pixels 8..15 | apply + store 0..7 | apply + store 8..15.  Register peak outside the accumulator file: 32 mask
        // + 64 outputs + 30 intermediates during the transform, 64 + 64 behind it.
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            float4 mk[16];
            const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EPI == 2 ? msn : yb), 0, img_bytes, 0x00020000);
            if (EPI == 2) {
#pragma unroll
                for (int px = 0; px < kMaskEarly; ++px) mk[px] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(mr, voff(px), soff(px, mb), 0));
            }
            float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (EPI == 1 && a.bias) bs = *reinterpret_cast<const float4*>(a.bias + co + 16 * mb);
            __builtin_amdgcn_sched_barrier(0);
            float o[16][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s[4][6];   // A^T M: rows 0..3, columns 0..5
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const float m0 = FS_W4_ACC(q, mb, r), m1 = FS_W4_ACC(6 + q, mb, r), m2 = FS_W4_ACC(12 + q, mb, r), m3 = FS_W4_ACC(18 + q, mb, r),
                                m4 = FS_W4_ACC(24 + q, mb, r), m5 = FS_W4_ACC(30 + q, mb, r);   // (each element read ONCE: the reads are volatile)
                    FS_W4_AT(m0, m1, m2, m3, m4, m5, s[0][q], s[1][q], s[2][q], s[3][q]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    FS_W4_AT(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], o[4 * i][r], o[4 * i + 1][r], o[4 * i + 2][r], o[4 * i + 3][r]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (EPI == 2) {
#pragma unroll
                for (int px = kMaskEarly; px < 16; ++px) mk[px] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(mr, voff(px), soff(px, mb), 0));
            }
            __builtin_amdgcn_sched_barrier(0);
            const float bsv[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
            for (int px = 0; px < 16; ++px) {
                float mv[4] = {1.f, 1.f, 1.f, 1.f};
                if (EPI == 2) {   // (mk is loaded in this form only: no read of an indeterminate value in the others)
                    mv[0] = mk[px].x;
                    mv[1] = mk[px].y;
                    mv[2] = mk[px].z;
                    mv[3] = mk[px].w;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = o[px][r];
                    if (EPI == 1) v = a.out_relu ? relu1(v + bsv[r]) : v + bsv[r];
                    if (EPI == 2) v = mv[r] > 0.f ? v : 0.f;
                    o[px][r] = v;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fs_u32x4, make_float4(o[px][0], o[px][1], o[px][2], o[px][3])), yr, voff(px),
                                                       soff(px, mb), 0);
            }
            if (pool) {   // 2x2/2 max-pool: the tile's four windows (tiles sit on multiples of four; Ho, Wo even)
                const float* pb_ = a.pool_out + (size_t)I.n * (a.Ho >> 1) * (a.Wo >> 1) * a.Cout;
                const unsigned pimg = __builtin_amdgcn_readfirstlane((unsigned)((a.Ho >> 1) * (a.Wo >> 1) * a.Cout) * 4u);
                const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(pb_)), 0, pimg, 0x00020000);
                const unsigned pbase = (unsigned)(((oy >> 1) * (a.Wo >> 1) + (ox >> 1)) * a.Cout + co) * 4u;
                const unsigned prow4 = __builtin_amdgcn_readfirstlane((unsigned)((a.Wo >> 1) * a.Cout) * 4u);
#pragma unroll
                for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                    for (int wx = 0; wx < 2; ++wx) {
                        const int p00 = (2 * wy) * 4 + 2 * wx;
                        float m4[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) m4[r] = fmaxf(fmaxf(o[p00][r], o[p00 + 1][r]), fmaxf(o[p00 + 4][r], o[p00 + 5][r]));
                        const bool okp = full || (2 * wy < ry && 2 * wx < cx);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fs_u32x4, make_float4(m4[0], m4[1], m4[2], m4[3])), pr, okp ? pbase : kOOB,
                                                               (unsigned)wy * prow4 + (unsigned)wx * col4 + (unsigned)mb * 64u, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto epilogue = [&](const Item& I) __attribute__((always_inline)) {
        FS_W4_MFMA_SETTLE();   // (the last inline-assembly matrix instructions of the item have written their accumulators)
        if (I.oy0 + kBH <= a.Ho && I.ox0 + kBW <= a.Wo)
            epilogue_body(std::true_type{}, I);
        else
            epilogue_body(std::false_type{}, I);
        zero_acc();
        FS_W4_MFMA_SETTLE();   // (the zeroed accumulators are the next sweep's SrcC)
        // Nothing of the epilogue may still be in flight at the loop header: loads and stores share ONE counter and complete out
        // of order with each other, so with stores pending the compiler places `s_waitcnt vmcnt(0)` -- every filter load just
        // issued included -- in front of the sweep's first use of a loaded register instead of an exact count.  The drain hides
        // behind the 288 zeroing moves.
#ifndef FS_W4_X1
        FS_WAIT_VMEM();
#endif
    };

    // ---- prologue: step 0 complete in stage 0 (patch, V, U), the patch of step 1 in stage 1, the patch of step 2 in registers
    const Addr AP0 = stage_addrs(kStageF, 0), AP1 = stage_addrs(0, kStageF);   // "next stage" = stage 0 / stage 1
    patch_offsets(CP.I, 1);
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_patch_one(CP.I, CP.chunk, i);
#pragma unroll
    for (int i = 0; i < 9; ++i) issue_filter_one(CU.I, CU.chunk, i);
#pragma unroll
    for (int i = 0; i < 3; ++i) commit_patch_one(AP1.pc[i], i);   // (AP1's current stage is stage 0)
#pragma unroll
    for (int i = 0; i < 9; ++i) commit_filter_one(AP0.un, i);
    if (cursor_next(CP) || !CP.live) patch_offsets(CP.I, CP.live);
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_patch_one(CP.I, CP.chunk, i);
#pragma unroll
    for (int i = 0; i < 3; ++i) commit_patch_one(AP0.pc[i], i);   // stage 1's patch area
    if (cursor_next(CP) || !CP.live) patch_offsets(CP.I, CP.live);
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_patch_one(CP.I, CP.chunk, i);
    cursor_next(CU);   // the filter cursor now points at step 1
    __syncthreads();
    transform_read(AP0.pn, 0, 18);
    transform_rows();
    transform_swap();
    transform_cols();
    transform_write(AP0.vn, 0, 18);
    __syncthreads();
    FS_WAIT_VMEM();
#ifdef FS_WINO4_TRACE
    tr_pro = FS_W4_NOW() - tr_t0;
#endif

    // Optional stagger (FS_WINO4_STAGGER = s > 0, launches with >= 8 items per workgroup): workgroups of equal item lists run in
    // lockstep -- every epilogue's 33 MB of stores (and the mask loads of the input gradients) hit HBM at the same moment, every
    // sweep phase has none.  Workgroup v starts (v mod 8) * s / 64 of an item period late.
    if (a.p.skew > 0 && my_items >= 8) {
        const int period = (nchunks_all / ks) * 3600;   // cycles of one item, roughly
        const long long wait = (long long)(vb & 7) * period * a.p.skew / 512;
        const long long t_begin = __builtin_readcyclecounter();
        while ((long long)__builtin_readcyclecounter() - t_begin < wait) __builtin_amdgcn_s_sleep(32);
    }

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
            const Addr AD = stage_addrs(o0, o1);
            __builtin_amdgcn_sched_barrier(0);
            sweep(AD);
#if defined(FS_W4_X2) && defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#endif
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
    // byte offsets inside one sample are 32-bit with the top bit reserved for "out of range"; the filter planes likewise
    const bool fits = (double)a.H * a.W * (a.Cin > a.Cout ? a.Cin : a.Cout) * 4.0 < 2147483648.0 && 36.0 * a.Cin * a.Cout * 4.0 < 2147483648.0;
    return fits && a.w_wino4 && wino_gen().f4_lds() && a.KH == 3 && a.KW == 3 && a.stride == 1 && pad_ok && a.src_mode == SRC_PLAIN &&
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
    p.lds_bytes = 4 * 2 * kStageF;
    p.ksplit = 1;
    p.skew = tune_int("FS_WINO4_STAGGER", 0);
    const long items = (long)a.N * p.tiles_y * p.tiles_x * (a.Cout / kBN);
    const int nchunks = a.Cin / kCC;
    const int max_ks = tune_int("FS_WINO_KSPLIT", 4);
    if (a.split_ws && !a.pool_out) {
        int ks = 1;
        const int min_steps = tune_int("FS_WINO4_KSPLIT_MINSTEPS", 16);   // chunks (steps) a split item must keep
        while (ks < max_ks && items * ks < 256 && nchunks / (ks * 2) >= min_steps && (size_t)(ks * 2) * a.N * a.Ho * a.Wo * a.Cout <= a.split_ws_floats) ks *= 2;
        p.ksplit = ks;
    }
    *out = p;
}

int wino4_launch(const ConvArgs& a, hipStream_t s) {
    const ConvPlan& p = a.p;
    const long items = (long)a.N * p.tiles_y * p.tiles_x * (a.Cout / kBN) * (p.ksplit > 1 ? p.ksplit : 1);
    const int wgs = tune_int("FS_WINO4_WGS", 256);
    const long grid = items < wgs ? items : wgs;
    static BigLds lds_attr[3];
    const int epi = p.ksplit > 1 ? 0 : (a.mask_src ? 2 : ((a.bias || a.out_relu || a.pool_out) ? 1 : 0));
    if (epi == 0) {
        lds_attr[0].ensure(reinterpret_cast<const void*>(wino4_conv_kernel<0>));
        hipLaunchKernelGGL(wino4_conv_kernel<0>, dim3((unsigned)grid), dim3(256), (size_t)p.lds_bytes, s, a);
    } else if (epi == 1) {
        lds_attr[1].ensure(reinterpret_cast<const void*>(wino4_conv_kernel<1>));
        hipLaunchKernelGGL(wino4_conv_kernel<1>, dim3((unsigned)grid), dim3(256), (size_t)p.lds_bytes, s, a);
    } else {
        lds_attr[2].ensure(reinterpret_cast<const void*>(wino4_conv_kernel<2>));
        hipLaunchKernelGGL(wino4_conv_kernel<2>, dim3((unsigned)grid), dim3(256), (size_t)p.lds_bytes, s, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
