// Winograd F(4x4, 3x3) for the transform net's residual convolutions (reference im_transf_net.py:250-276: ten 3x3 64 -> 64
// VALID convs, each behind an instance norm; in the training step also their input gradients, 3x3 'full' convs of dz) -- the
// 16-TILE sibling of fs_wino4.hip.  Same algorithm (36 products per 4x4 outputs, fp32 matrix cores, one wave per SIMD,
// persistent workgroups); what differs is forced by the shapes: a 720p frame is 180 x 320 pixels at this depth = 143 of
// fs_wino4's 16 x 32-pixel items for 256 CUs, a batch-4 training step 72.  With 16 x 16-pixel items they are 240 and 100-144.
//
//   * item = 16 tiles (4 x 4 tiles of 4 x 4 pixels) x 64 output channels x 36 positions: wave w owns the 16 channels 16 w .. of
//     all 16 tiles and 36 positions = 144 accumulator registers (channels on the matrix instruction's row side, so a lane holds
//     four consecutive channels of ONE tile at every position: output transform in registers, 16-byte stores);
//   * input channels in chunks of 8 = two matrix instructions per position: 72 per step and wave, the shape of fs_wino4's sweep;
//   * with half the tiles per item the filter traffic per product doubles, and through LDS it would cost as much as the products
//     of a chunk.  But here every wave needs a DIFFERENT 16-channel block of the filter, so nothing is shared between waves:
//     the A operand goes global -> registers directly, pre-arranged by wt_wino4t in the matrix instruction's own lane layout
//     (lane = k * 16 + m holds 4 consecutive slots per 16-byte load: 18 loads per step and lane, no LDS write, no LDS read).
//     A filter quad is reloaded for the next step right behind the last matrix instruction that reads it (one register set);
//   * LDS holds only the transformed input V (72 blocks of [k][tile], 72 floats apart: conflict-free for the transform's
//     writes and the operand reads) and the raw 18 x 18 x 8 patch (channel-planar, row pitch 20, plane pitch 385) -- 33 KB per stage;
//   * input transform of the 16 tiles x 8 channels of a chunk on lane pairs with v_permlane32_swap, as in fs_wino4.hip;
//   * on load: the producer's instance norm + ReLU (AFF; padding 0 only); epilogue forms: raw, raw + per-item instance-norm
//     partials {mean, M2, count} (the forward), + the residual gradient added in the interior (the first conv of a block, backward).
#include "fs_wino4.h"

#include <cstdlib>
#include <type_traits>

namespace fs {

namespace {
constexpr int kBH = 16, kBW = 16;                // output pixels per item
constexpr int kNT = 16;                          // tiles per item
constexpr int kPH = 18, kPW = 18;                // input patch
constexpr int kPR = 20;                          // patch row pitch in floats (4 tile rows apart = 80 floats = bank 16: two tile rows x four tile columns x four planes hit 32 banks)
constexpr int kPix = kPH * kPW;                  // 324 patch pixels
constexpr int kCC = 8;                           // input channels per step
constexpr int kBN = 64;                          // output channels per item
constexpr int kPlane = 385;                      // patch plane pitch (= 1 mod 32): 18 rows x 20 + 25 floats of sink
constexpr int kSink = kPH * kPR;                 // 360
constexpr int kVB = 72;                          // floats between the V blocks of consecutive slots: [k][tile] with 8 floats of skew behind k = 1
constexpr int kVF = 72 * kVB;                    // 5184
constexpr int kPatchF = kCC * kPlane + 8;        // 3088
constexpr int kStageF = kVF + kPatchF;           // 8272 floats = 33,088 bytes per stage
constexpr unsigned kOOB = 0x80000000u;
constexpr int kEarly = 8;                        // residual-gradient loads issued in front of the output transform (the rest behind it)
}  // namespace

// U4t[ci/8][co/16][g = slot/4 (18)][lane = (ci%4) * 16 + co%16][e = slot%4], slot = ((ci/4)%2) * 36 + pos: the 16-byte load `g` of lane
// `lane` of the wave that owns channel block co/16 in step ci/8.  float64 transform, rounded once.  blockIdx.y = filter of the batch.
__global__ __launch_bounds__(256) void wt_wino4t_kernel(WinoBatch b, int Cin, int Cout) {
    const float* __restrict__ w = b.w[blockIdx.y];
    float* __restrict__ U = b.U[blockIdx.y];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t cc = (size_t)Cin * Cout;
    if (i >= cc) return;
    const int ci = (int)(i / Cout), co = (int)(i - (size_t)ci * Cout);
    double o[36];
    wino4_filter_transform(w, cc, i, o);
    float* dst = U + ((size_t)(ci >> 3) * (Cout >> 4) + (co >> 4)) * (18 * 256) + ((ci & 3) * 16 + (co & 15)) * 4;
    const int sub = (ci >> 2) & 1;
#pragma unroll
    for (int pos = 0; pos < 36; ++pos) {
        const int slot = sub * 36 + pos;
        dst[(slot >> 2) * 256 + (slot & 3)] = (float)o[pos];
    }
}

int wt_wino4t_batch(const WinoBatch& b, int Cin, int Cout, hipStream_t s) {
    if (Cin % kCC || Cout % kBN || b.n < 0 || b.n > 12) return -1;
    if (b.n == 0) return 0;
    const size_t cc = (size_t)Cin * Cout;
    hipLaunchKernelGGL(wt_wino4t_kernel, dim3((unsigned)((cc + 255) / 256), (unsigned)b.n), dim3(256), 0, s, b, Cin, Cout);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int wt_wino4t(const float* w, float* U, int Cin, int Cout, hipStream_t s) {
    WinoBatch b{};
    b.w[0] = w;
    b.U[0] = U;
    b.n = 1;
    return wt_wino4t_batch(b, Cin, Cout, s);
}

#ifdef FS_WINO4T_TRACE
// debug build only (tools/micro_wino4t.py): per-workgroup phase cycle counts of the last launch
__device__ long long g_wino4t_trace[4096 * 8];
extern "C" int fs_debug_wino4t_trace(long long* out, int n_wg) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino4t_trace), sizeof(long long) * 8 * (size_t)n_wg, 0, hipMemcpyDeviceToHost);
}
#define FS_W4T_NOW() ((long long)__builtin_readcyclecounter())
#endif
#ifndef FS_W4T_ABL
#define FS_W4T_ABL 0   /* timing experiments (results wrong): 1 no input transform, 2 no filter loads, 4 no patch loads / commit, 8 no operand reads */
#endif

template <int E>
__device__ __forceinline__ float quad_elem(const float4& v) {
    if constexpr (E == 0) return v.x;
    else if constexpr (E == 1) return v.y;
    else if constexpr (E == 2) return v.z;
    else return v.w;
}

// EPI: 0 raw, 1 raw + instance-norm partials of the item (a.stats), 2 + a.add_src in the interior.  AFF: a.in_a / a.in_b (+ ReLU) on load.
template <int EPI, bool AFF>
__global__ __launch_bounds__(256) void wino4t_conv_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
#ifdef FS_WINO4T_TRACE
    const long long tr_t0 = FS_W4T_NOW();
    long long tr_sweep = 0, tr_bar = 0, tr_epi = 0, tr_pro = 0;
#endif
    const ConvPlan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto fdiv = [](int x, float inv_d) __attribute__((always_inline)) { return (int)(((float)x + 0.5f) * inv_d); };
    auto uniform_ptr = [](const float* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };

    // ---- the item list of this workgroup: item = (n * blocks + block) * ncob + channel block; workgroup b runs on XCD b % 8, the
    // virtual index gives every XCD a contiguous range of items (neighbouring blocks share patch rows in its L2)
    const int blocks = p.tiles_y * p.tiles_x;
    const int ncob = a.Cout / kBN;
    const int nchunks = a.Cin / kCC;
    const int nco16 = a.Cout >> 4;
    const int total_items = a.N * blocks * ncob;
    const int G = (int)gridDim.x;
    const int vb = (G & 7) ? (int)blockIdx.x : (((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3));
    const int my_items = (vb < total_items) ? (total_items - 1 - vb) / G + 1 : 0;
    if (my_items == 0) return;
    const float inv_ncob = 1.0f / (float)ncob, inv_blocks = 1.0f / (float)blocks, inv_tx = 1.0f / (float)p.tiles_x;
    struct Item {
        int n, oy0, ox0, cob, br;
    };
    auto decode = [&](int it) __attribute__((always_inline)) {
        Item r;
        const int lin = vb + it * G;
        const int t2 = fdiv(lin, inv_ncob);
        r.cob = lin - t2 * ncob;
        r.n = fdiv(t2, inv_blocks);
        r.br = t2 - r.n * blocks;
        const int byi = fdiv(r.br, inv_tx);
        r.oy0 = byi * kBH;
        r.ox0 = (r.br - byi * p.tiles_x) * kBW;
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.oy0 = __builtin_amdgcn_readfirstlane(r.oy0);
        r.ox0 = __builtin_amdgcn_readfirstlane(r.ox0);
        r.cob = __builtin_amdgcn_readfirstlane(r.cob);
        r.br = __builtin_amdgcn_readfirstlane(r.br);
        return r;
    };
    struct Cursor {   // over the (item, chunk) steps of this workgroup
        Item I;
        int it, chunk, live;
    };
    auto cursor_begin = [&]() __attribute__((always_inline)) {
        Cursor c;
        c.I = decode(0);
        c.it = 0;
        c.chunk = 0;
        c.live = 1;
        return c;
    };
    auto cursor_next = [&](Cursor& c) __attribute__((always_inline)) {   // returns 1 when the cursor moved to a new item
        if (!c.live) return 0;
        if (++c.chunk < nchunks) return 0;
        if (++c.it >= my_items) {
            c.live = 0;
            return 0;
        }
        c.I = decode(c.it);
        c.chunk = 0;
        return 1;
    };

    // ---- staging state.  Straight-line and identical in every wave; a step that does not exist is loaded through the
    // out-of-range offset (zeros, no traffic) and prepared into a stage nobody reads.
    // filter: uv[g] = slots 4g .. 4g+3 of the step (A operands of the wave's channel block, one per lane)
    float4 uv[18];
    // patch: float4 e = tid + 256 i of the 324 pixels x 2 quads (pixel e >> 1, channels 4 (e & 1) .. of the chunk); pixels >= 324: plane sink
    float4 pv[3];
    float4 fa = make_float4(1.f, 1.f, 1.f, 1.f), fb = make_float4(0.f, 0.f, 0.f, 0.f);   // AFF: scale / shift of the thread's four channels, loaded with the patch
    int pdst[3];
    unsigned gvo[3];
    unsigned avo = kOOB;   // AFF: offset of the thread's scale / shift quad (out of range while the step does not exist)
    const int q_t = tid & 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int pix = (tid + 256 * i) >> 1;
        const int py = pix / kPW;
        pdst[i] = q_t * 4 * kPlane + (pix < kPix ? py * kPR + (pix - py * kPW) : kSink + (pix - kPix) % (kPlane - kSink));
        gvo[i] = kOOB;
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * a.Cin) * 4u);
    const unsigned u_bytes = __builtin_amdgcn_readfirstlane((unsigned)(36 * a.Cin * a.Cout) * 4u);
    const float* ub = uniform_ptr(a.w_wino4t);
    const unsigned uvo = (unsigned)lane * 16u;
    unsigned uvo_eff = uvo;   // kOOB while the step the filter loads are for does not exist
    auto patch_offsets = [&](const Item& I, int live) __attribute__((always_inline)) {   // once per item
        int t_ = tid;
        FS_W4_PIN(t_);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pix = (t_ + 256 * i) >> 1;
            const int py = (int)(((float)pix + 0.5f) * (1.0f / (float)kPW)), px = pix - py * kPW;
            const int sy = I.oy0 - a.pad_t + py, sx = I.ox0 - a.pad_l + px;
            const bool ok = live && pix < kPix && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
            gvo[i] = ok ? (unsigned)((sy * a.W + sx) * a.Cin + 4 * (t_ & 1)) * 4u : kOOB;
        }
        avo = live ? (unsigned)(t_ & 1) * 16u : kOOB;
    };
    auto issue_patch_into = [&](float4& dst, const Item& I, int chunk, int i) __attribute__((always_inline)) {
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * a.Cin);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, x_bytes, 0x00020000);
        dst = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, gvo[i], chunk * kCC * 4, 0));
    };
    auto issue_patch_one = [&](const Item& I, int chunk, int i) __attribute__((always_inline)) { issue_patch_into(pv[i], I, chunk, i); };
    auto issue_affine_into = [&](float4& fa, float4& fb, const Item& I, int chunk) __attribute__((always_inline)) {
        if constexpr (AFF) {
            const unsigned ab_bytes = __builtin_amdgcn_readfirstlane((unsigned)a.Cin * 4u);
            const __amdgpu_buffer_rsrc_t ar =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(a.in_a + (size_t)I.n * a.in_nstride)), 0, ab_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t br =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(a.in_b + (size_t)I.n * a.in_nstride)), 0, ab_bytes, 0x00020000);
            fa = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ar, avo, chunk * kCC * 4, 0));
            fb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(br, avo, chunk * kCC * 4, 0));
        }
    };
    auto issue_affine = [&](const Item& I, int chunk) __attribute__((always_inline)) { issue_affine_into(fa, fb, I, chunk); };
    auto act = [&](float v, float s, float t) __attribute__((always_inline)) {   // AFF: ReLU(v s + t) -- two instructions (wino4t_eligible: in_a comes with in_relu)
        if constexpr (AFF) {
            const float r = fmaf(v, s, t);
#if defined(__HIP_DEVICE_COMPILE__)
            float m;
            asm("v_max_f32 %0, 0, %1" : "=v"(m) : "v"(r));
            return m;
#else
            return r > 0.f ? r : 0.f;
#endif
        } else {
            return v;
        }
    };
    auto commit_quad = [&](int a_pc, const float4& v, const float4& sa, const float4& sb) __attribute__((always_inline)) {   // a_pc: address of the float4's first plane
        FS_W4_LDS(float, a_pc) = act(v.x, sa.x, sb.x);
        FS_W4_LDS(float, a_pc + kPlane * 4) = act(v.y, sa.y, sb.y);
        FS_W4_LDS(float, a_pc + 2 * kPlane * 4) = act(v.z, sa.z, sb.z);
        FS_W4_LDS(float, a_pc + 3 * kPlane * 4) = act(v.w, sa.w, sb.w);
    };
    auto commit_patch_one = [&](int a_pc, int i) __attribute__((always_inline)) { commit_quad(a_pc, pv[i], fa, fb); };
    auto issue_filter_one = [&](const Item& I, int chunk, int i) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, u_bytes, 0x00020000);
        const unsigned so = (unsigned)(((chunk * nco16 + I.cob * 4 + wave) * 18 + i) * 1024);
        uv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ur, uvo_eff, so, 0));
    };
    // input transform V = B^T d B of the 16 tiles x 8 channels of a step on PAIRS of lanes: wave w owns tile rows 2 (w & 1), +1 and
    // channels 4 (w >> 1) .. +3; lane = (half h, channel c, tile row tyl, tile column tx).  Half h does B^T d for columns 3h .. 3h+2,
    // the halves trade nine registers (v_permlane32_swap), half h does (.) B for rows 3h .. 3h+2.
    const int h_t = lane >> 5, c_t = (lane >> 3) & 3, tyl_t = (lane >> 2) & 1, tx_t = lane & 3;
    const int tsrc = (4 * (wave >> 1) + c_t) * kPlane + (4 * (2 * (wave & 1) + tyl_t)) * kPR + 4 * tx_t + 3 * h_t;
    const int tdst = ((wave >> 1) * 36 + 18 * h_t) * kVB + c_t * 16 + 8 * (c_t >> 1) + (2 * (wave & 1) + tyl_t) * 4 + tx_t;
    float td[18], tt[18];
    auto transform_read = [&](int a_pn, int k0, int k1) __attribute__((always_inline)) {   // k = i * 3 + jj: d[i][3h + jj]
#pragma unroll
        for (int k = k0; k < k1; ++k) td[k] = FS_W4_LDS(float, a_pn + ((k / 3) * kPR + (k % 3)) * 4);
    };
    auto transform_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
            FS_W4_BT(td[jj], td[3 + jj], td[6 + jj], td[9 + jj], td[12 + jj], td[15 + jj], tt[jj], tt[3 + jj], tt[6 + jj], tt[9 + jj], tt[12 + jj], tt[15 + jj]);
    };
    auto transform_swap = [&]() __attribute__((always_inline)) {   // -> td[ii * 6 + j] = t[3h + ii][j]
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                float y = tt[i * 3 + jj], x = tt[(i + 3) * 3 + jj];
                FS_W4_SWAP(y, x);
                td[6 * i + jj] = y;
                td[6 * i + 3 + jj] = x;
            }
    };
    auto transform_cols = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ii = 0; ii < 3; ++ii)
            FS_W4_BT(td[6 * ii], td[6 * ii + 1], td[6 * ii + 2], td[6 * ii + 3], td[6 * ii + 4], td[6 * ii + 5], tt[6 * ii], tt[6 * ii + 1], tt[6 * ii + 2],
                     tt[6 * ii + 3], tt[6 * ii + 4], tt[6 * ii + 5]);
    };
    auto transform_write = [&](int a_vn, int k0, int k1) __attribute__((always_inline)) {   // position (3h + ii) * 6 + j of the lane's sub-chunk
#pragma unroll
        for (int k = k0; k < k1; ++k) FS_W4_LDS(float, a_vn + k * (kVB * 4)) = tt[k];
    };

    f32x4 acc[36];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int pos = 0; pos < 36; ++pos) acc[pos] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    Cursor CU = cursor_begin();   // filter cursor: step q+1 during sweep q
    Cursor CP = cursor_begin();   // patch cursor: the step whose patch loads are issued next / were issued last

    // One slice of the next steps' preparation per matrix-instruction slot (72 per sweep):
    //   every 4th  reload of the filter quad whose last slot has just issued (step q+1)
    //   10-15      LDS reads of the patch of step q+1 (the lane's 6 x 3 inputs), three per slot
    //   18         B^T d (36 vector instructions in ONE gap)
    //   22         the halves' exchange (9 swaps) + (.) B
    //   24-32      LDS writes of V, two per slot
    //   34-36      LDS writes of the patch of step q+2 (its loads went out during the previous sweep), affine + ReLU applied
    //   38-40      global loads of the patch of step q+3;  41: its scale / shift quads
    struct Addr {
        int pb;             // B operand of the current stage (+ lane part)
        int vn, pn;         // next stage: the thread's V position 18 h, its patch block d[0][3h]
        int pc[3];          // this stage's patch area: the thread's three float4s (first plane)
    };
    auto stage_addrs = [&](int o0, int o1) __attribute__((always_inline)) {   // o0 / o1: float offsets of the current / the other stage
        Addr A;
        A.pb = FS_W4_ADDR(smem + o0 + (lane >> 4) * 16 + 8 * (lane >> 5) + (lane & 15));
        A.vn = FS_W4_ADDR(smem + o1 + tdst);
        A.pn = FS_W4_ADDR(smem + o1 + kVF + tsrc);
#pragma unroll
        for (int i = 0; i < 3; ++i) A.pc[i] = FS_W4_ADDR(smem + o0 + kVF + pdst[i]);
        FS_W4_PIN(A.pb);
        FS_W4_PIN(A.vn);
        FS_W4_PIN(A.pn);
#pragma unroll
        for (int i = 0; i < 3; ++i) FS_W4_PIN(A.pc[i]);
        return A;
    };
    auto slice = [&](int sl, const Addr& AD) __attribute__((always_inline)) {
        if (sl >= 10 && sl < 16) {
            if (!(FS_W4T_ABL & 1)) transform_read(AD.pn, 3 * (sl - 10), 3 * (sl - 10) + 3);
        } else if (sl == 18) {
            if (!(FS_W4T_ABL & 1)) transform_rows();
        } else if (sl == 22) {
            if (!(FS_W4T_ABL & 1)) {
                transform_swap();
                transform_cols();
            }
        } else if (sl >= 24 && sl < 33) {
            if (!(FS_W4T_ABL & 1)) transform_write(AD.vn, 2 * (sl - 24), 2 * (sl - 24) + 2);
        } else if (sl >= 34 && sl < 37) {
            if (!(FS_W4T_ABL & 4)) commit_patch_one(AD.pc[sl - 34], sl - 34);
        } else if (sl >= 38 && sl < 41) {
            if (!(FS_W4T_ABL & 4)) issue_patch_one(CP.I, CP.chunk, sl - 38);
        } else if (sl == 41) {
            if (!(FS_W4T_ABL & 4)) issue_affine(CP.I, CP.chunk);
        }
    };
    auto sweep = [&](const Addr& AD) __attribute__((always_inline)) {
        float B[3];
        B[0] = FS_W4_LDS(float, AD.pb);
        B[1] = FS_W4_LDS(float, AD.pb + kVB * 4);
        fs_static_for<0, 72>([&](auto SLOT) __attribute__((always_inline)) {
            constexpr int s = decltype(SLOT)::value;
            constexpr int pos = s % 36, c = s % 3, n2 = (s + 2) % 3;
            FS_W4_MFMA_A(acc[pos], quad_elem<(s & 3)>(uv[s >> 2]), B[c]);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < 72 && !(FS_W4T_ABL & 8)) B[n2] = FS_W4_LDS(float, AD.pb + (s + 2) * (kVB * 4));   // operand two slots ahead
            slice(s, AD);
            if ((s & 3) == 3 && !(FS_W4T_ABL & 2)) issue_filter_one(CU.I, CU.chunk, s >> 2);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- epilogue of one item.  Lane (j = lane & 15, g = lane >> 4) of wave w holds tile j = (ty, tx) and the four channels
    // co0 + 16 w + 4 g .. + 3 of all 36 positions.  Pixels outside the image carry the out-of-range offset (loads 0, stores dropped).
    auto epilogue_body = [&](auto FULLT, const Item& I) __attribute__((always_inline)) {
        constexpr bool full = decltype(FULLT)::value;
        int ln = lane;
        FS_W4_PIN(ln);
        const int j = ln & 15;
        const int oy = I.oy0 + 4 * (j >> 2), ox = I.ox0 + 4 * (j & 3);
        const int co = I.cob * kBN + wave * 16 + 4 * (ln >> 4);
        const float* yb = a.y + (size_t)I.n * a.Ho * a.Wo * a.Cout;
        const unsigned img_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * a.Cout) * 4u);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yb)), 0, img_bytes, 0x00020000);
        const unsigned rowp4 = __builtin_amdgcn_readfirstlane((unsigned)(a.Wo * a.Cout) * 4u), col4 = __builtin_amdgcn_readfirstlane((unsigned)a.Cout * 4u);
        const unsigned obase = (unsigned)((oy * a.Wo + ox) * a.Cout + co) * 4u;
        const int ry = a.Ho - oy, cx = a.Wo - ox;   // valid rows / columns of the lane's tile (edge blocks)
        auto inside = [&](int px) __attribute__((always_inline)) { return full || ((px >> 2) < ry && (px & 3) < cx); };
        auto voff = [&](int px) __attribute__((always_inline)) { return inside(px) ? obase : kOOB; };
        auto soff = [&](int px) __attribute__((always_inline)) { return (unsigned)(px >> 2) * rowp4 + (unsigned)(px & 3) * col4; };
        // residual gradient (EPI 2): [N][Ho - 2 add_pad][Wo - 2 add_pad][Cout], added where it exists
        float4 ad[16];
        const int Ha = a.Ho - 2 * a.add_pad, Wa = a.Wo - 2 * a.add_pad;
        const float* adn = EPI == 2 ? a.add_src + (size_t)I.n * Ha * Wa * a.Cout : yb;
        const unsigned add_bytes = __builtin_amdgcn_readfirstlane((unsigned)(Ha * Wa * a.Cout) * 4u);
        const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(adn)), 0, add_bytes, 0x00020000);
        auto add_load = [&](int px) __attribute__((always_inline)) {
            const int ay = oy + (px >> 2) - a.add_pad, ax = ox + (px & 3) - a.add_pad;
            const bool ok = ay >= 0 && ay < Ha && ax >= 0 && ax < Wa;
            ad[px] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ar, ok ? (unsigned)((ay * Wa + ax) * a.Cout + co) * 4u : kOOB, 0, 0));
        };
        if (EPI == 2) {
#pragma unroll
            for (int px = 0; px < kEarly; ++px) add_load(px);
        }
        __builtin_amdgcn_sched_barrier(0);
        float o[16][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s[4][6];   // A^T M: rows 0..3, columns 0..5
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const float m0 = FS_ACC_READ(acc[q][r]), m1 = FS_ACC_READ(acc[6 + q][r]), m2 = FS_ACC_READ(acc[12 + q][r]), m3 = FS_ACC_READ(acc[18 + q][r]),
                            m4 = FS_ACC_READ(acc[24 + q][r]), m5 = FS_ACC_READ(acc[30 + q][r]);   // (each element read ONCE: the reads are volatile)
                FS_W4_AT(m0, m1, m2, m3, m4, m5, s[0][q], s[1][q], s[2][q], s[3][q]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                FS_W4_AT(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], o[4 * i][r], o[4 * i + 1][r], o[4 * i + 2][r], o[4 * i + 3][r]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (EPI == 2) {
#pragma unroll
            for (int px = kEarly; px < 16; ++px) add_load(px);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int px = 0; px < 16; ++px) {
            float4 v = make_float4(o[px][0], o[px][1], o[px][2], o[px][3]);
            if (EPI == 2) {
                v.x += ad[px].x;
                v.y += ad[px].y;
                v.z += ad[px].z;
                v.w += ad[px].w;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fs_u32x4, v), yr, voff(px), soff(px), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the statistics come BEHIND the stores: their ~300 instructions cover part of the stores' way to memory, which the
        // drain in front of the next item's sweeps would otherwise wait out idle)
        if (EPI == 1) {
            // per-item instance-norm partials of the RAW output {mean, M2, count} around a shift (the block's first pixel): sum over
            // the lane's 16 pixels, then over the 16 tiles = the 16 lanes of a row
            float cs[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[r] = __shfl(o[0][r], ln & 48);
#pragma unroll
            for (int px = 0; px < 16; ++px)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dv = inside(px) ? o[px][r] - cs[r] : 0.f;
                    s1[r] += dv;
                    s2[r] = fmaf(dv, dv, s2[r]);
                }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s1[r] += __shfl_xor(s1[r], m);
                    s2[r] += __shfl_xor(s2[r], m);
                }
            if (j == 0) {
                const int th_valid = min(kBH, a.Ho - I.oy0), tw_valid = min(kBW, a.Wo - I.ox0);
                const float cnt = (float)(th_valid * tw_valid);
                float* st = a.stats + ((size_t)(I.n * blocks + I.br) * a.Cout + co) * 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[3 * r] = cs[r] + s1[r] / cnt;
                    st[3 * r + 1] = fmaxf(s2[r] - s1[r] * s1[r] / cnt, 0.f);
                    st[3 * r + 2] = cnt;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto epilogue = [&](const Item& I) __attribute__((always_inline)) {
        if (I.oy0 + kBH <= a.Ho && I.ox0 + kBW <= a.Wo)
            epilogue_body(std::true_type{}, I);
        else
            epilogue_body(std::false_type{}, I);
    };
    // (No vector-memory drain between items or behind the prologue, unlike fs_wino4.hip: with the filter quads consumed one sweep
    // after their loads the compiler keeps exact vmcnt counts in the sweep either way (checked in the ISA: vmcnt(20) / vmcnt(22)), and
    // an exact count is safe with the epilogue's stores still in flight -- loads return in order among themselves, so "at most N
    // operations outstanding" implies that a load with N younger LOADS behind it has arrived, whatever the stores do.)
    auto next_item = [&]() __attribute__((always_inline)) { zero_acc(); };

    // ---- prologue: step 0 complete in stage 0 (patch, V) and in registers (filter), the patch of step 1 in stage 1, the patch of step 2
    // in registers.  Every load of steps 0 and 1 goes out before the first wait (patches first: they are needed first, and
    // the counter retires in order), the accumulators are zeroed while they fly.
    const Addr AP0 = stage_addrs(kStageF, 0), AP1 = stage_addrs(0, kStageF);   // "next stage" = stage 0 / stage 1
    patch_offsets(CP.I, 1);
    float4 pv0[3], fa0 = fa, fb0 = fb;   // step 0's patch and scale / shift: registers of their own, so that step 1's loads need not wait for them
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_patch_into(pv0[i], CP.I, CP.chunk, i);
    issue_affine_into(fa0, fb0, CP.I, CP.chunk);
    if (cursor_next(CP) || !CP.live) patch_offsets(CP.I, CP.live);
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_patch_one(CP.I, CP.chunk, i);
    issue_affine(CP.I, CP.chunk);
#pragma unroll
    for (int i = 0; i < 18; ++i) issue_filter_one(CU.I, CU.chunk, i);
    zero_acc();
#pragma unroll
    for (int i = 0; i < 3; ++i) commit_quad(AP1.pc[i], pv0[i], fa0, fb0);   // (AP1's current stage is stage 0)
#pragma unroll
    for (int i = 0; i < 3; ++i) commit_patch_one(AP0.pc[i], i);   // stage 1's patch area
    if (cursor_next(CP) || !CP.live) patch_offsets(CP.I, CP.live);
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_patch_one(CP.I, CP.chunk, i);
    issue_affine(CP.I, CP.chunk);
    cursor_next(CU);   // the filter cursor now points at step 1
    __syncthreads();
    transform_read(AP0.pn, 0, 18);
    transform_rows();
    transform_swap();
    transform_cols();
    transform_write(AP0.vn, 0, 18);
    __syncthreads();
#ifdef FS_WINO4T_TRACE
    tr_pro = FS_W4T_NOW() - tr_t0;
#endif

    // ---- the flat pipeline over (item, chunk) steps: step q multiplies V of stage q & 1 with the filter registers while they are
    // reloaded for step q+1, V of step q+1 is prepared into the other stage, the patch of step q+2 lands in this stage's patch
    // area and the patch loads of step q+3 go out
    int q = 0;
    for (int it = 0; it < my_items; ++it) {
        const Item cur_it = decode(it);
        for (int chunk = 0; chunk < nchunks; ++chunk, ++q) {
            uvo_eff = CU.live ? uvo : kOOB;
            if (cursor_next(CP) || !CP.live) patch_offsets(CP.I, CP.live);   // (the offsets change once per item)
#ifdef FS_WINO4T_TRACE
            const long long q0 = FS_W4T_NOW();
#endif
            const int o0 = (q & 1) ? kStageF : 0, o1 = kStageF - o0;
            const Addr AD = stage_addrs(o0, o1);
            __builtin_amdgcn_sched_barrier(0);
            sweep(AD);
            cursor_next(CU);
#ifdef FS_WINO4T_TRACE
            const long long q1 = FS_W4T_NOW();
#endif
            FS_LDS_BARRIER();
#ifdef FS_WINO4T_TRACE
            const long long q2 = FS_W4T_NOW();
            tr_sweep += q1 - q0;
            tr_bar += q2 - q1;
#endif
        }
#ifdef FS_WINO4T_TRACE
        const long long e0 = FS_W4T_NOW();
#endif
        epilogue(cur_it);
        next_item();   // (also behind the last item: an `if (it + 1 < my_items)` makes the compiler restructure the item loop -- 512 registers + scratch)
#ifdef FS_WINO4T_TRACE
        tr_epi += FS_W4T_NOW() - e0;
#endif
    }
#ifdef FS_WINO4T_TRACE
    if (tid == 0 && blockIdx.x < 4096) {
        long long* t = g_wino4t_trace + (size_t)blockIdx.x * 8;
        t[0] = tr_t0;
        t[1] = tr_pro;
        t[2] = tr_sweep;
        t[3] = tr_bar;
        t[4] = tr_epi;
        t[5] = q;
        t[6] = FS_W4T_NOW();
        t[7] = my_items;
    }
#endif
}

bool wino4t_eligible(const ConvArgs& a) {
    // 3x3 stride 1 with padding 0 (the residual convs), 2 (their input gradients) or 1; plain source; no bias / activation / mask
    const bool pad_ok = a.pad_t == a.pad_l && a.pad_t >= 0 && a.pad_t <= 2 && a.Ho == a.H + 2 * a.pad_t - 2 && a.Wo == a.W + 2 * a.pad_l - 2;
    // byte offsets inside one sample are 32-bit with the top bit reserved for "out of range"; the filter likewise
    const bool fits = (double)a.H * a.W * a.Cin * 4.0 < 2147483648.0 && (double)a.Ho * a.Wo * a.Cout * 4.0 < 2147483648.0 && 36.0 * a.Cin * a.Cout * 4.0 < 2147483648.0;
    const bool aff_ok = !a.in_a || (a.pad_t == 0 && a.in_b && a.in_relu && (a.in_nstride == 0 || a.in_nstride == a.Cin));
    return fits && a.w_wino4t && a.KH == 3 && a.KW == 3 && a.stride == 1 && pad_ok && a.src_mode == SRC_PLAIN && a.Cin % kCC == 0 && a.Cout % kBN == 0 &&
           !a.shuffle && aff_ok && !a.bias && !a.out_relu && !a.mask_src && !a.route_src && !a.pool_out && a.w_nstride == 0 && a.dil_x <= 1 &&
           !(a.add_src && (a.stats || a.in_a)) && !a.fin.counter && a.Ho > 0 && a.Wo > 0;
}

long wino4t_items(const ConvArgs& a) { return (long)a.N * cdiv(a.Ho, kBH) * cdiv(a.Wo, kBW) * (a.Cout / kBN); }

void wino4t_plan(const ConvArgs& a, ConvPlan* out) {
    ConvPlan p{};
    p.variant = 11;
    p.BN = kBN;
    p.CC = kCC;
    p.TH = kBH;
    p.TW = kBW;
    p.tiles_y = cdiv(a.Ho, kBH);
    p.tiles_x = cdiv(a.Wo, kBW);
    p.lds_bytes = 4 * 2 * kStageF;
    p.ksplit = 1;
    *out = p;
}

template <int EPI, bool AFF>
static int wino4t_launch_as(const ConvArgs& a, long grid, hipStream_t s) {
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(wino4t_conv_kernel<EPI, AFF>));
    hipLaunchKernelGGL((wino4t_conv_kernel<EPI, AFF>), dim3((unsigned)grid), dim3(256), (size_t)a.p.lds_bytes, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int wino4t_launch(const ConvArgs& a, hipStream_t s) {
    const long items = wino4t_items(a);
    const int wgs = tune_int("FS_WINO4T_WGS", 256);
    const long grid = items < wgs ? items : wgs;
    if (a.in_a) return a.stats ? wino4t_launch_as<1, true>(a, grid, s) : wino4t_launch_as<0, true>(a, grid, s);
    if (a.stats) return wino4t_launch_as<1, false>(a, grid, s);
    if (a.add_src) return wino4t_launch_as<2, false>(a, grid, s);
    return wino4t_launch_as<0, false>(a, grid, s);
}

}  // namespace fs
