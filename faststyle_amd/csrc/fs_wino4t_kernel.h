// The kernel template of fs_wino4t.hip / fs_wino4t2.hip (Winograd F(4x4,3x3), filter operand global -> registers) -- a header so that the
// fourteen instantiations compile in four translation units (fs_wino4t.hip, fs_wino4t1b.hip, fs_wino4t2.hip, fs_wino4t2b.hip: the
// sanitizer build of one file with all of them took five minutes).  See fs_wino4t.hip for the description.
#pragma once
#include "fs_wino4.h"

#include <cstdlib>
#include <type_traits>

namespace fs {

namespace {   // (internal linkage: both translation units include this)
constexpr int kBH = 16;                          // output rows per item
constexpr int kPH = 18;                          // input patch rows
constexpr int kCC = 8;                           // input channels per step
constexpr int kBN = 64;                          // output channels per item
constexpr unsigned kOOB = 0x80000000u;
constexpr int kNA = 32;                          // TB = 2: positions whose accumulators live in the accumulator file (the other 4: vector registers, see fs_wino4.h)
// geometry of an item with TB tile blocks of 16 (4 tile rows x 4 TB tile columns)
template <int TB, bool FL = false>
struct Geo {
    static constexpr int kBW = 16 * TB;                      // output columns per item
    static constexpr int kPW = kBW + 2;                      // patch columns
    static constexpr int kPR = TB == 1 ? 20 : 34;            // patch row pitch in floats: the lanes of a transform half-wave (tile columns x tile rows x 4 planes) hit 32 banks
    static constexpr int kPix = kPH * kPW;                   // 324 / 612 patch pixels
    static constexpr int kSink = kPH * kPR;                  // start of the plane's sink (slots past the patch write here)
    static constexpr int kPlane = TB == 1 ? 385 : 641;       // plane pitch (= 1 mod 32)
    static constexpr int kNPV = TB == 1 ? 3 : 5;             // 16-byte patch loads per thread and step (648 / 1224 of them)
    static constexpr int kVB = TB == 1 ? 72 : 160;           // floats between the V blocks of consecutive slots ([k][tile], skewed per k)
    static constexpr int kVF = 72 * kVB;
    static constexpr int kPatchF = kCC * kPlane + 8;
    static constexpr int kStageF = kVF + kPatchF;            // 8272 / 15792 floats per stage
    static constexpr int kSlots = 72 * TB;                   // matrix instructions per sweep and wave
    static constexpr int kEarly = 8;                         // residual-gradient / mask loads issued in front of the output transform (the rest behind it)
    // offset of row k of a V block: TB = 1 [k][16 tiles] with 8 floats of skew behind k = 1; TB = 2 [k][16][2 tile blocks] at 0, 48, 80, 128
    // (the operand read is ONE 8-byte read per lane -- both tile blocks --, conflict-free per 16 lanes; the transform's writes
    // meet two-way: 18 per pass)
    static constexpr int koff(int k) { return TB == 1 ? k * 16 + 8 * (k >> 1) : k * 32 + 16 * ((k + 1) >> 1); }
    static constexpr bool kDefer = TB == 2;                  // an item's last sweep loads nothing; next_item does (see there)
};
// FL (round 5): 16 tiles of a FLATTENED per-sample tile list -- any 16 consecutive tiles of the row-major tile grid, not a 4 x 4 block -- so every
// tile stages its OWN 6 x 6 input patch (no shared halo): plane = [tile 16][36 pixels], tile pitch 37 and plane pitch = 8 mod 32 put the 8 tiles x
// 4 channels of a transform half-wave on 32 distinct banks (5 t + 8 c mod 32).  Items per sample = ceil(tiles / 16) instead of
// ceil(Ho / 16) ceil(Wo / 16): 23 instead of 25 at 74 x 74 -- at batch 32 the difference between three and four rounds of a 256-workgroup grid.
template <>
struct Geo<1, true> {
    static constexpr int kBW = 16;
    static constexpr int kPW = 6;                            // (a tile's patch is 6 x 6)
    static constexpr int kPR = 6;                            // patch row pitch inside a tile's patch
    static constexpr int kTP = 37;                           // tile pitch
    static constexpr int kPix = 16 * 36;                     // 576 pixel slots
    static constexpr int kSink = 16 * kTP;                   // 592
    static constexpr int kPlane = 616;                       // (= 8 mod 32)
    static constexpr int kNPV = 5;                           // 16-byte patch loads per thread and step (1152 of 1280 slots used)
    static constexpr int kVB = 72;
    static constexpr int kVF = 72 * kVB;
    static constexpr int kPatchF = kCC * kPlane + 8;
    static constexpr int kStageF = kVF + kPatchF;            // 10120 floats per stage
    static constexpr int kSlots = 72;
    static constexpr int kEarly = 8;
    static constexpr int koff(int k) { return k * 16 + 8 * (k >> 1); }
    static constexpr bool kDefer = false;
};
}  // namespace

#ifdef FS_WINO4T_TRACE
// debug build only (tools/micro_wino4t.py, tools/micro_wino4.py): per-workgroup phase cycle counts of the last launch -- one buffer per
// translation unit (fs_debug_wino4t_trace_<unit>, unit = 1a, 1b, 2a, 2b; the micro-benchmarks read all four and keep the latest)
static __device__ long long g_wino4t_trace[4096 * 8];
static inline int wino4t_trace_read(long long* out, int n_wg) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino4t_trace), sizeof(long long) * 8 * (size_t)n_wg, 0, hipMemcpyDeviceToHost);
}
#define FS_W4T_NOW() ((long long)__builtin_readcyclecounter())
#endif
#ifndef FS_W4T_EARLY_TB2_MASK
#define FS_W4T_EARLY_TB2_MASK 8   /* 32-tile items, consumer-mask epilogue: mask loads issued in front of the output transform (of 16 per tile block) */
#endif
#ifndef FS_W4T_INB_DEFER
#define FS_W4T_INB_DEFER 1   /* EPI 5 / 6: an item's last sweep loads nothing (as the 32-tile forms): the filter registers are free for the epilogue's 32 quads of loads */
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

// M: item form -- 1: 16 tiles x 64 channels; 2: 32 tiles (two tile blocks) x 64 channels; 3: 16 tiles x 128 channels (two CHANNEL blocks per
// wave: the input transform of a step serves twice the products; 36 filter quads per step through an 18-register window); 4: form 1 on a
// flattened per-sample tile list (Geo<1, true>: a.p.flat_tiles, a.p.tiles_x = items per sample).
// EPI: 0 raw (also split-K partials), 1 raw + instance-norm partials of the item (a.stats), 2 + a.add_src in
// the interior, 3 bias + ReLU (+ a.pool_out), 4 a.mask_src, 5 raw + the instance-norm-BACKWARD partial sums of the unit whose output
// gradient the launch writes (a.inb_*, round 5), 6 = 2 + the same.  AFF: a.in_a / a.in_b + ReLU on load.
template <int M, int EPI, bool AFF>
__global__ __launch_bounds__(256) void wino4t_conv_kernel(ConvArgs a) {
    constexpr int TB = M == 2 ? 2 : 1;      // tile blocks of 16 per item (geometry)
    constexpr bool CB2 = M == 3;            // two channel blocks of 16 per wave
    constexpr bool FL = M == 4;             // flattened tile list, per-tile patches
    constexpr int NB = (M == 1 || M == 4) ? 1 : 2;      // accumulator blocks per position
    constexpr int kBNi = CB2 ? 2 * kBN : kBN;   // output channels per item
    constexpr int QPS = CB2 ? 36 : 18;      // filter quads per step and wave
    static_assert(!(CB2 && (EPI == 1 || EPI == 2 || AFF)), "the 128-channel form carries the VGG16 epilogues only");
    static_assert(!((EPI == 5 || EPI == 6) && ((M != 1 && M != 4) || AFF)), "the instance-norm-backward epilogues exist for 16-tile items only");
    static_assert(!(FL && (EPI == 3 || EPI == 4)), "the flattened form carries the transform net's epilogues");
    constexpr bool kAdd = EPI == 2 || EPI == 6, kInb = EPI == 5 || EPI == 6;
    using GEO = Geo<TB, FL>;
    constexpr int kBW = GEO::kBW, kPW = GEO::kPW, kPR = GEO::kPR, kPix = GEO::kPix, kSink = GEO::kSink, kPlane = GEO::kPlane, kNPV = GEO::kNPV, kVB = GEO::kVB,
                  kVF = GEO::kVF, kStageF = GEO::kStageF;
    constexpr int kEarly = (TB == 2 && EPI == 4) ? FS_W4T_EARLY_TB2_MASK : (((EPI == 5 || EPI == 6) && FS_W4T_INB_DEFER) ? 16 : GEO::kEarly);
    // (288 accumulator registers: the epilogue needs the staging registers; the instance-norm-backward forms keep 32 quads of loads beside the 64 outputs)
    constexpr bool kDefer = NB == 2 || ((EPI == 5 || EPI == 6) && FS_W4T_INB_DEFER);
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

    // ---- the item list of this workgroup: item = ((n * blocks + block) * ncob + channel block) * ksplit + z; workgroup b runs on XCD b % 8,
    // the virtual index gives every XCD a contiguous range of items (the channel blocks of a pixel block and neighbouring blocks
    // share patch rows in its L2)
    const int blocks = p.tiles_y * p.tiles_x;
    const int ncob = a.Cout / kBNi;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int nchunks_all = a.Cin / kCC;
    const int nblk = CB2 ? a.Cout >> 5 : a.Cout >> 4;   // filter blocks of the layer (one per wave)
    const int total_items = a.N * blocks * ncob * ks;
    const int G = (int)gridDim.x;
    const int vb = (G & 7) ? (int)blockIdx.x : (((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3));
    const int my_items = (vb < total_items) ? (total_items - 1 - vb) / G + 1 : 0;
    if (my_items == 0) return;
    const float inv_ks = 1.0f / (float)ks, inv_ncob = 1.0f / (float)ncob, inv_blocks = 1.0f / (float)blocks, inv_tx = 1.0f / (float)p.tiles_x;
    const int gTx = (a.Wo + 3) >> 2, gT = ((a.Ho + 3) >> 2) * gTx;   // FL: the sample's grid of 4 x 4-pixel tiles, row-major
    const float inv_gtx = 1.0f / (float)gTx;
    struct Item {
        int n, oy0, ox0, cob, br, cbeg, cend, z;
    };
    auto decode = [&](int it) __attribute__((always_inline)) {
        Item r;
        const int lin = vb + it * G;
        const int t1 = fdiv(lin, inv_ks);
        r.z = lin - t1 * ks;
        const int t2 = fdiv(t1, inv_ncob);
        r.cob = t1 - t2 * ncob;
        r.n = fdiv(t2, inv_blocks);
        r.br = t2 - r.n * blocks;
        const int byi = fdiv(r.br, inv_tx);
        r.oy0 = byi * kBH;
        r.ox0 = (r.br - byi * p.tiles_x) * kBW;
        r.cbeg = ks > 1 ? r.z * nchunks_all / ks : 0;
        r.cend = ks > 1 ? (r.z + 1) * nchunks_all / ks : nchunks_all;
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.oy0 = __builtin_amdgcn_readfirstlane(r.oy0);
        r.ox0 = __builtin_amdgcn_readfirstlane(r.ox0);
        r.cob = __builtin_amdgcn_readfirstlane(r.cob);
        r.br = __builtin_amdgcn_readfirstlane(r.br);
        r.cbeg = __builtin_amdgcn_readfirstlane(r.cbeg);
        r.cend = __builtin_amdgcn_readfirstlane(r.cend);
        r.z = __builtin_amdgcn_readfirstlane(r.z);
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

    // ---- staging state.  Straight-line and identical in every wave; a step that does not exist is loaded through the
    // out-of-range offset (zeros, no traffic) and prepared into a stage nobody reads.
    // filter: uv[g] = slots 4g .. 4g+3 of the step (A operands of the wave's channel block, one per lane)
    float4 uv[18];
    // patch: float4 e = tid + 256 i of the patch pixels x 2 quads (pixel e >> 1, channels 4 (e & 1) .. of the chunk); pixels past the patch: plane sink
    float4 pv[kNPV];
    float4 fa = make_float4(1.f, 1.f, 1.f, 1.f), fb = make_float4(0.f, 0.f, 0.f, 0.f);   // AFF: scale / shift of the thread's four channels, loaded with the patch
    int pdst[kNPV];
    unsigned gvo[kNPV];
    unsigned avo = kOOB;   // AFF: offset of the thread's scale / shift quad (out of range while the step does not exist)
    const int q_t = tid & 1;
#pragma unroll
    for (int i = 0; i < kNPV; ++i) {
        const int pix = (tid + 256 * i) >> 1;
        if constexpr (FL) {   // pixel slot = (tile, pixel of its 6 x 6 patch)
            const int tl = pix / 36;
            pdst[i] = q_t * 4 * kPlane + (pix < kPix ? tl * GEO::kTP + (pix - 36 * tl) : kSink + (pix - kPix) % (kPlane - kSink));
        } else {
            const int py = pix / kPW;
            pdst[i] = q_t * 4 * kPlane + (pix < kPix ? py * kPR + (pix - py * kPW) : kSink + (pix - kPix) % (kPlane - kSink));
        }
        gvo[i] = kOOB;
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * a.Cin) * 4u);
    const unsigned u_bytes = __builtin_amdgcn_readfirstlane((unsigned)(36 * a.Cin * a.Cout) * 4u);
    const unsigned ab_bytes = __builtin_amdgcn_readfirstlane((unsigned)a.Cin * 4u);
    unsigned load_on = 1u;   // 0 during the LAST sweep of an item: every staging load of that sweep gets an empty buffer (see next_item)
    const float* ub = uniform_ptr(CB2 ? a.w_wino4u : a.w_wino4t);
    const unsigned uvo = (unsigned)lane * 16u;
    unsigned uvo_eff = uvo;   // kOOB while the step the filter loads are for does not exist
    auto patch_offsets = [&](const Item& I, int live) __attribute__((always_inline)) {   // once per item
        int t_ = tid;
        FS_W4_PIN(t_);
#pragma unroll
        for (int i = 0; i < kNPV; ++i) {
            const int pix = (t_ + 256 * i) >> 1;
            if constexpr (FL) {
                const int tl = (int)(((float)pix + 0.5f) * (1.0f / 36.0f)), pp = pix - 36 * tl;
                const int py = (int)(((float)pp + 0.5f) * (1.0f / 6.0f)), px = pp - 6 * py;
                const int t = 16 * I.br + tl;
                const int ty = fdiv(t, inv_gtx), tx = t - ty * gTx;
                const int sy = 4 * ty - a.pad_t + py, sx = 4 * tx - a.pad_l + px;
                const bool ok = live && pix < kPix && t < gT && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
                gvo[i] = ok ? (unsigned)((sy * a.W + sx) * a.Cin + 4 * (t_ & 1)) * 4u : kOOB;
            } else {
                const int py = (int)(((float)pix + 0.5f) * (1.0f / (float)kPW)), px = pix - py * kPW;
                const int sy = I.oy0 - a.pad_t + py, sx = I.ox0 - a.pad_l + px;
                const bool ok = live && pix < kPix && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
                gvo[i] = ok ? (unsigned)((sy * a.W + sx) * a.Cin + 4 * (t_ & 1)) * 4u : kOOB;
            }
        }
        avo = live ? (unsigned)(t_ & 1) * 16u : kOOB;
    };
    auto issue_patch_into = [&](float4& dst, const Item& I, int chunk, int i) __attribute__((always_inline)) {
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * a.Cin);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, load_on ? x_bytes : 0u, 0x00020000);
        dst = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, gvo[i], chunk * kCC * 4, 0));
    };
    auto issue_patch_one = [&](const Item& I, int chunk, int i) __attribute__((always_inline)) { issue_patch_into(pv[i], I, chunk, i); };
    auto issue_affine_into = [&](float4& da, float4& db, const Item& I, int chunk) __attribute__((always_inline)) {
        if constexpr (AFF) {
            const unsigned ab_eff = load_on ? ab_bytes : 0u;
            const __amdgpu_buffer_rsrc_t ar =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(a.in_a + (size_t)I.n * a.in_nstride)), 0, ab_eff, 0x00020000);
            const __amdgpu_buffer_rsrc_t br =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(a.in_b + (size_t)I.n * a.in_nstride)), 0, ab_eff, 0x00020000);
            da = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ar, avo, chunk * kCC * 4, 0));
            db = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(br, avo, chunk * kCC * 4, 0));
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
    // quad `quad` (of QPS) of step (I, chunk) into register set `dst`; on = 0: against an empty buffer (zeros, no traffic)
    auto issue_filter_q = [&](const Item& I, int chunk, int quad, int dst, unsigned on, unsigned vo) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, on ? u_bytes : 0u, 0x00020000);
        const unsigned so = (unsigned)(((chunk * nblk + I.cob * 4 + wave) * QPS + quad) * 1024);
        uv[dst] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ur, vo, so, 0));
    };
    auto issue_filter_one = [&](const Item& I, int chunk, int i) __attribute__((always_inline)) { issue_filter_q(I, chunk, i, i, load_on, uvo_eff); };
    // input transform V = B^T d B of the tiles x 8 channels of a step on PAIRS of lanes.  Half h does B^T d for columns 3h .. 3h+2, the
    // halves trade nine registers (v_permlane32_swap), half h does (.) B for rows 3h .. 3h+2.
    //   TB = 1: one pass; wave w owns tile rows 2 (w & 1), +1 and channels 4 (w >> 1) .. +3; lane = (half h, channel c, tile row tyl, tile column tx)
    //   TB = 2: two passes (pass = the step's first / second four channels); wave w owns tile row w; lane = (half h, channel c, tile column tx of 8)
    const int h_t = lane >> 5, c_t = (lane >> 3) & 3;
    int tsrc_ = TB == 1 ? (4 * (wave >> 1) + c_t) * kPlane + (4 * (2 * (wave & 1) + ((lane >> 2) & 1))) * kPR + 4 * (lane & 3) + 3 * h_t
                        : c_t * kPlane + (4 * wave) * kPR + 4 * (lane & 7) + 3 * h_t;
    if constexpr (FL)   // the lane's tile (4 (2 (w & 1) + tile row bit) + tile column of the rectangular form = the same tile INDEX) has a patch of its own
        tsrc_ = (4 * (wave >> 1) + c_t) * kPlane + ((2 * (wave & 1) + ((lane >> 2) & 1)) * 4 + (lane & 3)) * Geo<1, true>::kTP + 3 * h_t;
    const int tsrc = tsrc_;
    const int tdst = TB == 1 ? ((wave >> 1) * 36 + 18 * h_t) * kVB + GEO::koff(c_t) + (2 * (wave & 1) + ((lane >> 2) & 1)) * 4 + (lane & 3)
                             : 18 * h_t * kVB + GEO::koff(c_t) + 2 * (8 * (wave & 1) + (lane & 7)) + (wave >> 1);
    constexpr int kPassSrc = 4 * kPlane * 4, kPassDst = 36 * kVB * 4;   // TB = 2: byte offsets of the second pass
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
        for (int k = k0; k < k1; ++k) FS_W4_LDS_STORE1(a_vn + k * (kVB * 4), tt[k]);
    };

    // accumulators: [position][tile block].  TB = 2 needs 288 registers: positions >= kNA live in ordinary vector registers (fs_wino4.h)
    constexpr int kNAcc = NB == 1 ? 36 : kNA;
    f32x4 acc[kNAcc][NB];
    f32x4 accv[NB == 1 ? 1 : 36 - kNA][NB];
#define FS_W4T_ACC(pos, tb, r) ((pos) < kNAcc ? FS_ACC_READ(acc[(pos) < kNAcc ? (pos) : 0][tb][r]) : accv[(pos) >= kNAcc ? (pos) - kNAcc : 0][tb][r])
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int pos = 0; pos < kNAcc; ++pos)
#pragma unroll
            for (int tb = 0; tb < NB; ++tb) acc[pos][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NB == 2) {
#pragma unroll
            for (int pos = 0; pos < 36 - kNA; ++pos)
#pragma unroll
                for (int tb = 0; tb < NB; ++tb) accv[pos][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    Cursor CU = cursor_begin();   // filter cursor: step q+1 during sweep q
    Cursor CP = cursor_begin();   // patch cursor: the step whose patch loads are issued next / were issued last

    // One slice of the next steps' preparation per matrix-instruction slot (72 TB per sweep).  TB = 1 | TB = 2:
    //   behind every 4th operand  reload of the filter quad whose last slot has just issued (step q+1)
    //   10-15 | 10-15, 46-51      LDS reads of the patch of step q+1 (the lane's 6 x 3 inputs), three per slot
    //   18    | 18, 54            B^T d (36 vector instructions in ONE gap)
    //   22    | 22, 58            the halves' exchange (9 swaps) + (.) B
    //   24-32 | 24-32, 60-68      LDS writes of V, two per slot
    //   34-36 | 76-80             LDS writes of the patch of step q+2 (its loads went out during the previous sweep), affine + ReLU applied
    //   38-40 | 86-90             global loads of the patch of step q+3;  41 | 91: its scale / shift quads
    struct Addr {
        int pb;             // B operand of the current stage (+ lane part)
        int vn, pn;         // next stage: the thread's V position 18 h, its patch block d[0][3h]
        int pc[kNPV];       // this stage's patch area: the thread's float4s (first plane)
    };
    auto stage_addrs = [&](int o0, int o1) __attribute__((always_inline)) {   // o0 / o1: float offsets of the current / the other stage
        Addr A;
        A.pb = FS_W4_ADDR(smem + o0 + GEO::koff(lane >> 4) + TB * (lane & 15));
        A.vn = FS_W4_ADDR(smem + o1 + tdst);
        A.pn = FS_W4_ADDR(smem + o1 + kVF + tsrc);
#pragma unroll
        for (int i = 0; i < kNPV; ++i) A.pc[i] = FS_W4_ADDR(smem + o0 + kVF + pdst[i]);
        FS_W4_PIN(A.pb);
        FS_W4_PIN(A.vn);
        FS_W4_PIN(A.pn);
#pragma unroll
        for (int i = 0; i < kNPV; ++i) FS_W4_PIN(A.pc[i]);
        return A;
    };
    auto transform_slice = [&](int sl, int a_pn, int a_vn) __attribute__((always_inline)) {   // sl relative to the pass: reads 0-5, rows 8, swap + cols 12, writes 14-22
        if (FS_W4T_ABL & 1) return;
        if (sl >= 0 && sl < 6) transform_read(a_pn, 3 * sl, 3 * sl + 3);
        else if (sl == 8) transform_rows();
        else if (sl == 12) {
            transform_swap();
            transform_cols();
        } else if (sl >= 14 && sl < 23) transform_write(a_vn, 2 * (sl - 14), 2 * (sl - 14) + 2);
    };
    auto slice = [&](int sl, const Addr& AD) __attribute__((always_inline)) {
        constexpr int c0 = TB == 1 ? 34 : 76, l0 = FL ? c0 + kNPV : (TB == 1 ? 38 : 86);   // (FL commits 5 quads: slots 34-38, its loads follow at 39-43, the affine at 44)
        if (sl >= 10 && sl < 33) transform_slice(sl - 10, AD.pn, AD.vn);
        else if (TB == 2 && sl >= 46 && sl < 69) transform_slice(sl - 46, AD.pn + kPassSrc, AD.vn + kPassDst);
        else if (sl >= c0 && sl < c0 + kNPV) {
            if (!(FS_W4T_ABL & 4)) commit_patch_one(AD.pc[sl - c0], sl - c0);
        } else if (sl >= l0 && sl < l0 + kNPV) {
            if (!(FS_W4T_ABL & 4)) issue_patch_one(CP.I, CP.chunk, sl - l0);
        } else if (sl == l0 + kNPV) {
            if (!(FS_W4T_ABL & 4)) issue_affine(CP.I, CP.chunk);
        }
    };
#define FS_W4T_MFMA(pos, tb, av, bv)                                                  \
    do {                                                                              \
        if constexpr ((pos) < kNAcc) FS_W4_MFMA_A(acc[(pos) < kNAcc ? (pos) : 0][tb], av, bv);       \
        else FS_W4_MFMA_V(accv[(pos) >= kNAcc ? (pos) - kNAcc : 0][tb], av, bv);       \
    } while (0)
    Item cur_I = CU.I;     // the step being multiplied (M = 3 reloads the second half of ITS filter quads during the sweep)
    int cur_chunk = 0;
    auto sweep = [&](const Addr& AD) __attribute__((always_inline)) {
        float B[3][TB];
        auto read_b = [&](int slot, int into) __attribute__((always_inline)) {
            if constexpr (TB == 1) {
                B[into][0] = FS_W4_LDS(float, AD.pb + slot * (kVB * 4));
            } else {   // both tile blocks in one 8-byte read (16-bit byte offset: no address arithmetic, which a ds_read2_b32 pair would need)
                const f32x2 v = FS_W4_LDS(f32x2, AD.pb + slot * (kVB * 4));
                B[into][0] = v.x;
                B[into][TB - 1] = v.y;
            }
        };
        read_b(0, 0);
        read_b(1, 1);
        fs_static_for<0, 72>([&](auto SLOT) __attribute__((always_inline)) {
            constexpr int s = decltype(SLOT)::value;
            constexpr int pos = s % 36, c = s % 3, n2 = (s + 2) % 3;
            if constexpr (CB2) {
                // quad g = s / 2 of the step's 36 holds slots 2g, 2g+1 for both channel blocks; it lives in register set g % 18 and is replaced,
                // right behind its last use, by quad (g + 18) % 36 -- of THIS step for g < 18, of the next step beyond
                constexpr int g = s >> 1, e0 = (s & 1) * 2;
                FS_W4T_MFMA(pos, 0, quad_elem<e0>(uv[g % 18]), B[c][0]);
                __builtin_amdgcn_sched_barrier(0);
                slice(2 * s, AD);
                __builtin_amdgcn_sched_barrier(0);
                FS_W4T_MFMA(pos, 1, quad_elem<e0 + 1>(uv[g % 18]), B[c][0]);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 2 < 72 && !(FS_W4T_ABL & 8)) read_b(s + 2, n2);
                slice(2 * s + 1, AD);
                if ((s & 1) == 1 && !(FS_W4T_ABL & 2)) {
                    if constexpr (g < 18) issue_filter_q(cur_I, cur_chunk, g + 18, g, 1u, uvo);
                    else issue_filter_q(CU.I, CU.chunk, g - 18, g - 18, load_on, uvo_eff);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                const float av = quad_elem<(s & 3)>(uv[s >> 2]);
                FS_W4T_MFMA(pos, 0, av, B[c][0]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (TB == 2) {
                    slice(2 * s, AD);
                    __builtin_amdgcn_sched_barrier(0);
                    FS_W4T_MFMA(pos, TB - 1, av, B[c][TB - 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (s + 2 < 72 && !(FS_W4T_ABL & 8)) read_b(s + 2, n2);   // operands two slots ahead
                slice(TB * s + TB - 1, AD);
                if ((s & 3) == 3 && !(FS_W4T_ABL & 2)) issue_filter_one(CU.I, CU.chunk, s >> 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };

    // ---- epilogue of one item.  Lane (j = lane & 15, g = lane >> 4) of wave w holds, per tile block, tile 16 tb + j and the four
    // channels co0 + 16 w + 4 g .. + 3 of all 36 positions.  Pixels outside the image carry the out-of-range offset (loads 0, stores dropped).
    auto relu1 = [](float x) __attribute__((always_inline)) {   // ONE v_max_f32 (fmaxf comes with a canonicalising second instruction)
#if defined(__HIP_DEVICE_COMPILE__)
        float r;
        asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
        return r;
#else
        return x > 0.f ? x : 0.f;
#endif
    };
    // sum over the 16 lanes of a row (= the 16 tiles of a tile block), result in every lane.  On the GPU four DPP adds (quad_perm xor 1, xor 2, then
    // row_half_mirror and row_mirror: after two steps a quad's lanes are equal, so "the mirrored lane" carries the other quad's / half's sum --
    // the same pairs as the xor butterfly, bit-identical to it) instead of four ds_bpermute round trips.
    auto row16_sum = [](float v) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
        return v;
#else
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m);
        return v;
#endif
    };
    auto epilogue_body = [&](auto FULLT, const Item& I) __attribute__((always_inline)) {
        constexpr bool full = decltype(FULLT)::value;
        int ln = lane;
        FS_W4_PIN(ln);
        const int j = ln & 15;
        const float* yb = a.y + ((size_t)I.n + (ks > 1 ? (size_t)I.z * a.N : 0)) * a.Ho * a.Wo * a.Cout;
        const unsigned img_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * a.Cout) * 4u);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yb)), 0, img_bytes, 0x00020000);
        const unsigned rowp4 = __builtin_amdgcn_readfirstlane((unsigned)(a.Wo * a.Cout) * 4u), col4 = __builtin_amdgcn_readfirstlane((unsigned)a.Cout * 4u);
        auto soff = [&](int px) __attribute__((always_inline)) { return (unsigned)(px >> 2) * rowp4 + (unsigned)(px & 3) * col4; };
        // second source of the epilogue: the residual gradient (EPI 2: [N][Ho - 2 add_pad][Wo - 2 add_pad][Cout], added where it exists) or the
        // consumer's ReLU mask (EPI 4: [N][Ho][Wo][Cout])
        const int Ha = kAdd ? a.Ho - 2 * a.add_pad : a.Ho, Wa = kAdd ? a.Wo - 2 * a.add_pad : a.Wo;
        const int apad = kAdd ? a.add_pad : 0;
        const float* adn = kAdd ? a.add_src + (size_t)I.n * Ha * Wa * a.Cout : (EPI == 4 ? a.mask_src + (size_t)I.n * Ha * Wa * a.Cout : yb);
        // EPI 5 / 6: z of the unit whose output gradient this is, [N][Ho][Wo][Cout]
        const __amdgpu_buffer_rsrc_t zr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(kInb ? a.inb_z + (size_t)I.n * a.Ho * a.Wo * a.Cout : yb)), 0, img_bytes, 0x00020000);
        const unsigned add_bytes = __builtin_amdgcn_readfirstlane((unsigned)(Ha * Wa * a.Cout) * 4u);
        const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(adn)), 0, add_bytes, 0x00020000);
        const bool pool = EPI == 3 && a.pool_out != nullptr;
        float cs[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};   // EPI 1
        fs_static_for<0, NB>([&](auto TBI) __attribute__((always_inline)) {
            constexpr int tb = decltype(TBI)::value;          // accumulator block: tile block (M = 2) or channel block (M = 3)
            const int t = CB2 ? j : 16 * tb + j;
            const int co = I.cob * kBNi + wave * (CB2 ? 32 : 16) + (CB2 ? 16 * tb : 0) + 4 * (ln >> 4);
            float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (EPI == 3 && a.bias) bs = *reinterpret_cast<const float4*>(a.bias + co);
            const float bsv[4] = {bs.x, bs.y, bs.z, bs.w};
            int oy = I.oy0 + 4 * (TB == 1 ? (t >> 2) : (t >> 3)), ox = I.ox0 + 4 * (TB == 1 ? (t & 3) : (t & 7));
            bool tile_ok = true;
            if constexpr (FL) {   // tile 16 br + t of the sample's row-major tile grid
                const int tg = 16 * I.br + t;
                const int tyg = fdiv(tg, inv_gtx);
                oy = 4 * tyg;
                ox = 4 * (tg - tyg * gTx);
                tile_ok = tg < gT;
            }
            const unsigned obase = (unsigned)((oy * a.Wo + ox) * a.Cout + co) * 4u;
            const int ry = tile_ok ? a.Ho - oy : 0, cx = tile_ok ? a.Wo - ox : 0;   // valid rows / columns of the lane's tile (edge blocks; a tile past the grid: none)
            auto inside = [&](int px) __attribute__((always_inline)) { return full || ((px >> 2) < ry && (px & 3) < cx); };
            auto voff = [&](int px) __attribute__((always_inline)) { return inside(px) ? obase : kOOB; };
            float4 ad[16];
            auto add_load = [&](int px) __attribute__((always_inline)) {
                if (EPI == 5) {
                    ad[px] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(zr, voff(px), soff(px), 0));
                } else if (kAdd) {
                    const int ay = oy + (px >> 2) - apad, ax = ox + (px & 3) - apad;
                    const bool ok = ay >= 0 && ay < Ha && ax >= 0 && ax < Wa;
                    ad[px] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ar, ok ? (unsigned)((ay * Wa + ax) * a.Cout + co) * 4u : kOOB, 0, 0));
                } else {
                    ad[px] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ar, voff(px), soff(px), 0));
                }
            };
            float4 zz[EPI == 6 ? 16 : 1];   // EPI 6: z beside the residual-gradient quads (EPI 5 keeps z in ad[])
            auto z_load = [&](int px) __attribute__((always_inline)) {
                if (EPI == 6) zz[px] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(zr, voff(px), soff(px), 0));
            };
            if (kAdd || EPI == 4 || EPI == 5) {
#pragma unroll
                for (int px = 0; px < kEarly; ++px) add_load(px);
#pragma unroll
                for (int px = 0; px < kEarly; ++px) z_load(px);
            }
            __builtin_amdgcn_sched_barrier(0);
            float o[16][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s[4][6];   // A^T M: rows 0..3, columns 0..5
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const float m0 = FS_W4T_ACC(q, tb, r), m1 = FS_W4T_ACC(6 + q, tb, r), m2 = FS_W4T_ACC(12 + q, tb, r), m3 = FS_W4T_ACC(18 + q, tb, r),
                                m4 = FS_W4T_ACC(24 + q, tb, r), m5 = FS_W4T_ACC(30 + q, tb, r);   // (each element read ONCE: the reads are volatile)
                    FS_W4_AT(m0, m1, m2, m3, m4, m5, s[0][q], s[1][q], s[2][q], s[3][q]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    FS_W4_AT(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], o[4 * i][r], o[4 * i + 1][r], o[4 * i + 2][r], o[4 * i + 3][r]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kAdd || EPI == 4 || EPI == 5) {
#pragma unroll
                for (int px = kEarly; px < 16; ++px) add_load(px);
#pragma unroll
                for (int px = kEarly; px < 16; ++px) z_load(px);
            }
            // EPI 5 / 6: the per-(sample, channel) constants of the lane's four channels
            float4 ib_m = make_float4(0.f, 0.f, 0.f, 0.f), ib_a = ib_m, ib_b = make_float4(1.f, 1.f, 1.f, 1.f);
            if (kInb) {
                const size_t k = (size_t)I.n * a.Cout + co;
                ib_m = *reinterpret_cast<const float4*>(a.inb_mean + k);
                if (a.inb_relu) {
                    ib_a = *reinterpret_cast<const float4*>(a.inb_a + k);
                    ib_b = *reinterpret_cast<const float4*>(a.inb_b + k);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kInb) {
                // instance-norm backward of the unit that produced inb_z, partial sums of the item: s1 = sum g', s2 = sum g' (z - mean) (x rstd below),
                // g' = g where the unit's activation passed (relu(a z + b) > 0; a = 0, b = 1 without a ReLU).  All of it IN FRONT of the item's
                // stores: a load result needed while stores are in flight costs `s_waitcnt vmcnt(0)` (one counter, out of order between the
                // two kinds) -- i.e. the write burst of all 256 workgroups: measured +4.5 us per item with the sums behind the stores.
                const float mq[4] = {ib_m.x, ib_m.y, ib_m.z, ib_m.w}, aq[4] = {ib_a.x, ib_a.y, ib_a.z, ib_a.w}, bq[4] = {ib_b.x, ib_b.y, ib_b.z, ib_b.w};
#pragma unroll
                for (int px = 0; px < 16; ++px) {
                    const float4 z4 = EPI == 6 ? zz[EPI == 6 ? px : 0] : ad[px];
                    const float zq[4] = {z4.x, z4.y, z4.z, z4.w};
                    const float av4[4] = {ad[px].x, ad[px].y, ad[px].z, ad[px].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (kAdd) o[px][r] += av4[r];
                        const bool keep = inside(px) && fmaf(zq[r], aq[r], bq[r]) > 0.f;
                        const float gq = keep ? o[px][r] : 0.f;
                        s1[r] += gq;
                        s2[r] = fmaf(gq, zq[r] - mq[r], s2[r]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // (EPI 3 with a.y_keep_n: this sample's full-resolution result is read by nobody -- only its pooled tensor is stored; wave-uniform)
            const bool keep_y = !(EPI == 3 && pool && a.y_keep_n > 0 && I.n >= a.y_keep_n);
#pragma unroll
            for (int px = 0; px < 16; ++px) {
                const float av4[4] = {ad[px].x, ad[px].y, ad[px].z, ad[px].w};
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = o[px][r];
                    if (kAdd && !kInb) v[r] += av4[r];
                    if (EPI == 3) v[r] = a.out_relu ? relu1(v[r] + bsv[r]) : v[r] + bsv[r];
                    if (EPI == 4) v[r] = av4[r] > 0.f ? v[r] : 0.f;
                    if (EPI == 3) o[px][r] = v[r];   // (the pool reads the stored values)
                }
                if (EPI != 3 || keep_y)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fs_u32x4, make_float4(v[0], v[1], v[2], v[3])), yr, voff(px), soff(px), 0);
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
                                                               (unsigned)wy * prow4 + (unsigned)wx * col4, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (EPI == 1) {
                // per-item instance-norm partials of the RAW output {mean, M2, count} around a shift (the block's first pixel): the lane's
                // 16 pixels per tile block here -- BEHIND the stores, whose way to memory these ~200 instructions cover --, the 16 tiles
                // of a block (= the 16 lanes of a row) below
                if (tb == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) cs[r] = __shfl(o[0][r], ln & 48);
                }
#pragma unroll
                for (int px = 0; px < 16; ++px)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dv = inside(px) ? o[px][r] - cs[r] : 0.f;
                        s1[r] += dv;
                        s2[r] = fmaf(dv, dv, s2[r]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (kInb) {   // the 16 tiles of the item (= the 16 lanes of a row), fixed order; lane j = 0 of a row writes the item's record of its four channels
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[r] = row16_sum(s1[r]);
                s2[r] = row16_sum(s2[r]);
            }
            if (j == 0) {
                const int co = I.cob * kBNi + wave * 16 + 4 * (ln >> 4);
                const float4 rs = *reinterpret_cast<const float4*>(a.inb_rstd + (size_t)I.n * a.Cout + co);
                float* rc = a.inb_rec + ((size_t)(I.n * blocks + I.br) * a.Cout + co) * 2;
                *reinterpret_cast<float4*>(rc) = make_float4(s1[0], s2[0] * rs.x, s1[1], s2[1] * rs.y);
                *reinterpret_cast<float4*>(rc + 4) = make_float4(s1[2], s2[2] * rs.z, s1[3], s2[3] * rs.w);
            }
        }
        if (EPI == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[r] = row16_sum(s1[r]);
                s2[r] = row16_sum(s2[r]);
            }
            float cnt_fl = 0.f;
            if constexpr (FL) {   // pixels of the item: the lanes' valid tile areas, summed over the 16 tiles of a row
                const int tg = 16 * I.br + j;
                const int tyg = fdiv(tg, inv_gtx);
                const int ryv = tg < gT ? min(4, a.Ho - 4 * tyg) : 0, cxv = tg < gT ? min(4, a.Wo - 4 * (tg - tyg * gTx)) : 0;
                cnt_fl = row16_sum((float)(ryv * cxv));
            }
            if (j == 0) {
                const int th_valid = min(kBH, a.Ho - I.oy0), tw_valid = min(kBW, a.Wo - I.ox0);
                const float cnt = FL ? cnt_fl : (float)(th_valid * tw_valid);
                const int co = I.cob * kBNi + wave * 16 + 4 * (ln >> 4);
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
        FS_W4_MFMA_SETTLE();   // (the item's last inline-assembly matrix instructions have written their accumulators: fs_wino4.h)
        if (FL ? (!(a.Ho & 3) && !(a.Wo & 3) && 16 * I.br + 16 <= gT) : (I.oy0 + kBH <= a.Ho && I.ox0 + kBW <= a.Wo))
            epilogue_body(std::true_type{}, I);
        else
            epilogue_body(std::false_type{}, I);
    };
    // (No vector-memory drain between items or behind the prologue, unlike fs_wino4.hip: with the filter quads consumed one sweep
    // after their loads the compiler keeps exact vmcnt counts in the sweep either way (checked in the ISA: vmcnt(20) / vmcnt(22)), and
    // an exact count is safe with the epilogue's stores still in flight -- loads return in order among themselves, so "at most N
    // operations outstanding" implies that a load with N younger LOADS behind it has arrived, whatever the stores do.)
    // kDefer (the 32-tile forms): the LAST sweep of an item loads nothing: its staging loads run against empty buffers (load_on = 0: zeros, no traffic -- ONE copy of the
    // sweep, and two copies would make the register allocator shuffle the accumulators between their assignments), and the filter quads of
    // the next item's first step and the patch of the step after next go out here, behind the epilogue.  What the last sweep
    // "loaded" is thereby dead during the output transform: 72 + 4 kNPV registers free (with them live the 32-tile forms spill ~50
    // registers to scratch memory around every epilogue).  The filter latency hides behind the zeroing of the accumulators, the
    // patch is not needed before the middle of the next sweep.
    auto next_item = [&]() __attribute__((always_inline)) {
        if constexpr (kDefer) {
            load_on = 1u;
            uvo_eff = CU.live ? uvo : kOOB;
#pragma unroll
            for (int i = 0; i < 18; ++i) issue_filter_one(CU.I, CU.chunk, i);
            cursor_next(CU);
#pragma unroll
            for (int i = 0; i < kNPV; ++i) issue_patch_one(CP.I, CP.chunk, i);
            issue_affine(CP.I, CP.chunk);
        }
        zero_acc();
        FS_W4_MFMA_SETTLE();   // (the zeroed accumulators are the next sweep's SrcC)
    };

    // ---- prologue: step 0 complete in stage 0 (patch, V) and in registers (filter), the patch of step 1 in stage 1, the patch of step 2
    // in registers.  Every load of steps 0 and 1 goes out before the first wait (patches first: they are needed first, and
    // the counter retires in order), the accumulators are zeroed while they fly.
    const Addr AP0 = stage_addrs(kStageF, 0), AP1 = stage_addrs(0, kStageF);   // "next stage" = stage 0 / stage 1
    patch_offsets(CP.I, 1);
    float4 pv0[kNPV], fa0 = fa, fb0 = fb;   // step 0's patch and scale / shift: registers of their own, so that step 1's loads need not wait for them
#pragma unroll
    for (int i = 0; i < kNPV; ++i) issue_patch_into(pv0[i], CP.I, CP.chunk, i);
    issue_affine_into(fa0, fb0, CP.I, CP.chunk);
    if (cursor_next(CP) || !CP.live) patch_offsets(CP.I, CP.live);
#pragma unroll
    for (int i = 0; i < kNPV; ++i) issue_patch_one(CP.I, CP.chunk, i);
    issue_affine(CP.I, CP.chunk);
#pragma unroll
    for (int i = 0; i < 18; ++i) issue_filter_one(CU.I, CU.chunk, i);
    zero_acc();
    FS_W4_MFMA_SETTLE();
#pragma unroll
    for (int i = 0; i < kNPV; ++i) commit_quad(AP1.pc[i], pv0[i], fa0, fb0);   // (AP1's current stage is stage 0)
#pragma unroll
    for (int i = 0; i < kNPV; ++i) commit_patch_one(AP0.pc[i], i);   // stage 1's patch area
    if (cursor_next(CP) || !CP.live) patch_offsets(CP.I, CP.live);
#pragma unroll
    for (int i = 0; i < kNPV; ++i) issue_patch_one(CP.I, CP.chunk, i);
    issue_affine(CP.I, CP.chunk);
    cursor_next(CU);   // the filter cursor now points at step 1
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < TB; ++pass) {
        transform_read(AP0.pn + pass * kPassSrc, 0, 18);
        transform_rows();
        transform_swap();
        transform_cols();
        transform_write(AP0.vn + pass * kPassDst, 0, 18);
    }
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
        for (int chunk = cur_it.cbeg; chunk < cur_it.cend; ++chunk, ++q) {
            load_on = (!kDefer || chunk + 1 < cur_it.cend) ? 1u : 0u;
            cur_I = cur_it;
            cur_chunk = chunk;
            uvo_eff = CU.live ? uvo : kOOB;
            if (cursor_next(CP) || !CP.live) patch_offsets(CP.I, CP.live);   // (the offsets change once per item)
#ifdef FS_WINO4T_TRACE
            const long long q0 = FS_W4T_NOW();
#endif
            const int o0 = (q & 1) ? kStageF : 0, o1 = kStageF - o0;
            const Addr AD = stage_addrs(o0, o1);
            __builtin_amdgcn_sched_barrier(0);
            sweep(AD);
            if (load_on) cursor_next(CU);   // (behind an item's last sweep the cursor moves in next_item, after the loads it describes)
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
#undef FS_W4T_ACC
#undef FS_W4T_MFMA
}

template <int M, int EPI, bool AFF>
static int wino4t_launch_as(const ConvArgs& a, long grid, hipStream_t s) {
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(wino4t_conv_kernel<M, EPI, AFF>));
    hipLaunchKernelGGL((wino4t_conv_kernel<M, EPI, AFF>), dim3((unsigned)grid), dim3(256), (size_t)a.p.lds_bytes, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// the instantiations of one tile-block count are split over two translation units (A: on-load norm forms + raw + statistics; B: residual
// gradient, bias / ReLU / pool, consumer mask), so that no single compile carries more than four of them
template <int TB>
static int wino4t_launch_part_a(const ConvArgs& a, int epi, long grid, hipStream_t s) {
    if (a.in_a) return epi == 1 ? wino4t_launch_as<TB, 1, true>(a, grid, s) : wino4t_launch_as<TB, 0, true>(a, grid, s);
    return epi == 1 ? wino4t_launch_as<TB, 1, false>(a, grid, s) : wino4t_launch_as<TB, 0, false>(a, grid, s);
}
template <int TB>
static int wino4t_launch_part_b(const ConvArgs& a, int epi, long grid, hipStream_t s) {
    switch (epi) {
        case 2: return wino4t_launch_as<TB, 2, false>(a, grid, s);
        case 3: return wino4t_launch_as<TB, 3, false>(a, grid, s);
        default: return wino4t_launch_as<TB, 4, false>(a, grid, s);
    }
}
// 16-tile items with the instance-norm-backward partial sums (EPI 5 raw, 6 residual gradient): fs_wino4t1d.hip
template <int TB>   // (a template like the other parts: instantiated only by the translation unit that calls it)
static int wino4t_launch_part_d(const ConvArgs& a, int epi, long grid, hipStream_t s) {
    static_assert(TB == 1, "16-tile items only");
    return epi == 5 ? wino4t_launch_as<1, 5, false>(a, grid, s) : wino4t_launch_as<1, 6, false>(a, grid, s);
}
// the flattened 16-tile form (M = 4): the transform net's epilogues only -- fs_wino4t4a.hip (on-load norm forms + raw + statistics), fs_wino4t4b.hip
// (residual gradient, instance-norm-backward partial sums)
template <int M>
static int wino4t_launch_part_fa(const ConvArgs& a, int epi, long grid, hipStream_t s) {
    static_assert(M == 4, "the flattened form");
    if (a.in_a) return epi == 1 ? wino4t_launch_as<4, 1, true>(a, grid, s) : wino4t_launch_as<4, 0, true>(a, grid, s);
    return epi == 1 ? wino4t_launch_as<4, 1, false>(a, grid, s) : wino4t_launch_as<4, 0, false>(a, grid, s);
}
template <int M>
static int wino4t_launch_part_fb(const ConvArgs& a, int epi, long grid, hipStream_t s) {
    static_assert(M == 4, "the flattened form");
    switch (epi) {
        case 2: return wino4t_launch_as<4, 2, false>(a, grid, s);
        case 5: return wino4t_launch_as<4, 5, false>(a, grid, s);
        case 6: return wino4t_launch_as<4, 6, false>(a, grid, s);
        default: return -7;
    }
}
// the 128-channel item form (M = 3): raw (split-K partials, input gradients in front of a pool), bias + ReLU (+ pool), consumer mask
template <int M>
static int wino4t_launch_part_c(const ConvArgs& a, int epi, long grid, hipStream_t s) {
    static_assert(M == 3, "the 128-channel form");
    switch (epi) {
        case 0: return wino4t_launch_as<3, 0, false>(a, grid, s);
        case 3: return wino4t_launch_as<3, 3, false>(a, grid, s);
        case 4: return wino4t_launch_as<3, 4, false>(a, grid, s);
        default: return -7;
    }
}

}  // namespace fs
