// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores, HALF items: the small-grid sibling of fs_wino2.hip for the ten
// 64-channel residual convs of the transform net and their input gradients (reference im_transf_net.py:250-276) when a
// launch cannot fill the chip with 64-tile items -- batch 4 per GPU (BASELINE configs[2] / [3]): 4 x 36 = 144 items of
// 16x16 pixels for 256 CUs, where the direct kernel (conv_igemm<32,1,1>, 34 us, 48 TFLOP/s) used to stay.
//
// An item here is <= 32 tiles (a BH x BW block of 2x2-pixel tiles, BH*BW <= 32; the plan picks the shape that gives the
// fewest blocks: 4x8 or 5x6) x 64 output channels, so the same layer makes twice as many items, and the four waves split the
// item by CHANNEL BLOCK x WINOGRAD-POSITION ROW PAIR instead of tile block x channel block:
//     wave (nb, ph): channels 32 nb .. 32 nb + 31, positions 8 ph .. 8 ph + 7 (rows 2 ph, 2 ph + 1 of the 4x4 position grid)
// -- 8 accumulator blocks of 32x32 = 128 registers, 32 matrix instructions per 8-channel chunk and wave (half of fs_wino2's),
// no split over the reduction dimension.  The output transform Y = A^T M A is linear in the position rows: each wave applies
// the column stage to its two rows and forms its share of the two output rows (ph 0: R0 + R1 | R1;  ph 1: R2 | -R2 - R3);
// the two shares of a (tile, channel) meet ONCE through LDS, each wave finishing half of the tile rows.  Everything else is
// the fs_wino2 recipe: one wave per SIMD, persistent workgroups over a strided item list as one flat software pipeline over
// (item, chunk) steps (filter of step q+1 and patch of step q+2 in flight), K-contiguous operands in LDS
// (V[pos][k/4][tile][4], U[pos][k/4][channel][4]: one conflict-free 16-byte read feeds four matrix instructions), the input
// transform of the next step and the filter commit threaded through the sweep in sched_barrier-pinned slots (packed fp32; the filter
// is requested one whole step before it is committed -- a sweep here is too short to cover a load's round trip),
// producer instance norm + ReLU applied while the patch is committed, per-item instance-norm partials in the epilogue.
#include "fs_kernels.h"

#include <cstdlib>
#include <type_traits>

namespace fs {

namespace {
constexpr int kNTh = 32;             // tiles per item (MFMA rows)
constexpr int kCC = 8;               // input channels per chunk
constexpr int kPS = kCC + 1;         // patch pixel pitch
constexpr int kBN = 64;              // output channels per item
constexpr int kMaxPatchPx = 180;     // 10 x 18 (4x8 tiles), 12 x 14 = 168 (5x6), 18 x 10 (8x4)
constexpr int kPatchF = ((kMaxPatchPx * kPS + 8 + 3) & ~3);
constexpr int kVF = 16 * kNTh * kCC;   // 4096
constexpr int kUF = 16 * kBN * kCC;    // 8192
constexpr int kStageF = kPatchF + kVF + kUF;
constexpr int kXF = 2 * 2 * 8 * 4 * 64;   // exchange of the output-row shares: [nb][destination ph][row][value][lane]
constexpr int kRedF = 64 * 2 * 4 + 64;    // statistics scratch: [4 contributors][64][2] + shift[64]
constexpr unsigned kOOB = 0x80000000u;
}  // namespace

#ifdef FS_WINO2H_TRACE
// debug build only (tools/w2h_trace.py): per-workgroup phase cycle counts of the last launch
__device__ long long g_wino2h_trace[4096 * 8];
extern "C" int fs_debug_conv_trace(long long* out, int n_wg) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino2h_trace), sizeof(long long) * 8 * (size_t)n_wg, 0, hipMemcpyDeviceToHost);
}
extern "C" int fs_debug_conv_trace_reset() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wino2h_trace)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(long long) * 8 * 4096);
}
#define FS_W2H_NOW() ((long long)__builtin_readcyclecounter())
#endif

__global__ __launch_bounds__(256) void wino2h_conv_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
#ifdef FS_WINO2H_TRACE
    const long long tr_t0 = FS_W2H_NOW();
    long long tr_sweep = 0, tr_bar = 0, tr_epi = 0, tr_pro = 0, tr_commit = 0;
#endif
    const ConvPlan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, kq = lane >> 5;
    const int nb = wave & 1, ph = wave >> 1;   // this wave's 32-channel block / pair of position rows
    const int BH = p.TH >> 1, BW = p.TW >> 1;  // tiles per block side
    const int PW = p.PW;                       // patch width in pixels (2 BW + 2)
    const int ntile = BH * BW;                 // real tiles of a block (<= 32)
    auto fdiv = [](int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); };
    auto uniform_ptr = [](const float* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };
    float* const xch = smem + 2 * kStageF;
    float* const red = xch + kXF;
    const float inv_bw = 1.0f / (float)BW, inv_pw = 1.0f / (float)PW;

    // ---- the item list of this workgroup: item = (n * blocks + block) * ncob + channel block
    const int blocks = p.tiles_y * p.tiles_x;
    const int ncob = a.Cout / kBN;
    const int nchunks = a.Cin / kCC;
    const int total_items = a.N * blocks * ncob;
    const int G = (int)gridDim.x;
    const int my_items = ((int)blockIdx.x < total_items) ? (total_items - 1 - (int)blockIdx.x) / G + 1 : 0;
    const float inv_ncob = 1.0f / (float)ncob, inv_blocks = 1.0f / (float)blocks, inv_tx = 1.0f / (float)p.tiles_x;
    struct Item {
        int n, oy0, ox0, co0, tile_lin;
    };
    auto decode = [&](int it) {
        Item r;
        const int lin = (int)blockIdx.x + it * G;
        const int t2 = fdiv(lin, inv_ncob);
        const int cob = lin - t2 * ncob;
        r.tile_lin = t2;
        r.n = fdiv(t2, inv_blocks);
        const int br = t2 - r.n * blocks;
        const int byi = fdiv(br, inv_tx);
        r.oy0 = byi * p.TH;
        r.ox0 = (br - byi * p.tiles_x) * p.TW;
        r.co0 = cob * kBN;
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.oy0 = __builtin_amdgcn_readfirstlane(r.oy0);
        r.ox0 = __builtin_amdgcn_readfirstlane(r.ox0);
        r.co0 = __builtin_amdgcn_readfirstlane(r.co0);
        r.tile_lin = __builtin_amdgcn_readfirstlane(r.tile_lin);
        return r;
    };

    // ---- staging descriptors.  patch: PH*PW pixels x 2 float4 <= 360 elements, <= 2 per thread
    const int npx = p.PH * PW;
    int pq[2], pdst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * 256;
        pq[i] = -1;
        pdst[i] = kMaxPatchPx * kPS;   // sink
        if (e < npx * 2) {
            const int pix = e >> 1, c4 = e & 1;
            const int py = fdiv(pix, inv_pw), px = pix - py * PW;
            pq[i] = (py << 8) | px;
            pdst[i] = pix * kPS + c4 * 4;
        }
    }
    const int pc4 = (tid & 1) * 4;
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * a.Cin) * 4u);
    const unsigned u_bytes = __builtin_amdgcn_readfirstlane((unsigned)(16 * a.Cin * a.Cout) * 4u);
    const bool has_ab = a.in_a != nullptr;
    const float* ub = uniform_ptr(a.w_wino2);
    float4 pv[2], uv[8];
    const __amdgpu_buffer_rsrc_t ur_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, u_bytes, 0x00020000);
    float4 va = make_float4(1.f, 1.f, 1.f, 1.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned gvo[2];
    unsigned uvo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + i * 256;
        uvo[i] = (unsigned)((e >> 7) * a.Cin * a.Cout + ((e >> 6) & 1) * a.Cout * 4 + (e & 63) * 4) * 4u;
    }
    auto item_offsets = [&](const Item& I) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int py = pq[i] >> 8, px = pq[i] & 255;
            const int sy = I.oy0 - a.pad_t + py, sx = I.ox0 - a.pad_l + px;
            const bool ok = pq[i] >= 0 && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
            gvo[i] = ok ? (unsigned)((sy * a.W + sx) * a.Cin + pc4) * 4u : kOOB;
        }
    };
    // (every global load of the pipeline is issued UNCONDITIONALLY -- a step that does not exist reads through the out-of-range
    // offset, which the hardware answers with zeros without touching memory -- so that the compiler can count the loads in
    // flight exactly: behind a branch it falls back to s_waitcnt vmcnt(0) before the filter commit, i.e. waits for loads it
    // issued a few instructions earlier)
    const unsigned ab_bytes = __builtin_amdgcn_readfirstlane(has_ab ? (unsigned)(a.N * a.in_nstride) * 4u : 0u);
    const __amdgpu_buffer_rsrc_t ar_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(has_ab ? a.in_a : a.x)), 0, ab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t br_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(has_ab ? a.in_b : a.x)), 0, ab_bytes, 0x00020000);
    auto issue_patch = [&](const Item& I, int chunk, int live) {
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * a.Cin);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, x_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, live ? gvo[i] : kOOB, chunk * kCC * 4, 0));
        const unsigned abo = (live && has_ab) ? (unsigned)(I.n * a.in_nstride + chunk * kCC + pc4) * 4u : kOOB;
        va = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ar_k, abo, 0, 0));
        vb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(br_k, abo, 0, 0));
    };
    auto issue_filter_pair = [&](const Item& I, int chunk, int i0) {
        const unsigned so = (unsigned)((chunk * 2 * a.Cout + I.co0) * 4) * 4u;
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) uv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ur_k, uvo[i], so, 0));
    };
    auto commit_patch = [&](float* patch) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 v = pv[i];
            if (has_ab) {   // (pad 0 only: every patch pixel that reaches a stored output is a real pixel)
                v.x = fmaf(v.x, va.x, vb.x);
                v.y = fmaf(v.y, va.y, vb.y);
                v.z = fmaf(v.z, va.z, vb.z);
                v.w = fmaf(v.w, va.w, vb.w);
                if (a.in_relu) {
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f);
                    v.w = fmaxf(v.w, 0.f);
                }
            }
            float* d = patch + pdst[i];
            d[0] = v.x;
            d[1] = v.y;
            d[2] = v.z;
            d[3] = v.w;
        }
    };
    auto commit_filter = [&](float* Ul) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(Ul + (tid + i * 256) * 4) = uv[i];
    };
    // input transform V = B^T d B: thread = (tile tt, channel tk) of the chunk, ONE pair per chunk
    const int tt_t = tid >> 3, tk_t = tid & 7;
    int tsrc_off;   // patch offset of the thread's tile (tiles beyond the block read the block's first tile: their rows are never stored)
    {
        const int ttc = tt_t < ntile ? tt_t : 0;
        const int tty = fdiv(ttc, inv_bw), ttx = ttc - tty * BW;
        tsrc_off = ((2 * tty) * PW + 2 * ttx) * kPS + tk_t;
    }
    const int tdst_off = (tk_t >> 2) * (kNTh * 4) + tt_t * 4 + (tk_t & 3);
    auto transform_now = [&](const float* patch, float* Vl) {
        const float* src = patch + tsrc_off;
        float d[4][4], r[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = src[(i * PW + j) * kPS];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[0][j] = d[0][j] - d[2][j];
            r[1][j] = d[1][j] + d[2][j];
            r[2][j] = d[2][j] - d[1][j];
            r[3][j] = d[1][j] - d[3][j];
        }
        float* dst = Vl + tdst_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[(i * 4 + 0) * kNTh * kCC] = r[i][0] - r[i][2];
            dst[(i * 4 + 1) * kNTh * kCC] = r[i][1] + r[i][2];
            dst[(i * 4 + 2) * kNTh * kCC] = r[i][2] - r[i][1];
            dst[(i * 4 + 3) * kNTh * kCC] = r[i][1] - r[i][3];
        }
    };

    f32x16 acc[8];   // one 32x32 block per Winograd position of this wave (positions 8 ph + j)
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    };
    zero_acc();

    f32x2 td[4][2];   // the 4x4 input block of the thread's (tile, channel) pair of the NEXT step, as column pairs
    int has1 = 0, has2 = 0, load_live = 0;
    Item L = decode(0);          // cursor of the load stream: the step whose loads were issued last
    int l_it = 0, l_chunk = 0;
    auto advance_load = [&]() {
        if (++l_chunk < nchunks) return true;
        if (++l_it >= my_items) return false;
        L = decode(l_it);
        l_chunk = 0;
        item_offsets(L);
        return true;
    };
    // 24 slots per sweep of step q.  0: the load cursor moves to step q+2 and its patch loads go out; 2-5: LDS reads of the 4x4
    // input block of step q+1 (one row per slot); 9: its transform arithmetic + stores in ONE gap; 12-19: the filter of
    // step q+1 -- loaded during the PREVIOUS sweep -- goes to LDS one 16-byte row per slot, and the register it frees is
    // reloaded with the filter of step q+2 at once.  (A sweep here is 32 matrix instructions, ~2k cycles: shorter than the
    // round trip of a filter load under 250 workgroups streaming 32 KB per step, so -- unlike fs_wino2 -- a filter has to be
    // requested a whole step before it is committed; the shallow fs_wino2 order measured 25.6 us per launch at batch 4.)
    auto slice = [&](int sl, const float* patch_n, float* Vn, float* Un) {
        if (sl == 0) {
            has2 = 0;
            if (has1 && load_live) {
                has2 = advance_load() ? 1 : 0;
                load_live = has2;
            }
            issue_patch(L, l_chunk, has2);
            return;
        }
        if (sl >= 12 && sl < 20) {
            const int i0 = sl - 12;
            *reinterpret_cast<float4*>(Un + (tid + i0 * 256) * 4) = uv[i0];
            const unsigned so = (unsigned)((l_chunk * 2 * a.Cout + L.co0) * 4) * 4u;
            uv[i0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ur_k, has2 ? uvo[i0] : kOOB, so, 0));
            return;
        }
        if (sl >= 2 && sl < 6) {
            const int k = sl - 2;
            const float* src = patch_n + tsrc_off;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                td[k][jp].x = src[(k * PW + 2 * jp) * kPS];
                td[k][jp].y = src[(k * PW + 2 * jp + 1) * kPS];
            }
            return;
        }
        if (sl == 9) {
            float* dst = Vn + tdst_off;
            f32x2 tr[4][2];
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {   // B^T d (rows), two columns per instruction
                tr[0][jp] = fs_pk_sub(td[0][jp], td[2][jp]);
                tr[1][jp] = fs_pk_add(td[1][jp], td[2][jp]);
                tr[2][jp] = fs_pk_sub(td[2][jp], td[1][jp]);
                tr[3][jp] = fs_pk_sub(td[1][jp], td[3][jp]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // (.) B (columns)
                const f32x2 o01 = fs_wino_cols01(tr[i][0], tr[i][1]), o23 = fs_wino_cols23(tr[i][0], tr[i][1]);
                dst[(i * 4 + 0) * kNTh * kCC] = o01.x;
                dst[(i * 4 + 1) * kNTh * kCC] = o01.y;
                dst[(i * 4 + 2) * kNTh * kCC] = o23.x;
                dst[(i * 4 + 3) * kNTh * kCC] = o23.y;
            }
        }
    };
    auto sweep = [&](const float* Vl, const float* Ul, const float* patch_n, float* Vn, float* Un) {
        const float* pa = Vl + (8 * ph) * (kNTh * kCC) + kq * (kNTh * 4) + lm * 4;
        const float* pb = Ul + (8 * ph) * (kBN * kCC) + kq * (kBN * 4) + (nb * 32 + lm) * 4;
        float4 A[2][2], B[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            A[0][h] = *reinterpret_cast<const float4*>(pa + h * kNTh * kCC);
            B[0][h] = *reinterpret_cast<const float4*>(pb + h * kBN * kCC);
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int c = pp & 1, n = c ^ 1, p0 = 2 * pp, p1 = 2 * pp + 1;
            const float a0[4] = {A[c][0].x, A[c][0].y, A[c][0].z, A[c][0].w}, b0[4] = {B[c][0].x, B[c][0].y, B[c][0].z, B[c][0].w};
            const float a1[4] = {A[c][1].x, A[c][1].y, A[c][1].z, A[c][1].w}, b1[4] = {B[c][1].x, B[c][1].y, B[c][1].z, B[c][1].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[k], b0[k], acc[p0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0 && pp + 1 < 4) {   // operands of the next position pair
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        A[n][h] = *reinterpret_cast<const float4*>(pa + (p0 + 2 + h) * kNTh * kCC);
                        B[n][h] = *reinterpret_cast<const float4*>(pb + (p0 + 2 + h) * kBN * kCC);
                    }
                } else if (k > 0) {
                    slice(pp * 6 + (k - 1) * 2, patch_n, Vn, Un);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[p1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[k], b1[k], acc[p1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (k > 0) slice(pp * 6 + (k - 1) * 2 + 1, patch_n, Vn, Un);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- epilogue of one item.  Accumulator register r of lane (lm, kq) is tile t = (r & 3) + 8 (r >> 2) + 4 kq of the block
    // (row t / BW, column t % BW), channel nb*32 + lm.  This wave finishes registers r = 8 ph .. 8 ph + 7 after the exchange.
    int toy[8], tox[8];   // top-left output pixel of the tile of register 8 ph + i, relative to the block (or -1: no such tile)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = 8 * ph + i;
        const int t = (r & 3) + 8 * (r >> 2) + 4 * kq;
        const int ty = fdiv(t, inv_bw), tx = t - ty * BW;
        toy[i] = t < ntile ? 2 * ty : -1;
        tox[i] = 2 * tx;
    }
    auto epilogue = [&](const Item& I) {
        const int co = I.co0 + nb * 32 + lm;
        const float bs = a.bias ? a.bias[co] : 0.f;
        const bool relu_out = a.out_relu != 0;
        float* yn = a.y + (size_t)I.n * a.Ho * a.Wo * a.Cout;
        const int Ha = a.Ho - 2 * a.add_pad, Wa = a.Wo - 2 * a.add_pad;
        const float* adn = a.add_src ? a.add_src + (size_t)I.n * Ha * Wa * a.Cout : nullptr;
        const unsigned img_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * a.Cout) * 4u);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yn)), 0, img_bytes, 0x00020000);
        float ad[8][4];
        if (adn) {
            const unsigned add_bytes = __builtin_amdgcn_readfirstlane((unsigned)(Ha * Wa * a.Cout) * 4u);
            const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(adn)), 0, add_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ay = I.oy0 + toy[i] + (k >> 1) - a.add_pad, ax = I.ox0 + tox[i] + (k & 1) - a.add_pad;
                    const bool ok = toy[i] >= 0 && ay >= 0 && ay < Ha && ax >= 0 && ax < Wa;
                    ad[i][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ar, ok ? (unsigned)((ay * Wa + ax) * a.Cout + co) * 4u : kOOB, 0, 0));
                }
        }
        // column stage on this wave's two position rows, its share of the two output rows; the 8 registers the PARTNER
        // finishes go to LDS, the 8 this wave finishes stay
        float mine[8][4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float m[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = FS_ACC_READ(acc[j][r]);
            const float c00 = m[0] + m[1] + m[2], c01 = m[1] - m[2] - m[3];   // position row 2 ph
            const float c10 = m[4] + m[5] + m[6], c11 = m[5] - m[6] - m[7];   // position row 2 ph + 1
            float s[4];   // {y0 col0, y0 col1, y1 col0, y1 col1} share
            if (ph == 0) {   // R0 + R1 | R1
                s[0] = c00 + c10;
                s[1] = c01 + c11;
                s[2] = c10;
                s[3] = c11;
            } else {         // R2 | -R2 - R3
                s[0] = c00;
                s[1] = c01;
                s[2] = -c00 - c10;
                s[3] = -c01 - c11;
            }
            if ((r >> 3) == ph) {
#pragma unroll
                for (int k = 0; k < 4; ++k) mine[r & 7][k] = s[k];
            } else {
                float* xo = xch + ((((nb * 2 + (r >> 3)) * 8 + (r & 7)) * 4) * 64) + lane;
#pragma unroll
                for (int k = 0; k < 4; ++k) xo[k * 64] = s[k];
            }
        }
        FS_LDS_BARRIER();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* xi = xch + ((((nb * 2 + ph) * 8 + i) * 4) * 64) + lane;
#pragma unroll
            for (int k = 0; k < 4; ++k) mine[i][k] += xi[k * 64];
        }
        float s1 = 0.f, s2 = 0.f, cs = 0.f;
        if (a.stats) {   // shift of the one-pass statistics: the block's first pixel of this channel (tile 0: ph 0, kq 0, register 0)
            if (ph == 0 && kq == 0) red[512 + nb * 32 + lm] = mine[0][0];
            FS_LDS_BARRIER();
            cs = red[512 + nb * 32 + lm];
        }
        const int rowp = a.Wo * a.Cout;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int oy = I.oy0 + toy[i] + (k >> 1), ox = I.ox0 + tox[i] + (k & 1);
                const bool ok = toy[i] >= 0 && oy < a.Ho && ox < a.Wo;
                float val = mine[i][k];
                if (a.stats) {
                    const float dv = ok ? val - cs : 0.f;
                    s1 += dv;
                    s2 = fmaf(dv, dv, s2);
                }
                val += bs;
                val = relu_out ? fmaxf(val, 0.f) : val;
                if (adn) val += ad[i][k];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), yr, ok ? (unsigned)(oy * rowp + ox * a.Cout + co) * 4u : kOOB, 0, 0);
            }
        }
        if (a.stats) {
            red[(((ph * 2 + kq) * 64) + nb * 32 + lm) * 2] = s1;
            red[(((ph * 2 + kq) * 64) + nb * 32 + lm) * 2 + 1] = s2;
            FS_LDS_BARRIER();
            if (tid < 64) {
                float S1 = 0.f, S2 = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    S1 += red[(g * 64 + tid) * 2];
                    S2 += red[(g * 64 + tid) * 2 + 1];
                }
                const int th_valid = min(p.TH, a.Ho - I.oy0), tw_valid = min(p.TW, a.Wo - I.ox0);
                const float cnt = (float)(th_valid * tw_valid);
                float* st = a.stats + ((size_t)I.tile_lin * a.Cout + I.co0 + tid) * 3;
                const float shift = red[512 + tid];
                if (a.fin.counter) {   // read by the launch's last workgroup (fused finalize): coherent stores
                    FS_COHERENT_STORE(st, shift + S1 / cnt);
                    FS_COHERENT_STORE(st + 1, fmaxf(S2 - S1 * S1 / cnt, 0.f));
                    FS_COHERENT_STORE(st + 2, cnt);
                } else {
                    st[0] = shift + S1 / cnt;
                    st[1] = fmaxf(S2 - S1 * S1 / cnt, 0.f);
                    st[2] = cnt;
                }
            }
        }
        FS_LDS_BARRIER();   // `xch` / `red` are reused by the next item
        zero_acc();
    };

    // ---- the flat pipeline over (item, chunk) steps.  Before step q is multiplied: its V and U sit in stage q&1, the patch of
    // step q+1 in the other stage, the filter of step q+1 in registers, the patch of step q+2 is requested in slot 0.
    if (my_items == 0) return;
    item_offsets(L);
    float* const st0 = smem;
    float* const st1 = smem + kStageF;
    issue_patch(L, l_chunk, 1);
    issue_filter_pair(L, l_chunk, 0);
    issue_filter_pair(L, l_chunk, 2);
    issue_filter_pair(L, l_chunk, 4);
    issue_filter_pair(L, l_chunk, 6);
    commit_patch(st0);
    commit_filter(st0 + kPatchF + kVF);
    const bool have1 = advance_load();
    issue_patch(L, l_chunk, have1 ? 1 : 0);
    if (have1) {
        issue_filter_pair(L, l_chunk, 0);   // stays in registers until the sweep of step 0 commits it
        issue_filter_pair(L, l_chunk, 2);
        issue_filter_pair(L, l_chunk, 4);
        issue_filter_pair(L, l_chunk, 6);
    }
    __syncthreads();
    transform_now(st0, st0 + kPatchF);
    if (have1) commit_patch(st1);
    __syncthreads();
    load_live = have1 ? 1 : 0;
    FS_WAIT_VMEM();   // (see fs_kernels.h: keeps a vmcnt(0) out of every iteration of the step loop)
#ifdef FS_WINO2H_TRACE
    tr_pro = FS_W2H_NOW() - tr_t0;
#endif
    int q = 0;
    for (int it = 0; it < my_items; ++it) {
        const Item cur_it = decode(it);
        for (int chunk = 0; chunk < nchunks; ++chunk, ++q) {
            has1 = ((chunk + 1 < nchunks) || (it + 1 < my_items)) ? 1 : 0;
            const int o0 = (q & 1) ? kStageF : 0, o1 = kStageF - o0;
#ifdef FS_WINO2H_TRACE
            const long long q0 = FS_W2H_NOW();
#endif
            sweep(smem + o0 + kPatchF, smem + o0 + kPatchF + kVF, smem + o1, smem + o1 + kPatchF, smem + o1 + kPatchF + kVF);
#ifdef FS_WINO2H_TRACE
            const long long q1 = FS_W2H_NOW();
#endif
            if (has2) commit_patch(smem + o0);   // this stage's patch was consumed by the transform of the previous step
#ifdef FS_WINO2H_TRACE
            const long long q2 = FS_W2H_NOW();
#endif
            FS_LDS_BARRIER();                    // (LDS only: the filter loads of step q+2 stay in flight across it)
#ifdef FS_WINO2H_TRACE
            const long long q3 = FS_W2H_NOW();
            tr_sweep += q1 - q0;
            tr_commit += q2 - q1;
            tr_bar += q3 - q2;
#endif
        }
#ifdef FS_WINO2H_TRACE
        const long long e0 = FS_W2H_NOW();
#endif
        epilogue(cur_it);
#ifdef FS_WINO2H_TRACE
        tr_epi += FS_W2H_NOW() - e0;
#endif
    }
    fs_fused_in_finalize(a.fin, a.stats, a.N, smem);   // (every workgroup has at least one item: grid <= items)
#ifdef FS_WINO2H_TRACE
    if (tid == 0 && blockIdx.x < 4096) {
        long long* t = g_wino2h_trace + (size_t)blockIdx.x * 8;
        t[0] = tr_t0;
        t[1] = tr_pro;
        t[2] = tr_sweep;
        t[3] = tr_commit;
        t[4] = tr_bar;
        t[5] = tr_epi;
        t[6] = FS_W2H_NOW();
        t[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
}

// the residual convs of the transform net and their input gradients on grids too small for 64-tile items
bool wino2h_eligible(const ConvArgs& a) {
    const bool pad_ok = a.pad_t == a.pad_l && a.pad_t >= 0 && a.pad_t <= 2 && a.Ho == a.H + 2 * a.pad_t - 2 && a.Wo == a.W + 2 * a.pad_l - 2;
    return a.half_items && a.w_wino2 && a.KH == 3 && a.KW == 3 && a.stride == 1 && pad_ok && a.src_mode == SRC_PLAIN && a.Cin % kCC == 0 &&
           a.Cout % kBN == 0 && !a.shuffle && (!a.in_a || a.pad_t == 0) && a.w_nstride == 0 && a.dil_x <= 1 && (!a.add_src || !a.stats) &&
           !a.mask_src && !a.pool_out && !a.route_src && a.Ho >= 2 && a.Wo >= 2;
}

// block shape (in tiles) with the fewest blocks: 4x8, 8x4, 5x6, 6x5 -- 30 or 32 tiles, patches of <= 180 pixels
static void wino2h_shape(int Ho, int Wo, int* bh, int* bw) {
    static const int shapes[4][2] = {{4, 8}, {5, 6}, {6, 5}, {8, 4}};
    long best = -1;
    const int forced = tune_int("FS_WINO2H_SHAPE", -1);
    for (int s = 0; s < 4; ++s) {
        if (forced >= 0 && forced != s) continue;
        const long n = (long)cdiv(Ho, 2 * shapes[s][0]) * cdiv(Wo, 2 * shapes[s][1]);
        if (best < 0 || n < best) {
            best = n;
            *bh = shapes[s][0];
            *bw = shapes[s][1];
        }
    }
}

long wino2h_items(const ConvArgs& a) {
    int bh = 4, bw = 8;
    wino2h_shape(a.Ho, a.Wo, &bh, &bw);
    return (long)a.N * cdiv(a.Ho, 2 * bh) * cdiv(a.Wo, 2 * bw) * (a.Cout / kBN);
}

void wino2h_plan(const ConvArgs& a, ConvPlan* out) {
    ConvPlan p{};
    int bh = 4, bw = 8;
    wino2h_shape(a.Ho, a.Wo, &bh, &bw);
    p.variant = 8;
    p.BN = kBN;
    p.CC = kCC;
    p.TH = 2 * bh;
    p.TW = 2 * bw;
    p.PH = 2 * bh + 2;
    p.PW = 2 * bw + 2;
    p.tiles_y = cdiv(a.Ho, p.TH);
    p.tiles_x = cdiv(a.Wo, p.TW);
    p.lds_bytes = 4 * (2 * kStageF + kXF + kRedF);
    p.ksplit = 1;
    *out = p;
}

int wino2h_launch(const ConvArgs& a, hipStream_t s) {
    const ConvPlan& p = a.p;
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(wino2h_conv_kernel));
    const long items = (long)a.N * p.tiles_y * p.tiles_x * (a.Cout / kBN);
    const int wgs = tune_int("FS_WINO2_WGS", 256);
    const long grid = items < wgs ? items : wgs;
    hipLaunchKernelGGL(wino2h_conv_kernel, dim3((unsigned)grid), dim3(256), (size_t)p.lds_bytes, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
