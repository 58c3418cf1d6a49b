// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores, second generation (same arithmetic as fs_wino.hip:
// Y = A^T [ (G g G^T) . (B^T d B) ] A, Lavin & Gray, cross-correlation form as tf.nn.conv2d; reference
// libs/vgg16.py:36-220 forward and input gradients, im_transf_net.py:250-276 residual convs).
//
// What changed against the first kernel, and why (measured there: a fixed cost of ~27k cycles per 16x16-pixel block --
// prologue + the LDS exchange of the 16 position planes in the epilogue -- against ~5.5k cycles per 8-channel chunk, i.e.
// 38 % of the lifetime of a block on the 64-channel layers):
//   * ONE wave per SIMD (256 threads, up to 512 registers): a wave owns a 32-tile x 32-channel block of the result for
//     ALL 16 Winograd positions (256 accumulator registers), so the output transform A^T M A runs in registers -- no
//     exchange through LDS, no barriers in the epilogue, 128-byte coalesced stores straight from the accumulators;
//   * PERSISTENT workgroups walk a strided list of (block, channel-block) items and the staging pipeline (global loads
//     two chunks ahead, patch -> V transform one chunk ahead) runs across item boundaries: the prologue of the next
//     block is hidden behind the last chunks of the current one;
//   * operands are laid out K-CONTIGUOUS in LDS (V[pos][kq][tile][4], U[pos][kq][co][4]; lane half kq multiplies
//     channels 4kq..4kq+3 of the chunk): one 16-byte LDS read feeds four matrix instructions (8 reads per chunk and
//     position pair instead of 32) and the 32 lanes of a half read 512 contiguous bytes -- conflict-free (a 32-byte lane
//     stride is a 2-way conflict that costs ~30 cycles per read beside the matrix instructions); the transformed filters
//     are stored in that order in HBM (fs::wt_wino2: [pos][Cin/8][2][Cout][4]) so their staging is a straight 16-byte copy.
#include "fs_kernels.h"

#include <cstdlib>
#include <type_traits>

namespace fs {

namespace {
constexpr int kTT = 8;               // tiles per side of a block (16x16 output pixels)
constexpr int kPT = 2 * kTT + 2;     // patch side (18)
constexpr int kCC = 8;               // input channels per chunk
constexpr int kPS = kCC + 1;         // patch pixel pitch (odd: the transform's reads of a tile row spread over the banks)
constexpr int kBN = 64;              // output channels per item
constexpr int kNT = kTT * kTT;       // tiles per block
constexpr int kPatchF = ((kPT * kPT * kPS + 8 + 3) & ~3);   // 2924 -> patch + 8 floats of sink
constexpr int kVF = 16 * kNT * kCC;  // 8192
constexpr int kUF = 16 * kBN * kCC;  // 8192
constexpr int kStageF = kPatchF + kVF + kUF;
constexpr int kRedF = 64 * 2 * 4 + 64;   // statistics scratch: [4 contributors][64][2] + shift[64]
constexpr unsigned kOOB = 0x80000000u;
}  // namespace

// U2[pos][ci/8][(ci/4)%2][co][ci%4] = (G g G^T)[pos] for g = w[:, :, ci, co]   (w HWIO [3][3][Cin][Cout]); blockIdx.y = filter of the batch
__global__ __launch_bounds__(256) void wt_wino2_batch_kernel(WinoBatch b, int Cin, int Cout) {
    const float* __restrict__ w = b.w[blockIdx.y];
    float* __restrict__ U = b.U[blockIdx.y];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t cc = (size_t)Cin * Cout;
    if (i >= cc) return;
    const int ci = (int)(i / Cout), co = (int)(i - (size_t)ci * Cout);
    float g[3][3], t[4][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) g[kh][kw] = w[(size_t)(kh * 3 + kw) * cc + i];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        t[0][kw] = g[0][kw];
        t[1][kw] = 0.5f * (g[0][kw] + g[1][kw] + g[2][kw]);
        t[2][kw] = 0.5f * (g[0][kw] - g[1][kw] + g[2][kw]);
        t[3][kw] = g[2][kw];
    }
    const size_t pos_stride = cc;   // floats per position plane
    float* dst = U + (((size_t)(ci >> 3) * 2 + ((ci >> 2) & 1)) * Cout + co) * 4 + (ci & 3);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dst[(size_t)(r * 4 + 0) * pos_stride] = t[r][0];
        dst[(size_t)(r * 4 + 1) * pos_stride] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
        dst[(size_t)(r * 4 + 2) * pos_stride] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
        dst[(size_t)(r * 4 + 3) * pos_stride] = t[r][2];
    }
}

int wt_wino2_batch(const WinoBatch& b, int Cin, int Cout, hipStream_t s) {
    if (b.n <= 0) return 0;
    const size_t cc = (size_t)Cin * Cout;
    hipLaunchKernelGGL(wt_wino2_batch_kernel, dim3((unsigned)((cc + 255) / 256), (unsigned)b.n), dim3(256), 0, s, b, Cin, Cout);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int wt_wino2(const float* w, float* U, int Cin, int Cout, hipStream_t s) {
    WinoBatch b{};
    b.w[0] = w;
    b.U[0] = U;
    b.n = 1;
    return wt_wino2_batch(b, Cin, Cout, s);
}

#ifdef FS_WINO2_TRACE
// debug build only (tools/conv_trace.py --wino2): per-workgroup phase cycle counts of the last launch
__device__ long long g_wino2_trace[4096 * 8];
extern "C" int fs_debug_conv_trace(long long* out, int n_wg) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino2_trace), sizeof(long long) * 8 * (size_t)n_wg, 0, hipMemcpyDeviceToHost);
}
extern "C" int fs_debug_conv_trace_reset() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wino2_trace)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(long long) * 8 * 4096);
}
#define FS_W2_NOW() ((long long)__builtin_readcyclecounter())
#endif

// REM = true (transform-net launches with a.p.rem_ks > 0): REMAINDER SPLIT.  A persistent launch of `items` items on G
// workgroups takes ceil(items / G) rounds, and the last one is mostly empty (batch 32, 74-pixel maps: 800 items = 3.1 rounds
// -> 4; a 720p frame: 273 items = 1.07 rounds -> 2).  Here the items of that last partial round are split over their
// input-channel chunks into rem_ks units each (rem * rem_ks <= G): every workgroup multiplies ONE unit after its whole items
// and leaves the output-transformed partial tile in scratch; wino2_rem_epilogue_kernel sums the rem_ks partials of an item
// and does what the item's epilogue would have done (statistics record / residual add, store).  The VGG16 launches
// (thousands of items) instantiate REM = false: the code of round 2, unchanged.
template <bool REM>
__global__ __launch_bounds__(256) void wino2_conv_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
#ifdef FS_WINO2_TRACE
    const long long tr_t0 = FS_W2_NOW();
    long long tr_sweep = 0, tr_post = 0, tr_bar = 0, tr_epi = 0, tr_pro = 0;
#endif
    const ConvPlan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, kq = lane >> 5;
    const int mb = wave & 1, nb = wave >> 1;   // this wave's 32-tile block / 32-channel block of the 64 x 64 item
    auto fdiv = [](int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); };
    auto uniform_ptr = [](const float* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };
    float* red = smem + 2 * kStageF;   // statistics scratch (transform-net form only)


    // ---- the item list of this workgroup: item = ((n * blocks + block) * ncob + channel block) * ksplit + z
    const int blocks = p.tiles_y * p.tiles_x;
    const int ncob = a.Cout / kBN;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int nchunks_all = a.Cin / kCC;
    const int whole_items = a.N * blocks * ncob * ks;
    const int rem_full = REM ? p.rem_full : whole_items, rem_ks = REM ? p.rem_ks : 1;
    const int total_items = REM ? rem_full + (whole_items - rem_full) * rem_ks : whole_items;   // whole items + units
    const float inv_rks = 1.0f / (float)rem_ks;
    const int G = (int)gridDim.x;
    const int my_items = ((int)blockIdx.x < total_items) ? (total_items - 1 - (int)blockIdx.x) / G + 1 : 0;
    const float inv_ks = 1.0f / (float)ks, inv_ncob = 1.0f / (float)ncob, inv_blocks = 1.0f / (float)blocks, inv_tx = 1.0f / (float)p.tiles_x;
    struct Item {
        int n, oy0, ox0, co0, cbeg, cend, z, tile_lin, slot;   // slot >= 0: a UNIT of the remainder split (its partial tile's place in scratch)
    };
    auto decode = [&](int it) {
        Item r;
        int lin = (int)blockIdx.x + it * G;
        int rz = 0;
        r.slot = -1;
        if (REM && lin >= rem_full) {   // unit u of the remainder split: item rem_full + u / rem_ks, chunk range u % rem_ks
            const int u = lin - rem_full;
            const int ri = fdiv(u, inv_rks);
            rz = u - ri * rem_ks;
            r.slot = u;
            lin = rem_full + ri;
        }
        const int t1 = fdiv(lin, inv_ks);
        r.z = lin - t1 * ks;
        const int t2 = fdiv(t1, inv_ncob);
        const int cob = t1 - t2 * ncob;
        r.tile_lin = t2;
        r.n = fdiv(t2, inv_blocks);
        const int br = t2 - r.n * blocks;
        const int byi = fdiv(br, inv_tx);
        r.oy0 = byi * 2 * kTT;
        r.ox0 = (br - byi * p.tiles_x) * 2 * kTT;
        r.co0 = cob * kBN;
        r.cbeg = ks > 1 ? r.z * nchunks_all / ks : 0;
        r.cend = ks > 1 ? (r.z + 1) * nchunks_all / ks : nchunks_all;
        if (REM && r.slot >= 0) {
            r.cbeg = rz * nchunks_all / rem_ks;
            r.cend = (rz + 1) * nchunks_all / rem_ks;
        }
        r.slot = __builtin_amdgcn_readfirstlane(r.slot);
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.oy0 = __builtin_amdgcn_readfirstlane(r.oy0);
        r.ox0 = __builtin_amdgcn_readfirstlane(r.ox0);
        r.co0 = __builtin_amdgcn_readfirstlane(r.co0);
        r.cbeg = __builtin_amdgcn_readfirstlane(r.cbeg);
        r.cend = __builtin_amdgcn_readfirstlane(r.cend);
        r.z = __builtin_amdgcn_readfirstlane(r.z);
        r.tile_lin = __builtin_amdgcn_readfirstlane(r.tile_lin);
        return r;
    };

    // ---- staging descriptors
    // patch: 18*18 pixels x 2 float4 = 648 elements, <= 3 per thread; element -> (py << 8 | px), channel quad c4
    int pq[3], pdst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = tid + i * 256;
        pq[i] = -1;
        pdst[i] = kPT * kPT * kPS;   // sink
        if (e < kPT * kPT * 2) {
            const int pix = e >> 1, c4 = e & 1;
            const int py = pix / kPT, px = pix - py * kPT;
            pq[i] = (py << 8) | px;
            pdst[i] = pix * kPS + c4 * 4;
        }
    }
    const int pc4 = (tid & 1) * 4;     // every element of a thread has the same channel quad (256 is even)
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * a.Cin) * 4u);
    const unsigned u_bytes = __builtin_amdgcn_readfirstlane((unsigned)(16 * a.Cin * a.Cout) * 4u);
    const bool has_ab = a.in_a != nullptr;
    const float* ub = uniform_ptr(a.w_wino2);
    float4 pv[3], uv[8];
    const __amdgpu_buffer_rsrc_t ur_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, u_bytes, 0x00020000);   // kernel-constant
    float4 va = make_float4(1.f, 1.f, 1.f, 1.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned gvo[3];
    unsigned uvo[8];   // filter element offsets (kernel-constant per thread)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + i * 256;
        uvo[i] = (unsigned)((e >> 7) * a.Cin * a.Cout + ((e >> 6) & 1) * a.Cout * 4 + (e & 63) * 4) * 4u;
    }

    auto item_offsets = [&](const Item& I) {   // global offsets of the thread's patch elements for this item
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int py = pq[i] >> 8, px = pq[i] & 255;
            const int sy = I.oy0 - a.pad_t + py, sx = I.ox0 - a.pad_l + px;
            const bool ok = pq[i] >= 0 && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
            gvo[i] = ok ? (unsigned)((sy * a.W + sx) * a.Cin + pc4) * 4u : kOOB;
        }
    };
    auto issue_patch = [&](const Item& I, int chunk) {
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * a.Cin);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, x_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < 3; ++i) pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, gvo[i], chunk * kCC * 4, 0));
        if (has_ab) {
            va = *reinterpret_cast<const float4*>(a.in_a + (size_t)I.n * a.in_nstride + chunk * kCC + pc4);
            vb = *reinterpret_cast<const float4*>(a.in_b + (size_t)I.n * a.in_nstride + chunk * kCC + pc4);
        }
    };
    auto issue_filter = [&](const Item& I, int chunk) {
        // U2[pos][chunk][kq][co][4]: per (position, lane half) the 64 channels of the item are 1 KB contiguous; element e of
        // the thread = (pos = e >> 7, kq = (e >> 6) & 1, channel = e & 63), LDS position = e
        const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ub), 0, u_bytes, 0x00020000);
        const unsigned so = (unsigned)((chunk * 2 * a.Cout + I.co0) * 4) * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ur, uvo[i], so, 0));
        }
    };
    auto issue_filter_pair = [&](const Item& I, int chunk, int i0) {   // elements i0, i0+1 of issue_filter (one sweep slot)
        const unsigned so = (unsigned)((chunk * 2 * a.Cout + I.co0) * 4) * 4u;
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
            uv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ur_k, uvo[i], so, 0));
        }
    };
    auto commit_patch = [&](float* patch) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float4 v = pv[i];
            if (has_ab) {   // (VALID padding only: every patch pixel that reaches a stored output is a real pixel)
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
    // input transform V = B^T d B: a thread does two (tile, channel) pairs per chunk
    auto transform_pair = [&](const float* patch, float* Vl, int pidx) {
        const int tt = pidx >> 3, tk = pidx & 7;
        const int tty = tt >> 3, ttx = tt & 7;
        const float* src = patch + ((2 * tty) * kPT + 2 * ttx) * kPS + tk;
        float d[4][4], r[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = src[(i * kPT + j) * kPS];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[0][j] = d[0][j] - d[2][j];
            r[1][j] = d[1][j] + d[2][j];
            r[2][j] = d[2][j] - d[1][j];
            r[3][j] = d[1][j] - d[3][j];
        }
        float* dst = Vl + (tk >> 2) * (kNT * 4) + tt * 4 + (tk & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[(i * 4 + 0) * kNT * kCC] = r[i][0] - r[i][2];
            dst[(i * 4 + 1) * kNT * kCC] = r[i][1] + r[i][2];
            dst[(i * 4 + 2) * kNT * kCC] = r[i][2] - r[i][1];
            dst[(i * 4 + 3) * kNT * kCC] = r[i][1] - r[i][3];
        }
    };

    f32x16 acc[16];   // one 32x32 block per Winograd position
    auto zero_acc = [&]() {
#pragma unroll
        for (int pos = 0; pos < 16; ++pos)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pos][r] = 0.f;
    };
    zero_acc();

    // sweep of one chunk: per position one 16-byte read of A (tiles) and of B (channels), four matrix instructions
    // (channels 4kq..4kq+3 of the chunk in lane half kq); the reads of position pos+1 are issued behind the first matrix
    // instruction of position pos.  The preparation of the NEXT step (input transform of the thread's two (tile, channel)
    // pairs, filter commit) is threaded through the sweep in small slices -- three slots per position, pinned with
    // sched_barrier: with one wave per SIMD nothing else covers that work, and left to itself the compiler emits each
    // transform as one block of ~60 instructions and LDS round trips during which the matrix pipe idles.
    // Measured on gfx950 (tools/mfma_overlap.hip, one wave per SIMD): the fp32 matrix instruction and the vector ALU exclude
    // each other -- every VALU instruction between two v_mfma_f32_32x32x2_f32 adds ~6.5 cycles, plus ~10 for the first one
    // of a gap; scalar instructions are free (up to ~8 per MFMA), LDS reads nearly so.  Hence: LDS reads and global loads
    // are spread over the gaps, the VALU work of the input transform is bunched into TWO gaps.
    // the 4x4 input blocks of the thread's two (tile, channel) pairs, as column PAIRS: the transform arithmetic runs as packed
    // fp32 (v_pk_add_f32, fs_wino_cols: 16 instead of 32 vector instructions per block)
    f32x2 td[2][4][2];
    // loop state the slices touch (the load stream runs inside the sweep: nothing but the barrier stands between two sweeps)
    // (ints, not bools: a bool that lives across basic blocks is kept as a 64-bit lane mask -- two scalar registers each, spilled
    // to vector lanes and read back with v_readlane inside the sweep)
    int has1 = 0, has2 = 0, load_live = 0;
    Item L = decode(0);          // cursor of the load stream: the step whose loads were issued last
    int l_it = 0, l_chunk = L.cbeg;
    auto advance_load = [&]() {   // returns false when the stream is exhausted
        if (++l_chunk < L.cend) return true;
        if (++l_it >= my_items) return false;
        L = decode(l_it);
        l_chunk = L.cbeg;
        item_offsets(L);
        return true;
    };
#ifndef FS_W2_DMA
#define FS_W2_DMA 0   /* 1: the filter chunk goes global -> LDS by DMA (global_load_lds_dwordx4) -- no staging registers, no ds_write pass, but MEASURED SLOWER (sweep 5.9k -> 6.6k cycles per chunk): kept as a recorded experiment */
#endif
#if FS_W2_DMA && !defined(FS_LDS_BARRIER_OFF)
#error "FS_W2_DMA: the filter stage arrives by global_load_lds (completes under vmcnt) -- build with -DFS_LDS_BARRIER_OFF so that the chunk barrier waits for it; FS_LDS_BARRIER orders LDS traffic only"
#endif
#ifndef FS_W2_ABL
#define FS_W2_ABL 0   /* timing experiments (tools/conv_trace.py): 1 no input transform, 2 no filter commit, 4 no global loads, 8 no operand reads */
#endif
    auto slice = [&](int sl, const float* patch_n, float* Vn, float* Un) {
        // 48 slots per sweep.  0-3: global loads of the filter of step q+1; 4: the load cursor moves to step q+2 and its
        // patch loads go out; 5-12: LDS reads of the two 4x4 input blocks (one row per slot); 28 / 34: the transform
        // arithmetic + stores of pair `tid` / `tid + 256`, each in ONE gap; 40-47: the filter commit, as late as its loads allow
        if (sl < 4) {
            if (has1 && !(FS_W2_ABL & 4)) {
                if (FS_W2_DMA) {
                    // U2 -> LDS directly: the LDS image of a filter chunk is lane-linear (element e of the chunk at float4 slot e), so a
                    // wave's 64 lanes write 1 KB contiguous; the barrier at the end of the sweep waits for the transfers
                    const float* gsrc = ub + (size_t)(l_chunk * 2 * a.Cout + L.co0) * 4;
#pragma unroll
                    for (int i = 2 * sl; i < 2 * sl + 2; ++i)
                        FS_GLOBAL_LOAD_LDS_B128(gsrc + (uvo[i] >> 2), Un + (wave * 64 + i * 256) * 4);
                } else {
                    issue_filter_pair(L, l_chunk, 2 * sl);
                }
            }
            return;
        }
        if (sl == 4) {
            has2 = 0;
            if (has1 && load_live) {
                has2 = advance_load() ? 1 : 0;
                load_live = has2;
                if (has2 && !(FS_W2_ABL & 4)) issue_patch(L, l_chunk);
            }
            return;
        }
        if (sl >= 40) {
            if ((FS_W2_ABL & 2) || FS_W2_DMA) return;
            const int i0 = sl - 40;
            *reinterpret_cast<float4*>(Un + (tid + i0 * 256) * 4) = uv[i0];
            return;
        }
        if (FS_W2_ABL & 1) return;
        if (sl >= 5 && sl < 13) {
            const int half = (sl - 5) >> 2, k = (sl - 5) & 3;
            const int pidx = tid + half * 256;
            const int tt = pidx >> 3, tk = pidx & 7;
            const float* src = patch_n + ((2 * (tt >> 3)) * kPT + 2 * (tt & 7)) * kPS + tk;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                td[half][k][jp].x = src[(k * kPT + 2 * jp) * kPS];
                td[half][k][jp].y = src[(k * kPT + 2 * jp + 1) * kPS];
            }
            return;
        }
        if (sl == 28 || sl == 34) {
            const int half = sl == 34 ? 1 : 0;
            const int pidx = tid + half * 256;
            const int tt = pidx >> 3, tk = pidx & 7;
            float* dst = Vn + (tk >> 2) * (kNT * 4) + tt * 4 + (tk & 3);
            f32x2 tr[4][2];
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {   // B^T d (rows), two columns per instruction
                tr[0][jp] = fs_pk_sub(td[half][0][jp], td[half][2][jp]);
                tr[1][jp] = fs_pk_add(td[half][1][jp], td[half][2][jp]);
                tr[2][jp] = fs_pk_sub(td[half][2][jp], td[half][1][jp]);
                tr[3][jp] = fs_pk_sub(td[half][1][jp], td[half][3][jp]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // (.) B (columns)
                const f32x2 o01 = fs_wino_cols01(tr[i][0], tr[i][1]), o23 = fs_wino_cols23(tr[i][0], tr[i][1]);
                dst[(i * 4 + 0) * kNT * kCC] = o01.x;
                dst[(i * 4 + 1) * kNT * kCC] = o01.y;
                dst[(i * 4 + 2) * kNT * kCC] = o23.x;
                dst[(i * 4 + 3) * kNT * kCC] = o23.y;
            }
        }
    };
    auto sweep = [&](const float* Vl, const float* Ul, const float* patch_n, float* Vn, float* Un) {
        const float* pa = Vl + kq * (kNT * 4) + (mb * 32 + lm) * 4;
        const float* pb = Ul + kq * (kBN * 4) + (nb * 32 + lm) * 4;
        // two positions at a time, their matrix instructions alternating: consecutive MFMAs never share an accumulator (an
        // instruction slipped between two MFMAs on the SAME accumulator costs ~40 cycles, between independent ones ~6)
        float4 A[2][2], B[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            A[0][h] = *reinterpret_cast<const float4*>(pa + h * kNT * kCC);
            B[0][h] = *reinterpret_cast<const float4*>(pb + h * kBN * kCC);
        }
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int c = pp & 1, n = c ^ 1, p0 = 2 * pp, p1 = 2 * pp + 1;
            const float a0[4] = {A[c][0].x, A[c][0].y, A[c][0].z, A[c][0].w}, b0[4] = {B[c][0].x, B[c][0].y, B[c][0].z, B[c][0].w};
            const float a1[4] = {A[c][1].x, A[c][1].y, A[c][1].z, A[c][1].w}, b1[4] = {B[c][1].x, B[c][1].y, B[c][1].z, B[c][1].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[k], b0[k], acc[p0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0 && pp + 1 < 8 && !(FS_W2_ABL & 8)) {   // operands of the next position pair
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        A[n][h] = *reinterpret_cast<const float4*>(pa + (p0 + 2 + h) * kNT * kCC);
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

    // ---- epilogue of one item: output transform in registers, bias / ReLU / residual add / consumer mask / statistics, store.
    // Accumulator register r of lane (lm, kq) is tile (4 mb + (r >> 2), (r & 3) + 4 kq) of the block, channel nb*32 + lm.
    // FULL: the 16x16-pixel block lies inside the image (no per-pixel predicates).  Every consumer-mask / residual load of
    // the item is issued up front (clamped addresses for pixels outside the image) -- one memory latency per item, not
    // one per pixel.
    auto epilogue_body = [&](auto FULLT, const Item& I) {
        constexpr bool full = decltype(FULLT)::value;
        const int co = I.co0 + nb * 32 + lm;
        const float bs = a.bias ? a.bias[co] : 0.f;
        const bool relu_out = a.out_relu != 0;
        float* yn = a.y + ((size_t)I.n + (ks > 1 ? (size_t)I.z * a.N : 0)) * a.Ho * a.Wo * a.Cout;
        const float* msn = a.mask_src ? a.mask_src + (size_t)I.n * a.Ho * a.Wo * a.Cout : nullptr;
        const int Ha = a.Ho - 2 * a.add_pad, Wa = a.Wo - 2 * a.add_pad;
        const float* adn = a.add_src ? a.add_src + (size_t)I.n * Ha * Wa * a.Cout : nullptr;
        const int oyb = I.oy0 + 8 * mb, oxb = I.ox0 + 8 * kq;     // first pixel of the lane's tiles: rows r>>2, columns r&3
        const int rowp = a.Wo * a.Cout;
        const int obase = (oyb * a.Wo + oxb) * a.Cout + co;
        // buffer resources: pixels outside the image (edge blocks) get the out-of-range offset -- loads return 0, stores are
        // dropped by the hardware range check -- so there is no branch and no saved exec mask per pixel
        const unsigned img_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * a.Cout) * 4u);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yn)), 0, img_bytes, 0x00020000);
        float mk[16][4], ad[16][4];
        if (msn) {
            const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(msn)), 0, img_bytes, 0x00020000);
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int py = 2 * (r >> 2) + (k >> 1), px = 2 * (r & 3) + (k & 1);
                    const unsigned off = (unsigned)(obase + py * rowp + px * a.Cout) * 4u;
                    const bool ok = full || (oyb + py < a.Ho && oxb + px < a.Wo);
                    mk[r][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mr, ok ? off : kOOB, 0, 0));
                }
        }
        if (adn) {
            const unsigned add_bytes = __builtin_amdgcn_readfirstlane((unsigned)(Ha * Wa * a.Cout) * 4u);
            const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(adn)), 0, add_bytes, 0x00020000);
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ay = oyb + 2 * (r >> 2) + (k >> 1) - a.add_pad, ax = oxb + 2 * (r & 3) + (k & 1) - a.add_pad;
                    const bool ok = ay >= 0 && ay < Ha && ax >= 0 && ax < Wa;
                    ad[r][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ar, ok ? (unsigned)((ay * Wa + ax) * a.Cout + co) * 4u : kOOB, 0, 0));
                }
        }
        float* pon = a.pool_out ? a.pool_out + (size_t)I.n * (a.Ho >> 1) * (a.Wo >> 1) * a.Cout : nullptr;
        float s1 = 0.f, s2 = 0.f, cs = 0.f;
        if (a.stats) {   // shift of the one-pass statistics: the block's first pixel of this channel (held by tile 0: mb 0, kq 0)
            if (mb == 0 && kq == 0) {
                const float m00 = acc[0][0] + acc[1][0] + acc[2][0], m01 = acc[4][0] + acc[5][0] + acc[6][0],
                            m02 = acc[8][0] + acc[9][0] + acc[10][0];
                red[512 + nb * 32 + lm] = m00 + m01 + m02;
            }
            FS_LDS_BARRIER();   // (LDS exchange only; __syncthreads() would also wait for every global load / store in flight)
            cs = red[512 + nb * 32 + lm];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // the 16 position values of this row leave the accumulator file HERE (FS_ACC_READ: a volatile read on the GPU): left
            // to itself the compiler copies ~180 accumulators to ordinary registers in one block behind the chunk loop and spills
            float m[16];
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) m[pos] = FS_ACC_READ(acc[pos][r]);
            float s4[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // A^T M
                s4[0][j] = m[0 + j] + m[4 + j] + m[8 + j];
                s4[1][j] = m[4 + j] - m[8 + j] - m[12 + j];
            }
            float v[4];
#pragma unroll
            for (int ai = 0; ai < 2; ++ai) {
                v[ai * 2 + 0] = s4[ai][0] + s4[ai][1] + s4[ai][2];
                v[ai * 2 + 1] = s4[ai][1] - s4[ai][2] - s4[ai][3];
            }
            float pmax = -3.0e38f;   // max of the tile's four stored values: the 2x2/2 max-pool window (tiles sit on even coordinates)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int py = 2 * (r >> 2) + (k >> 1), px = 2 * (r & 3) + (k & 1);
                const bool ok = full || (oyb + py < a.Ho && oxb + px < a.Wo);
                float val = v[k];
                if (a.stats) {
                    const float dv = ok ? val - cs : 0.f;
                    s1 += dv;
                    s2 = fmaf(dv, dv, s2);
                }
                val += bs;
                val = relu_out ? fmaxf(val, 0.f) : val;
                if (adn) val += ad[r][k];
                if (msn) val = mk[r][k] > 0.f ? val : 0.f;
                pmax = fmaxf(pmax, val);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), yr, ok ? (unsigned)(obase + py * rowp + px * a.Cout) * 4u : kOOB, 0, 0);
            }
            if (pon) {
                const int qy = (oyb >> 1) + (r >> 2), qx = (oxb >> 1) + (r & 3);
                if (full || (2 * qy < a.Ho && 2 * qx < a.Wo)) pon[(qy * (a.Wo >> 1) + qx) * a.Cout + co] = pmax;
            }
        }
        if (a.stats) {
            // per-block instance-norm partials of the RAW conv output (mean, M2, count) around the shift `cs`: the four
            // contributors of a channel (2 lane halves x 2 tile blocks) meet in LDS
            red[(((mb * 2 + kq) * 64) + nb * 32 + lm) * 2] = s1;
            red[(((mb * 2 + kq) * 64) + nb * 32 + lm) * 2 + 1] = s2;
            FS_LDS_BARRIER();   // (the item's 64 stores per lane keep draining behind it)
            if (tid < 64) {
                float S1 = 0.f, S2 = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    S1 += red[(g * 64 + tid) * 2];
                    S2 += red[(g * 64 + tid) * 2 + 1];
                }
                const int th_valid = min(2 * kTT, a.Ho - I.oy0), tw_valid = min(2 * kTT, a.Wo - I.ox0);
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
            FS_LDS_BARRIER();   // `red` is reused by the next item
        }
    };
    // a unit of the remainder split: output transform in registers as usual, the partial 16x16-pixel x 64-channel tile goes
    // to scratch as [slot][pixel][channel] (no bias / statistics / add: wino2_rem_epilogue_kernel does those on the sum)
    auto epilogue_partial = [&](const Item& I) {
        float* dst = a.rem_ws + (size_t)I.slot * (4 * kNT * kBN) + ((8 * mb) * (2 * kTT) + 8 * kq) * kBN + nb * 32 + lm;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float m[16];
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) m[pos] = FS_ACC_READ(acc[pos][r]);
            float s4[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s4[0][j] = m[0 + j] + m[4 + j] + m[8 + j];
                s4[1][j] = m[4 + j] - m[8 + j] - m[12 + j];
            }
#pragma unroll
            for (int ai = 0; ai < 2; ++ai) {
                const int py = 2 * (r >> 2) + ai, px = 2 * (r & 3);
                dst[(py * (2 * kTT) + px) * kBN] = s4[ai][0] + s4[ai][1] + s4[ai][2];
                dst[(py * (2 * kTT) + px + 1) * kBN] = s4[ai][1] - s4[ai][2] - s4[ai][3];
            }
        }
    };
    auto epilogue = [&](const Item& I) {
        if (REM && I.slot >= 0)
            epilogue_partial(I);
        else if (I.oy0 + 2 * kTT <= a.Ho && I.ox0 + 2 * kTT <= a.Wo)
            epilogue_body(std::true_type{}, I);
        else
            epilogue_body(std::false_type{}, I);
        zero_acc();
    };

    // ---- the flat pipeline over (item, chunk) steps.  Step q multiplies out of stage q&1 while step q+1 is prepared
    // (patch -> V, filter commit) into the other stage and the global loads of step q+2 are in flight.
    if (my_items == 0) return;
    item_offsets(L);
    float* const st0 = smem;
    float* const st1 = smem + kStageF;
    // step 0: load + commit; step 1: its patch (the filter of step q+1 always travels during the sweep of step q)
    issue_patch(L, l_chunk);
    issue_filter(L, l_chunk);
    commit_patch(st0);
    commit_filter(st0 + kPatchF + kVF);
    const bool have1 = advance_load();
    if (have1) issue_patch(L, l_chunk);
    __syncthreads();
    transform_pair(st0, st0 + kPatchF, tid);
    transform_pair(st0, st0 + kPatchF, tid + 256);
    if (have1) commit_patch(st1);
    __syncthreads();
#ifdef FS_WINO2_TRACE
    tr_pro = FS_W2_NOW() - tr_t0;
    long long tr_last = FS_W2_NOW();
#endif
    load_live = have1 ? 1 : 0;   // a step beyond the one whose patch is already committed may still exist
    FS_WAIT_VMEM();   // (fs_kernels.h: no prologue load may still be pending at the loop header, or every iteration waits vmcnt(0))
    int q = 0;
    // (nested item / chunk loops rather than one flat loop with a conditional epilogue: a conditional re-zeroing of the 256
    // accumulators makes the register allocator merge two versions of them at the join -- copies and spills)
    for (int it = 0; it < my_items; ++it) {
        const Item cur_it = decode(it);
        for (int chunk = cur_it.cbeg; chunk < cur_it.cend; ++chunk, ++q) {
            // does step q+1 exist?  (its patch already sits in the other stage's patch area; L points at it)
            has1 = ((chunk + 1 < cur_it.cend) || (it + 1 < my_items)) ? 1 : 0;
#ifdef FS_WINO2_TRACE
            const long long q0 = FS_W2_NOW();
#endif
            // stage offsets as integers into the LDS array (selecting between two POINTERS makes them generic pointers:
            // a null check and 64-bit arithmetic per access); ONE copy of the step body -- a second one makes the register
            // allocator merge two versions of the 256 accumulators at the join.  The transform / filter-commit slices run
            // unconditionally: after the last step they work on stale data that nothing reads.
            const int o0 = (q & 1) ? kStageF : 0, o1 = kStageF - o0;
            sweep(smem + o0 + kPatchF, smem + o0 + kPatchF + kVF, smem + o1, smem + o1 + kPatchF, smem + o1 + kPatchF + kVF);
            if (has2 && !(FS_W2_ABL & 4)) commit_patch(smem + o0);   // this stage's patch was consumed by the transform of the previous step
#ifdef FS_WINO2_TRACE
            const long long q1 = FS_W2_NOW();
            const long long q2 = q1;
#endif
            FS_LDS_BARRIER();   // stages are exchanged through LDS only; the global loads of steps q+1 / q+2 stay in flight
#ifdef FS_WINO2_TRACE
            const long long q3 = FS_W2_NOW();
            tr_sweep += q1 - q0;
            tr_post += q2 - q1 + (q0 - tr_last);
            tr_bar += q3 - q2;
            tr_last = q3;
#endif
        }
#ifdef FS_WINO2_TRACE
        const long long e0 = FS_W2_NOW();
#endif
        epilogue(cur_it);
#ifdef FS_WINO2_TRACE
        tr_last = FS_W2_NOW();
        tr_epi += tr_last - e0;
#endif
    }
    fs_fused_in_finalize(a.fin, a.stats, a.N, smem);   // (every workgroup has at least one item: grid <= items)
#ifdef FS_WINO2_TRACE
    if (tid == 0 && blockIdx.x < 4096) {
        long long* t = g_wino2_trace + (size_t)blockIdx.x * 8;
        t[0] = tr_t0;
        t[1] = tr_pro;
        t[2] = tr_sweep;
        t[3] = tr_post;
        t[4] = tr_bar;
        t[5] = tr_epi;
        t[6] = FS_W2_NOW();
        t[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
}

// Second half of the remainder split: one workgroup per split item sums the rem_ks partial tiles of its units and finishes
// the item -- the per-block instance-norm record of the raw conv output (same {mean, M2, count} form and shift rule as the
// kernel's own epilogue) or the residual-gradient addend, and the store.  thread = (channel, one row of the 16x16 block):
// all of a thread's loads are issued before its first store (a serial pixel loop here costs one memory latency per pixel).
__global__ __launch_bounds__(1024) void wino2_rem_epilogue_kernel(ConvArgs a) {
    __shared__ float red[16 * 64 * 2];
    const ConvPlan& p = a.p;
    const int tid = threadIdx.x, c = tid & 63, g = tid >> 6;
    const int ncob = a.Cout / kBN, blocks = p.tiles_y * p.tiles_x;
    const int item = p.rem_full + (int)blockIdx.x;
    const int t2 = item / ncob, cob = item - t2 * ncob;
    const int n = t2 / blocks, br = t2 - n * blocks;
    const int byi = br / p.tiles_x;
    const int oy0 = byi * 2 * kTT, ox0 = (br - byi * p.tiles_x) * 2 * kTT, co = cob * kBN + c;
    const int ks = p.rem_ks;
    const float* __restrict__ part = a.rem_ws + (size_t)blockIdx.x * ks * (4 * kNT * kBN) + c;
    const int oy = oy0 + g;
    float v[16], cs = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
    // (every load of a thread in flight before the first add: with the partials walked one after the other the kernel paid one
    // memory latency per partial -- 23 us for 20 MB at batch 32; rem_ks is a power of two <= 8, summed in the same fixed order)
    auto sum_group = [&](auto KS, int z0) __attribute__((always_inline)) {   // partials z0 .. z0 + K - 1
        constexpr int K = decltype(KS)::value;
        float t[K][16], c0[K];
#pragma unroll
        for (int z = 0; z < K; ++z) {
            const float* pz = part + (size_t)(z0 + z) * (4 * kNT) * kBN;
            c0[z] = pz[0];   // shift of the one-pass statistics: the block's first pixel of the channel
#pragma unroll
            for (int i = 0; i < 16; ++i) t[z][i] = pz[(g * 16 + i) * kBN];
        }
#pragma unroll
        for (int z = 0; z < K; ++z) {
            cs += c0[z];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] += t[z][i];
        }
    };
    if (ks == 2) {
        sum_group(std::integral_constant<int, 2>{}, 0);
    } else if ((ks & 3) == 0) {
        for (int z0 = 0; z0 < ks; z0 += 4) sum_group(std::integral_constant<int, 4>{}, z0);
    } else {
        for (int z = 0; z < ks; ++z) sum_group(std::integral_constant<int, 1>{}, z);
    }
    const int Ha = a.Ho - 2 * a.add_pad, Wa = a.Wo - 2 * a.add_pad;
    if (a.stats) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float d = (oy < a.Ho && ox0 + i < a.Wo) ? v[i] - cs : 0.f;
            s1 += d;
            s2 = fmaf(d, d, s2);
        }
        red[(g * 64 + c) * 2] = s1;
        red[(g * 64 + c) * 2 + 1] = s2;
    }
    if (a.add_src) {
        const float* __restrict__ adn = a.add_src + (size_t)n * Ha * Wa * a.Cout;
        const int ay = oy - a.add_pad;
        float ad[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ax = ox0 + i - a.add_pad;
            ad[i] = (ay >= 0 && ay < Ha && ax >= 0 && ax < Wa) ? adn[((size_t)ay * Wa + ax) * a.Cout + co] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += ad[i];
    }
    if (oy < a.Ho) {
        float* yr = a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox0) * a.Cout + co;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (ox0 + i < a.Wo) yr[(size_t)i * a.Cout] = v[i];
    }
    if (a.stats) {
        __syncthreads();
        if (tid < 64) {
            float S1 = 0.f, S2 = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                S1 += red[(q * 64 + tid) * 2];
                S2 += red[(q * 64 + tid) * 2 + 1];
            }
            const int th_valid = min(2 * kTT, a.Ho - oy0), tw_valid = min(2 * kTT, a.Wo - ox0);
            const float cnt = (float)(th_valid * tw_valid);
            float* st = a.stats + ((size_t)t2 * a.Cout + co) * 3;
            st[0] = cs + S1 / cnt;
            st[1] = fmaxf(S2 - S1 * S1 / cnt, 0.f);
            st[2] = cnt;
        }
    }
}

bool wino2_eligible(const ConvArgs& a) {
    // SAME (pad 1: the VGG convs), VALID (pad 0: residual convs of the transform net) or FULL (pad 2: their input
    // gradients); the on-load affine needs pad 0
    const bool pad_ok = a.pad_t == a.pad_l && a.pad_t >= 0 && a.pad_t <= 2 && a.Ho == a.H + 2 * a.pad_t - 2 && a.Wo == a.W + 2 * a.pad_l - 2;
    // (measured: ahead of the first-generation kernel on the 64-channel layers -- 8 chunks per block, where the fixed cost of
    // a block weighs most --, level at 128 input channels, behind it beyond: FS_WINO2_MAXCIN)
    return a.w_wino2 && wino_gen().f2_second() && a.Cin <= tune_int("FS_WINO2_MAXCIN", 128) && a.KH == 3 && a.KW == 3 && a.stride == 1 && pad_ok && a.src_mode == SRC_PLAIN && a.Cin % kCC == 0 &&
           a.Cout % kBN == 0 && !a.shuffle && (!a.in_a || a.pad_t == 0) && a.w_nstride == 0 && (a.dil_x <= 1) &&
           (!a.add_src || !a.stats);
}

void wino2_plan(const ConvArgs& a, ConvPlan* out) {
    ConvPlan p{};
    p.variant = 6;
    p.BN = kBN;
    p.CC = kCC;
    p.TH = p.TW = 2 * kTT;
    p.tiles_y = cdiv(a.Ho, 2 * kTT);
    p.tiles_x = cdiv(a.Wo, 2 * kTT);
    p.lds_bytes = 4 * (2 * kStageF + kRedF);
    p.ksplit = 1;
    const long items = (long)a.N * p.tiles_y * p.tiles_x * (a.Cout / kBN);
    const int nchunks = a.Cin / kCC;
    const int max_ks = tune_int("FS_WINO_KSPLIT", 4);
    if (a.split_ws && !a.stats && !a.add_src) {
        int ks = 1;
        while (ks < max_ks && items * ks < 256 && nchunks / (ks * 2) >= 8 &&
               (size_t)(ks * 2) * a.N * a.Ho * a.Wo * a.Cout <= a.split_ws_floats)
            ks *= 2;
        p.ksplit = ks;
    }
    // remainder split (transform-net launches; see wino2_conv_kernel<true>): whole rounds of whole items, the last partial
    // round split over the input-channel chunks so that it occupies the whole chip for a fraction of a round
    if (a.rem_ws && a.tnet_plan && p.ksplit == 1 && !a.bias && !a.out_relu && !a.mask_src && !a.pool_out && !a.fin.counter && tune_int("FS_WINO2_REM", 1)) {
        const long G = tune_int("FS_WINO2_WGS", 256);
        if (items > G) {
            const long full = items / G * G, rem = items - full;
            int rks = 1;
            while (rks * 2 <= nchunks && rem * rks * 2 <= G) rks *= 2;
            if (rem > 0 && rks > 1 && (size_t)rem * rks * (4 * kNT * kBN) <= a.rem_ws_floats) {
                p.rem_full = (int)full;
                p.rem_ks = rks;
            }
        }
    }
    *out = p;
}

int wino2_launch(const ConvArgs& a, hipStream_t s) {
    const ConvPlan& p = a.p;
    const long items = (long)a.N * p.tiles_y * p.tiles_x * (a.Cout / kBN) * (p.ksplit > 1 ? p.ksplit : 1);
    const int wgs = tune_int("FS_WINO2_WGS", 256);
    if (p.rem_ks > 0) {   // whole items of the full rounds + units of the split remainder, then the remainder's epilogue
        static BigLds lds_attr_rem;
        lds_attr_rem.ensure(reinterpret_cast<const void*>(wino2_conv_kernel<true>));
        if (!a.rem_ws || p.rem_full <= 0 || p.rem_full % wgs) return -9;
        const long rem = items - p.rem_full;
        hipLaunchKernelGGL(wino2_conv_kernel<true>, dim3((unsigned)wgs), dim3(256), (size_t)p.lds_bytes, s, a);
        hipLaunchKernelGGL(wino2_rem_epilogue_kernel, dim3((unsigned)rem), dim3(1024), 0, s, a);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(wino2_conv_kernel<false>));
    const long grid = items < wgs ? items : wgs;
    hipLaunchKernelGGL(wino2_conv_kernel<false>, dim3((unsigned)grid), dim3(256), (size_t)p.lds_bytes, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
