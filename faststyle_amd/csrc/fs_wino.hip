// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores: the 3x3 stride-1 SAME convolutions of VGG16
// (reference libs/vgg16.py:36-220, forward, and their input gradients in the training step).
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A        (Lavin & Gray; cross-correlation form, as tf.nn.conv2d)
//
// A 4x4 input tile d (stride 2) yields a 2x2 output tile from 16 element-wise products instead of 36
// multiply-adds; summed over input channels the 16 products are 16 independent GEMMs
// [tiles x Cin] x [Cin x Cout] -- 2.25x less matrix work than the direct form, still fp32 throughout (the
// transforms only add, subtract and halve).  It is NOT bit-identical to the fmaf chain of the direct kernel:
// rounding differs at the 1e-7 level (tests hold both paths to the same tolerance against the fp64 oracle).
//
// Mapping.  512 threads (8 waves, two per SIMD).  A workgroup owns 8x8 tiles (16x16 output pixels) x 64 output
// channels and walks the input channels in chunks of 8:
//   * the 18x18 input patch of the chunk is staged in LDS ([pixel][8+1]), one thread transforms one (tile, channel)
//     4x4 block into V[16][64 tiles][8+1];
//   * the pre-transformed filter chunk U[16][8][64] (fs::wt_wino, built once per weight set) is staged beside it;
//   * wave w multiplies positions 2w and 2w+1: M = 64 tiles (2 MFMA row blocks), N = 64 channels (2 column blocks),
//     K = the chunk -- 8 accumulator blocks of v_mfma_f32_32x32x2_f32 per wave, 32 MFMAs per chunk and wave;
//   * two LDS stages: chunk j is multiplied while chunk j+1 is transformed / committed into the other stage -- one
//     slice of that work per MFMA slot of the sweep -- and the global loads of chunk j+2 are in flight (registers); one
//     barrier per chunk;
//   * epilogue: the 16 position planes are exchanged through LDS 32 channels at a time, one thread applies
//     A^T . A to four (tile, channel) pairs and stores their 2x2 pixels (bias / ReLU / consumer mask, raw split-K
//     partials, or -- transform-net form -- per-block instance-norm statistics);
//   * variants by argument: SAME padding (VGG16) or VALID padding with the producer's instance norm + ReLU applied
//     on load (residual convs of the transform net, reference im_transf_net.py:250-276).
#include "fs_kernels.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace fs {

namespace {
constexpr int kTT = 8;              // tiles per side of a workgroup block
constexpr int kPT = 2 * kTT + 2;    // patch side (18)
constexpr int kCC = 8;              // input channels per chunk
constexpr int kPS = kCC + 1;        // LDS pitch of patch pixels and of V rows
constexpr int kBN = 64;             // output channels per workgroup
constexpr int kNT = kTT * kTT;      // tiles per workgroup (64)
constexpr int kPatchFloats = (kPT * kPT * kPS + 7) & ~7;   // 2920 (+4 slack used as the zero sink)
constexpr int kVFloats = 16 * kNT * kPS;                   // 9216
constexpr int kUFloats = 16 * kCC * kBN;                   // 8192
}  // namespace

// U[pos][ci][co] = (G g G^T)[pos] for g = w[:, :, ci, co]   (w HWIO [3][3][Cin][Cout]); blockIdx.y = filter of the batch
__global__ __launch_bounds__(256) void wt_wino_batch_kernel(WinoBatch b, int Cin, int Cout) {
    const float* __restrict__ w = b.w[blockIdx.y];
    float* __restrict__ U = b.U[blockIdx.y];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t cc = (size_t)Cin * Cout;
    if (i >= cc) return;
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
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        U[(size_t)(r * 4 + 0) * cc + i] = t[r][0];
        U[(size_t)(r * 4 + 1) * cc + i] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
        U[(size_t)(r * 4 + 2) * cc + i] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
        U[(size_t)(r * 4 + 3) * cc + i] = t[r][2];
    }
}

int wt_wino_batch(const WinoBatch& b, int Cin, int Cout, hipStream_t s) {
    if (b.n <= 0) return 0;
    const size_t cc = (size_t)Cin * Cout;
    hipLaunchKernelGGL(wt_wino_batch_kernel, dim3((unsigned)((cc + 255) / 256), (unsigned)b.n), dim3(256), 0, s, b, Cin, Cout);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int wt_wino(const float* w, float* U, int Cin, int Cout, hipStream_t s) {
    WinoBatch b{};
    b.w[0] = w;
    b.U[0] = U;
    b.n = 1;
    return wt_wino_batch(b, Cin, Cout, s);
}

#ifdef FS_CONV_TRACE
extern __device__ long long g_conv_trace[4096 * 8];   // fs_conv.hip (tools/conv_trace.py)
#define FS_WINO_NOW() ((long long)__builtin_readcyclecounter())
#endif

__global__ __launch_bounds__(512) void wino_conv_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
#ifdef FS_CONV_TRACE
    const long long tr_t0 = FS_WINO_NOW();
    long long tr_sweep = 0, tr_commit = 0, tr_bar = 0;
#endif
    // two stages each of patch | V | U (162,624 bytes of the CU's 160 KiB); the epilogue's exchange buffer overlays V
    float* patch0 = smem;
    float* Vl0 = smem + 2 * kPatchFloats;
    float* Ul0 = Vl0 + 2 * kVFloats;
    const ConvPlan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 31, kq = lane >> 5;
    auto fdiv = [](int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); };
    const int blocks = p.tiles_y * p.tiles_x;
    const int n = __builtin_amdgcn_readfirstlane(fdiv((int)blockIdx.x, 1.0f / (float)blocks));
    const int br = (int)blockIdx.x - n * blocks;
    const int byi = __builtin_amdgcn_readfirstlane(fdiv(br, 1.0f / (float)p.tiles_x));
    const int oy0 = byi * 2 * kTT, ox0 = (br - byi * p.tiles_x) * 2 * kTT;   // first output pixel of the block
    const int co0 = (int)blockIdx.y * kBN;
    auto uniform_ptr = [](const float* ptr) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };
    const float* xn = uniform_ptr(a.x + (size_t)n * a.H * a.W * a.Cin);
    const float* ub = uniform_ptr(a.w_wino);
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * a.Cin) * 4u);
    const unsigned u_bytes = __builtin_amdgcn_readfirstlane((unsigned)(16 * a.Cin * a.Cout) * 4u);
    constexpr unsigned kOOB = 0x80000000u;

    // ---- staging descriptors (chunk-invariant) ----
    // patch: 18*18 pixels x 2 float4 = 648 elements, <= 2 per thread; zero padding / unowned -> kOOB (loads zeros)
    unsigned gvo[2];
    int pdst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * 512;
        gvo[i] = kOOB;
        pdst[i] = kPT * kPT * kPS;  // slack
        if (e < kPT * kPT * 2) {
            const int pix = e >> 1, c4 = e & 1;
            const int py = fdiv(pix, 1.0f / (float)kPT), px = pix - py * kPT;
            const int sy = oy0 - a.pad_t + py, sx = ox0 - a.pad_l + px;
            if (sy >= 0 && sy < a.H && sx >= 0 && sx < a.W) gvo[i] = (unsigned)((sy * a.W + sx) * a.Cin + c4 * 4) * 4u;
            pdst[i] = pix * kPS + c4 * 4;
        }
    }
    // filter: 16 positions x 8 rows x 16 float4 = 2048 elements, 4 per thread; LDS position = element index
    unsigned uvo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * 512;
        const int pos = e >> 7, k = (e >> 4) & 7, c4 = e & 15;
        uvo[i] = (unsigned)((pos * a.Cin + k) * a.Cout + co0 + c4 * 4) * 4u;
    }
    float4 pv[2], uv[4];
    // producer instance norm + ReLU folded into the load (transform-net convs; VALID padding only, so every patch pixel
    // that reaches a stored output is a real pixel and padding needs no masking).  Both elements of a thread belong
    // to the same channel quad (tid & 1), so one float4 of scales and of shifts per chunk.
    const bool has_ab = a.in_a != nullptr;
    const float* ian = has_ab ? uniform_ptr(a.in_a + (size_t)n * a.in_nstride) : nullptr;
    const float* ibn = has_ab ? uniform_ptr(a.in_b + (size_t)n * a.in_nstride) : nullptr;
    float4 va = make_float4(1.f, 1.f, 1.f, 1.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
    auto issue_patch = [&](int chunk) {
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(xn)), 0, x_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, gvo[i], chunk * kCC * 4, 0));
        if (has_ab) {
            va = *reinterpret_cast<const float4*>(ian + chunk * kCC + (tid & 1) * 4);
            vb = *reinterpret_cast<const float4*>(ibn + chunk * kCC + (tid & 1) * 4);
        }
    };
    auto issue_filter = [&](int chunk) {
        const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(ub)), 0, u_bytes, 0x00020000);
        const int uso = chunk * kCC * a.Cout * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) uv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ur, uvo[i], uso, 0));
    };
    auto commit_patch = [&](float* patch) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 v = pv[i];
            if (has_ab) {
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
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(Ul + (tid + i * 512) * 4) = uv[i];
    };
    // input transform: thread = (tile, channel of the chunk);  V = B^T d B
    const int tt = tid >> 3, tk = tid & 7;
    const int tty = tt >> 3, ttx = tt & 7;
    auto transform = [&](const float* patch, float* Vl) {
        const float* src = patch + ((2 * tty) * kPT + 2 * ttx) * kPS + tk;
        float d[4][4], r[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = src[(i * kPT + j) * kPS];
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // B^T d  (rows)
            r[0][j] = d[0][j] - d[2][j];
            r[1][j] = d[1][j] + d[2][j];
            r[2][j] = d[2][j] - d[1][j];
            r[3][j] = d[1][j] - d[3][j];
        }
        float* dst = Vl + tt * kPS + tk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // (.) B  (columns)
            dst[(i * 4 + 0) * kNT * kPS] = r[i][0] - r[i][2];
            dst[(i * 4 + 1) * kNT * kPS] = r[i][1] + r[i][2];
            dst[(i * 4 + 2) * kNT * kPS] = r[i][2] - r[i][1];
            dst[(i * 4 + 3) * kNT * kPS] = r[i][1] - r[i][3];
        }
    };

    f32x16 acc[2][2][2];  // [position of the wave][tile block][channel block]
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pp][m][nn][r] = 0.f;

    const int nchunks = a.Cin / kCC;
    const int cbeg = __builtin_amdgcn_readfirstlane(p.ksplit > 1 ? (int)blockIdx.z * nchunks / p.ksplit : 0);
    const int cend = __builtin_amdgcn_readfirstlane(p.ksplit > 1 ? ((int)blockIdx.z + 1) * nchunks / p.ksplit : nchunks);

    // sweep(chunk j) with the transform of chunk j+1 threaded through it: the wave's own VALU / LDS instructions issue
    // in the shadow of its own matrix instructions (another wave's do not: MFMA issue is in order and a streaming wave
    // leaves its SIMD partner almost no issue slots).  sched_barrier pins the interleaving.
    auto sweep_fused = [&](auto PREP, const float* Vl, const float* Ul, const float* patch, float* Vn, float* Un) {
        constexpr bool prep = decltype(PREP)::value;
        float av[2][2], bv[2][2];
        auto load_a = [&](int g, float (&a2)[2]) {
            const int k = (g >> 1) * 2 + kq, pos = wave * 2 + (g & 1);
            const float* pa = Vl + (pos * kNT + lm) * kPS + k;
            a2[0] = pa[0];
            a2[1] = pa[32 * kPS];
        };
        auto load_b = [&](int g, float (&b2)[2]) {
            const int k = (g >> 1) * 2 + kq, pos = wave * 2 + (g & 1);
            const float* pb = Ul + (pos * kCC + k) * kBN + lm;
            b2[0] = pb[0];
            b2[1] = pb[32];
        };
        f32x2 d[4][2], r[4][2];   // column pairs: the arithmetic runs as packed fp32 (fs_pk_*, fs_wino_cols*: 16 instructions, not 32)
        const float* src = patch + ((2 * tty) * kPT + 2 * ttx) * kPS + tk;
        float* dst = Vn + tt * kPS + tk;
        // one slice of the next chunk's preparation per MFMA slot (g = group 0..7, h = 0: after the 3rd MFMA of the group,
        // h = 1: after the 4th); every slice is a handful of instructions, about what one 64-cycle MFMA covers
        auto slice = [&](int g, int h) {
            if (!prep) return;
            if (g == 0) {                                     // the 4x4 block of the next chunk
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d[i][h].x = src[(i * kPT + 2 * h) * kPS];
                    d[i][h].y = src[(i * kPT + 2 * h + 1) * kPS];
                }
            } else if (g == 1) {                              // B^T d (rows), two columns per slice
                r[0][h] = fs_pk_sub(d[0][h], d[2][h]);
                r[1][h] = fs_pk_add(d[1][h], d[2][h]);
                r[2][h] = fs_pk_sub(d[2][h], d[1][h]);
                r[3][h] = fs_pk_sub(d[1][h], d[3][h]);
            } else if (g <= 5) {                              // (.) B (columns): row i of positions, two outputs per slice
                const int i = g - 2;
                if (h == 0) {
                    const f32x2 o = fs_wino_cols01(r[i][0], r[i][1]);
                    dst[(i * 4 + 0) * kNT * kPS] = o.x;
                    dst[(i * 4 + 1) * kNT * kPS] = o.y;
                } else {
                    const f32x2 o = fs_wino_cols23(r[i][0], r[i][1]);
                    dst[(i * 4 + 2) * kNT * kPS] = o.x;
                    dst[(i * 4 + 3) * kNT * kPS] = o.y;
                }
            } else if (g == 6) {                              // the next chunk's filter
                *reinterpret_cast<float4*>(Un + (tid + (2 * h) * 512) * 4) = uv[2 * h];
                *reinterpret_cast<float4*>(Un + (tid + (2 * h + 1) * 512) * 4) = uv[2 * h + 1];
            }
        };
        load_a(0, av[0]);
        load_b(0, bv[0]);
#pragma unroll
        for (int g = 0; g < kCC; ++g) {
            const int pp = g & 1, c = g & 1, nx = (g + 1) & 1;
            acc[pp][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][0], bv[c][0], acc[pp][0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < kCC) load_a(g + 1, av[nx]);
            __builtin_amdgcn_sched_barrier(0);
            acc[pp][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][0], bv[c][1], acc[pp][0][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < kCC) load_b(g + 1, bv[nx]);
            __builtin_amdgcn_sched_barrier(0);
            acc[pp][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][1], bv[c][0], acc[pp][1][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            slice(g, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc[pp][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][1], bv[c][1], acc[pp][1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            slice(g, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Pipeline.  Chunk j is multiplied out of stage j&1 while chunk j+1 is prepared (patch -> V transform, filter
    // commit) into the other stage and the global loads of chunk j+2 are in flight; one barrier per chunk.
    // (Measured: letting the two waves of a SIMD do "multiply" and "prepare" in opposite order changes nothing --
    // with two waves per SIMD, MFMA issue and the other wave's VALU work do not overlap; what counts is the number
    // of non-MFMA instructions per chunk.)
    const int nloc = cend - cbeg;
    issue_patch(cbeg);
    issue_filter(cbeg);
    commit_patch(patch0);
    commit_filter(Ul0);
    if (nloc > 1) {
        issue_patch(cbeg + 1);
        issue_filter(cbeg + 1);   // stays in registers until the first iteration prepares chunk 1
    }
    __syncthreads();
    transform(patch0, Vl0);
    if (nloc > 1) commit_patch(patch0 + kPatchFloats);
    __syncthreads();
#ifdef FS_CONV_TRACE
    const long long tr_pro = FS_WINO_NOW();
#endif
    for (int j = 0; j < nloc; ++j) {
        const int cur = j & 1, nxt = cur ^ 1;
        const bool has1 = j + 1 < nloc, has2 = j + 2 < nloc;
        const float* Vc = Vl0 + cur * kVFloats;
        const float* Uc = Ul0 + cur * kUFloats;
        float* Vn = Vl0 + nxt * kVFloats;
        float* Un = Ul0 + nxt * kUFloats;
#ifdef FS_CONV_TRACE
        const long long q0 = FS_WINO_NOW();
#endif
        if (has2) issue_patch(cbeg + j + 2);
        // (also on the last chunk, where the prepared stage is never read: a second, transform-free instantiation
        // makes the register allocator copy the 128 accumulators at the join and spill)
        sweep_fused(std::true_type{}, Vc, Uc, patch0 + nxt * kPatchFloats, Vn, Un);
#ifdef FS_CONV_TRACE
        tr_sweep += FS_WINO_NOW() - q0;
#endif
        if (has2) {
            issue_filter(cbeg + j + 2);
            commit_patch(patch0 + cur * kPatchFloats);
        }
#ifdef FS_CONV_TRACE
        const long long q4 = FS_WINO_NOW();
#endif
        __syncthreads();
#ifdef FS_CONV_TRACE
        const long long q5 = FS_WINO_NOW();
        tr_commit += q4 - q0;
        tr_bar += q5 - q4;
#endif
    }
#ifdef FS_CONV_TRACE
    tr_commit -= tr_sweep;
    const long long tr_main = FS_WINO_NOW();
#endif

    // ---- epilogue: exchange the 16 position planes through LDS 32 channels at a time ([16][64 tiles][32+1] floats
    // over the V and U stages), output transform A^T M A per (tile, channel), store ----
    constexpr int kMP = 33;
    float* Ml = Vl0;
    const float* bias = a.bias;
    const bool relu_out = a.out_relu != 0;
    float* yn = a.y + ((size_t)n + (p.ksplit > 1 ? (size_t)blockIdx.z * a.N : 0)) * a.Ho * a.Wo * a.Cout;
    const float* msn = a.mask_src ? a.mask_src + (size_t)n * a.Ho * a.Wo * a.Cout : nullptr;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        // accumulator element r of lane l: tile row (r&3) + 8*(r>>2) + 4*kq of its block, channel lm of block q
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                    Ml[((wave * 2 + pp) * kNT + t) * kMP + lm] = acc[pp][m][q][r];
                }
        // this thread's 4 (tile, channel) pairs x 2x2 pixels: offsets first, and ALL consumer-mask loads issued before
        // the barrier (unconditional, clamped to element 0 for pixels outside the image) -- per-element
        // "load, wait, store" chains would cost a global-memory latency each
        const int c32 = tid & 31;
        const int co = co0 + q * 32 + c32;
        const float bs = bias ? bias[co] : 0.f;
        int off[4][4];
        float mk[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = (tid >> 5) + i * 16;
            const int oy = oy0 + 2 * (t >> 3), ox = ox0 + 2 * (t & 7);
            const int o00 = (oy * a.Wo + ox) * a.Cout + co;
            const bool y0 = oy < a.Ho, y1 = oy + 1 < a.Ho, x0 = ox < a.Wo, x1 = ox + 1 < a.Wo;
            off[i][0] = (y0 && x0) ? o00 : -1;
            off[i][1] = (y0 && x1) ? o00 + a.Cout : -1;
            off[i][2] = (y1 && x0) ? o00 + a.Wo * a.Cout : -1;
            off[i][3] = (y1 && x1) ? o00 + (a.Wo + 1) * a.Cout : -1;
            if (msn) {
#pragma unroll
                for (int k = 0; k < 4; ++k) mk[i][k] = msn[off[i][k] >= 0 ? off[i][k] : 0];
            }
        }
        __syncthreads();
        float vals[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = (tid >> 5) + i * 16;
            float mm[4][4];
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) mm[pos >> 2][pos & 3] = Ml[(pos * kNT + t) * kMP + c32];
            float s4[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // A^T M
                s4[0][j] = mm[0][j] + mm[1][j] + mm[2][j];
                s4[1][j] = mm[1][j] - mm[2][j] - mm[3][j];
            }
#pragma unroll
            for (int ai = 0; ai < 2; ++ai) {
                vals[i][ai * 2 + 0] = s4[ai][0] + s4[ai][1] + s4[ai][2];
                vals[i][ai * 2 + 1] = s4[ai][1] - s4[ai][2] - s4[ai][3];
            }
        }
        if (a.stats) {
            // per-block instance-norm partials of the RAW conv output (mean, M2, count), one pass around a shift taken
            // from the block's first pixel -- same record the direct kernel writes (fs_conv.hip)
            float* shiftl = patch0;            // [32]   (the patch stages are free by now)
            float* red = patch0 + 32;          // [16][32][2]
            if ((tid >> 5) == 0) shiftl[c32] = vals[0][0];
            __syncthreads();
            const float cs = shiftl[c32];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dv = off[i][k] >= 0 ? vals[i][k] - cs : 0.f;
                    s1 += dv;
                    s2 = fmaf(dv, dv, s2);
                }
            red[((tid >> 5) * 32 + c32) * 2] = s1;
            red[((tid >> 5) * 32 + c32) * 2 + 1] = s2;
            __syncthreads();
            if (tid < 32) {
                float S1 = 0.f, S2 = 0.f;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    S1 += red[(g * 32 + tid) * 2];
                    S2 += red[(g * 32 + tid) * 2 + 1];
                }
                const int th_valid = min(2 * kTT, a.Ho - oy0), tw_valid = min(2 * kTT, a.Wo - ox0);
                const float cnt = (float)(th_valid * tw_valid);
                float* st = a.stats + ((size_t)blockIdx.x * a.Cout + co0 + q * 32 + tid) * 3;
                st[0] = cs + S1 / cnt;
                st[1] = fmaxf(S2 - S1 * S1 / cnt, 0.f);
                st[2] = cnt;
            }
        }
        float* pon = a.pool_out ? a.pool_out + (size_t)n * (a.Ho >> 1) * (a.Wo >> 1) * a.Cout : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float pmax = -3.0e38f;   // max of the tile's four stored values: the 2x2/2 max-pool window (tiles sit on even coordinates)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = vals[i][k] + bs;
                v = relu_out ? fmaxf(v, 0.f) : v;
                if (msn) v = mk[i][k] > 0.f ? v : 0.f;
                pmax = fmaxf(pmax, v);
                if (off[i][k] >= 0) yn[off[i][k]] = v;
            }
            if (pon && off[i][0] >= 0) {
                const int t = (tid >> 5) + i * 16;
                const int qy = (oy0 >> 1) + (t >> 3), qx = (ox0 >> 1) + (t & 7);
                pon[(qy * (a.Wo >> 1) + qx) * a.Cout + co] = pmax;
            }
        }
        if (q == 0) __syncthreads();
    }
#ifdef FS_CONV_TRACE
    {
        const long long tr_end = FS_WINO_NOW();
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (tid == 0 && lin < 4096) {
            long long* t = g_conv_trace + lin * 8;
            t[0] = tr_t0;
            t[1] = tr_pro - tr_t0;
            t[2] = tr_sweep;
            t[3] = tr_commit;
            t[4] = tr_bar;
            t[5] = tr_end - tr_main;
            t[6] = tr_end;
            t[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
#endif
}

bool wino_eligible(const ConvArgs& a) {
    // SAME (pad 1, the VGG convs) or VALID (pad 0, the residual convs of the transform net); the on-load affine needs
    // pad 0; statistics only without split-K (the plan takes care of that)
    const bool pad_ok = a.pad_t == a.pad_l && (a.pad_t == 0 || a.pad_t == 1) && a.Ho == a.H + 2 * a.pad_t - 2 &&
                        a.Wo == a.W + 2 * a.pad_l - 2;
    return a.w_wino && a.KH == 3 && a.KW == 3 && a.stride == 1 && pad_ok && a.src_mode == SRC_PLAIN && a.Cin % kCC == 0 &&
           a.Cout % kBN == 0 && !a.shuffle && !a.add_src && (!a.in_a || a.pad_t == 0) && a.w_nstride == 0 && (a.dil_x <= 1);
}

// plan: 16x16-pixel blocks; split the channel chunks over blockIdx.z while the launch cannot fill the CUs (one
// workgroup of 8 waves per CU) and scratch for the partials is available
void wino_plan(const ConvArgs& a, ConvPlan* out) {
    ConvPlan p{};
    p.variant = 5;
    p.BN = kBN;
    p.CC = kCC;
    p.TH = p.TW = 2 * kTT;
    p.tiles_y = cdiv(a.Ho, 2 * kTT);
    p.tiles_x = cdiv(a.Wo, 2 * kTT);
    p.lds_bytes = 4 * 2 * (kPatchFloats + kVFloats + kUFloats);
    p.ksplit = 1;
    const long wgs = (long)a.N * p.tiles_y * p.tiles_x * (a.Cout / kBN);
    const int nchunks = a.Cin / kCC;
    const int max_ks = tune_int("FS_WINO_KSPLIT", 4);  // tuning / debugging aid
    if (a.split_ws && !a.stats) {
        int ks = 1;
        while (ks < max_ks && wgs * ks < 256 && nchunks / (ks * 2) >= 8 &&
               (size_t)(ks * 2) * a.N * a.Ho * a.Wo * a.Cout <= a.split_ws_floats)
            ks *= 2;
        p.ksplit = ks;
    }
    *out = p;
}

int wino_launch(const ConvArgs& a, hipStream_t s) {
    const ConvPlan& p = a.p;
    // > 64 KiB of dynamic LDS needs the attribute once per device (a per-function property of the loaded code object)
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(wino_conv_kernel));
    dim3 grid((unsigned)(a.N * p.tiles_y * p.tiles_x), (unsigned)(a.Cout / kBN), (unsigned)(p.ksplit > 1 ? p.ksplit : 1));
    hipLaunchKernelGGL(wino_conv_kernel, grid, dim3(512), (size_t)p.lds_bytes, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
