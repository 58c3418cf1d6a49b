// Winograd filter gradients F(3x3, 2x2) on the fp32 matrix cores (round 4): dW of the 3x3 stride-1 VALID 64 -> 64 convs -- the ten
// residual convs of the transform net (reference im_transf_net.py:250-276; their filter gradients are the largest launch of
// train.py:203's gradient graph outside VGG16).
//
//   dW = A^T [ sum over tiles (B^T x B) . (G dz G^T) ] A
//
// The filter gradient of a 2x2 tile of dz against its 4x4 input tile is a correlation with a 3x3 result: nested F(3, 2) with the
// interpolation points 0, 1, -1, inf -- 16 products per (tile, ci, co) instead of the 36 multiply-adds of the direct form, i.e.
// 16 independent GEMMs M[pos][ci][co] += V[pos][tile][ci] * D[pos][tile][co] whose reduction dimension is the TILES (pixels / 4)
// and 2.25x less matrix work.  Unlike the forward Winograd kernels the transforms are cheap here: every transformed element is
// used by all 64 channels of the other operand.  B^T (input) is F(2x2,3x3)'s, G = [[1,0],[1/2,1/2],[1/2,-1/2],[0,1]] (applied to
// dz), A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]] (applied ONCE, in the slab reduction).  The transforms add, subtract and halve:
// fp32 rounding as the direct kernel's (tests hold it to the same 2e-5).
//
// Mapping: persistent workgroups, 256 threads, one wave per SIMD; wave w keeps positions 4w .. 4w+3 of M as 4 x (2 x 2) blocks of
// v_mfma_f32_32x32x2_f32 in the accumulator file (256 registers) for the workgroup's whole tile range -- one 256 KB slab per
// workgroup at the end.  A step is 16 tiles: thread (tile, channel quad) loads its 4x4 input tile (16 x 16 bytes) and its 2x2 dz
// tile (4 x 16 bytes) during the previous step's sweep, applies the producer's instance norm + ReLU, transforms in registers
// and writes V[16][16 tiles][64] / D[16][16 tiles][64] (64 KB each, one LDS stage) as 16-byte stores; the sweep is 128 matrix
// instructions per wave, operands by 4-byte LDS reads at one lane base + immediates.  The ten residual problems of a step are
// ONE launch (workgroups dealt to problems), followed by one reduction that sums the slabs in a fixed order (deterministic) and
// applies A^T . A.
#include "fs_kernels.h"

#include <cstdlib>

namespace fs {

namespace {
constexpr int kST = 16;                       // tiles per step
constexpr int kC = 64;                        // channels (both sides)
constexpr int kVF = 16 * kST * kC;            // floats of V (and of D) per stage
constexpr unsigned kOOB = 0x80000000u;
}  // namespace

// F(2x2,3x3)'s input transform of one channel: v = B^T d B, d / v row-major 4x4
__device__ __forceinline__ void wgw_bt(const float (&d)[16], float (&v)[16]) {
    float r[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[0 + j] = d[0 + j] - d[8 + j];
        r[4 + j] = d[4 + j] + d[8 + j];
        r[8 + j] = d[8 + j] - d[4 + j];
        r[12 + j] = d[4 + j] - d[12 + j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[4 * i + 0] = r[4 * i + 0] - r[4 * i + 2];
        v[4 * i + 1] = r[4 * i + 1] + r[4 * i + 2];
        v[4 * i + 2] = r[4 * i + 2] - r[4 * i + 1];
        v[4 * i + 3] = r[4 * i + 1] - r[4 * i + 3];
    }
}
// D = G g G^T of one channel: g row-major 2x2 -> 4x4
__device__ __forceinline__ void wgw_g(const float (&g)[4], float (&o)[16]) {
    float t[8];   // G g: 4x2
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        t[0 + j] = g[0 + j];
        t[2 + j] = 0.5f * (g[0 + j] + g[2 + j]);
        t[4 + j] = 0.5f * (g[0 + j] - g[2 + j]);
        t[6 + j] = g[2 + j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[4 * i + 0] = t[2 * i];
        o[4 * i + 1] = 0.5f * (t[2 * i] + t[2 * i + 1]);
        o[4 * i + 2] = 0.5f * (t[2 * i] - t[2 * i + 1]);
        o[4 * i + 3] = t[2 * i + 1];
    }
}

__global__ __launch_bounds__(256) void wgw_kernel(WgwArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    const WgwArgs* ka = FS_KERNARG_PTR(WgwArgs, a);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, kq = lane >> 5;
    // ---- this workgroup's problem and step range
    int pi = 0;
    for (int i = 1; i < ka->nprob; ++i)
        if ((int)blockIdx.x >= ka->prob[i].wg_begin) pi = i;
    pi = __builtin_amdgcn_readfirstlane(pi);
    const WgwProb& P = ka->prob[pi];
    const int wi = (int)blockIdx.x - P.wg_begin;
    const int g_beg = (int)((long long)P.steps * wi / P.wg_count), g_end = (int)((long long)P.steps * (wi + 1) / P.wg_count);
    const int H = P.H, W = P.W, Ho = P.Ho, Wo = P.Wo, Tx = P.Tx, tiles = P.Ty * P.Tx, sps = P.sps;
    const float inv_sps = 1.0f / (float)sps, inv_tx = 1.0f / (float)Tx;
    auto fdiv = [](int x, float inv_d) __attribute__((always_inline)) { return (int)(((float)x + 0.5f) * inv_d); };
    auto uniform_ptr = [](const float* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)((size_t)P.N * H * W * kC * 4));
    const unsigned d_bytes = __builtin_amdgcn_readfirstlane((unsigned)((size_t)P.N * Ho * Wo * kC * 4));
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(P.x)), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(P.dy)), 0, d_bytes, 0x00020000);
    const bool has_ab = P.in_a != nullptr;

    // ---- staging: thread = (tile of the step, channel quad)
    const int tl = tid >> 4, cq = tid & 15;
    float4 xv[16], dv[4], av = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
    auto issue = [&](int g) __attribute__((always_inline)) {
        const int n = fdiv(g, inv_sps);
        const int t = (g - n * sps) * kST + tl;
        const int ty = fdiv(t, inv_tx), tx = t - ty * Tx;
        const bool valid = t < tiles;
        const int y0 = 2 * ty, x0 = 2 * tx;
        const unsigned xb = (unsigned)(((n * H + y0) * W + x0) * kC + 4 * cq) * 4u;
        const unsigned db = (unsigned)(((n * Ho + y0) * Wo + x0) * kC + 4 * cq) * 4u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = valid && y0 + i < H && x0 + j < W;
                // (the offset is made opaque: the compiler otherwise turns some of these selects into branches with ONE LOAD PER ARM into the same
                // registers, and the wait-count pass guards the second arm with s_waitcnt vmcnt(0) -- the loads just issued drained inside the issue phase)
                unsigned vo = ok ? xb : kOOB;
                FS_OPAQUE(vo);
                xv[4 * i + j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, vo, (unsigned)((i * W + j) * kC * 4), 0));
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = valid && y0 + i < Ho && x0 + j < Wo;
                unsigned vo = ok ? db : kOOB;
                FS_OPAQUE(vo);
                dv[2 * i + j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(dr, vo, (unsigned)((i * Wo + j) * kC * 4), 0));
            }
        if (has_ab) {   // the producer's instance norm (+ ReLU) of this sample's channels, applied in `commit`
            av = *reinterpret_cast<const float4*>(P.in_a + (size_t)n * ka->in_nstride + 4 * cq);
            bv = *reinterpret_cast<const float4*>(P.in_b + (size_t)n * ka->in_nstride + 4 * cq);
        }
    };
    // transform the loaded tiles in registers and write V / D: [pos][tile][channel], one 16-byte store per position and operand.
    // (Input pixels beyond the image arrive as zeros and become relu(b) under the affine: finite garbage that only meets
    // dz = 0 -- every product with it is exactly zero.)
    float* const Vl = smem;
    float* const Dl = smem + kVF;
    auto commit = [&]() __attribute__((always_inline)) {
        float vo[4][16];
        const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float d[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float raw = c == 0 ? xv[k].x : (c == 1 ? xv[k].y : (c == 2 ? xv[k].z : xv[k].w));
                float v = raw;
                if (has_ab) {
                    v = fmaf(raw, a4[c], b4[c]);
                    if (ka->in_relu) v = fmaxf(v, 0.f);
                }
                d[k] = v;
            }
            wgw_bt(d, vo[c]);
        }
        float* vd = Vl + tl * kC + 4 * cq;
#pragma unroll
        for (int k = 0; k < 16; ++k) *reinterpret_cast<float4*>(vd + k * (kST * kC)) = make_float4(vo[0][k], vo[1][k], vo[2][k], vo[3][k]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = c == 0 ? dv[k].x : (c == 1 ? dv[k].y : (c == 2 ? dv[k].z : dv[k].w));
            wgw_g(g, vo[c]);
        }
        float* dd = Dl + tl * kC + 4 * cq;
#pragma unroll
        for (int k = 0; k < 16; ++k) *reinterpret_cast<float4*>(dd + k * (kST * kC)) = make_float4(vo[0][k], vo[1][k], vo[2][k], vo[3][k]);
    };

    f32x16 acc[4][2][2];   // [position of the wave][ci block][co block]
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][m][nn][r] = 0.f;

    // sweep of one step: 8 tile pairs x 4 positions x 2 x 2 blocks = 128 matrix instructions; the operand reads of the next
    // (pair, position) ride behind the first instruction of the current one; the 20 global loads of the NEXT step are
    // issued one per group
    const float* pa = Vl + ((4 * wave) * kST + kq) * kC + lm;
    const float* pb = Dl + ((4 * wave) * kST + kq) * kC + lm;
    auto sweep = [&]() __attribute__((always_inline)) {
        float A[2][2], B[2][2];
        A[0][0] = FS_LDS_LOAD1(pa);   // (unpaired 4-byte reads, 16-bit immediates: no address arithmetic between the matrix instructions)
        A[0][1] = FS_LDS_LOAD1(pa + 32);
        B[0][0] = FS_LDS_LOAD1(pb);
        B[0][1] = FS_LDS_LOAD1(pb + 32);
#pragma unroll
        for (int q = 0; q < 32; ++q) {   // q = pair j (0..7) * 4 + position p
            const int c = q & 1, nx = c ^ 1, p = q & 3;
            acc[p][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c][0], B[c][0], acc[p][0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < 32) {
                const int off = ((q + 1) & 3) * (kST * kC) + ((q + 1) >> 2) * (2 * kC);
                A[nx][0] = FS_LDS_LOAD1(pa + off);
                A[nx][1] = FS_LDS_LOAD1(pa + off + 32);
                B[nx][0] = FS_LDS_LOAD1(pb + off);
                B[nx][1] = FS_LDS_LOAD1(pb + off + 32);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[p][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c][0], B[c][1], acc[p][0][1], 0, 0, 0);
            acc[p][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c][1], B[c][0], acc[p][1][0], 0, 0, 0);
            acc[p][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c][1], B[c][1], acc[p][1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (g_beg < g_end) {
        issue(g_beg);
        commit();
        FS_WAIT_VMEM_FENCED();
        __syncthreads();
        for (int g = g_beg; g < g_end; ++g) {
            const bool more = g + 1 < g_end;
            if (more) issue(g + 1);   // (in flight during the sweep)
            sweep();
            __syncthreads();
            if (more) commit();
            // (free: commit consumed every load of this step.  Without it the loads of the conditional `issue` stay "possibly pending" at the loop
            // header and the next issue phase drains its own first loads -- s_waitcnt vmcnt(0) twice among its 22 loads -- before the sweep; fs_kernels.h)
            FS_WAIT_VMEM_FENCED();
            __syncthreads();
        }
    }
    // ---- the workgroup's slab: M[pos][ci][co]; accumulator register r of lane (lm, kq): ci = m*32 + (r & 3) + 8 (r >> 2) + 4 kq,
    // co = nn*32 + lm.  (Workgroups without steps write zeros: the reduction reads every slab.)
    float* slab = ka->slabs + P.slab_off + (size_t)wi * (16 * kC * kC);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq, co = nn * 32 + lm;
                    slab[((size_t)(4 * wave + p) * kC + ci) * kC + co] = acc[p][m][nn][r];
                }
}

// dW[kh][kw][ci][co] = scale * (A^T M A)[kh][kw], M = the fixed-order sum of the problem's slabs.  A workgroup owns 64 consecutive (ci, co) pairs;
// its four waves each sum every fourth slab (16 independent 4-byte loads per slab and thread, 256 KB apart), the four partial sums meet through
// LDS in a fixed order.  (Round 4 had one thread walk all ~26 slabs of a pair: 160 workgroups of dependent round trips, 99 us at batch 32 for
// 65 MB -- the same bytes now move in a quarter of the trips on four times the workgroups.)
__global__ __launch_bounds__(256) void wgw_reduce_kernel(WgwReduce r) {
    __shared__ float sh[3][16][64];
    const WgwReduce* kr = FS_KERNARG_PTR(WgwReduce, r);
    const WgwReduce::Job& J = kr->job[blockIdx.y];
    const int pl = (int)threadIdx.x & 63, sg = (int)threadIdx.x >> 6;
    const int i = (int)blockIdx.x * 64 + pl;   // ci * 64 + co
    float m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = 0.f;
    for (int s = sg; s < J.n_slabs; s += 4) {
        const float* sl = J.slabs + (size_t)s * (16 * kC * kC) + i;
#pragma unroll
        for (int k = 0; k < 16; ++k) m[k] += sl[(size_t)k * (kC * kC)];
    }
    if (sg > 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) sh[sg - 1][k][pl] = m[k];
    }
    __syncthreads();
    if (sg > 0) return;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 16; ++k) m[k] += sh[g][k][pl];
    float t[3][4];   // A^T M
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = m[0 + j] + m[4 + j] + m[8 + j];
        t[1][j] = m[4 + j] - m[8 + j];
        t[2][j] = m[4 + j] + m[8 + j] - m[12 + j];
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        J.out[(size_t)(kh * 3 + 0) * (kC * kC) + i] = kr->scale * (t[kh][0] + t[kh][1] + t[kh][2]);
        J.out[(size_t)(kh * 3 + 1) * (kC * kC) + i] = kr->scale * (t[kh][1] - t[kh][2]);
        J.out[(size_t)(kh * 3 + 2) * (kC * kC) + i] = kr->scale * (t[kh][1] + t[kh][2] - t[kh][3]);
    }
}

bool wgw_eligible(const WgradArgs& a) {
    return tune_int("FS_WGW", 1) && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.Cin == kC && a.Cout == kC && a.pad_t == 0 && a.pad_l == 0 &&
           a.Ho == a.H - 2 && a.Wo == a.W - 2 && a.src_mode == SRC_PLAIN && !a.per_sample && !a.dy_unshuffle && !a.dy_a && a.dil_x <= 1 &&
           (double)a.N * a.H * a.W * kC * 4.0 < 2147483648.0;
}

// plan n problems of one launch; returns the slab scratch in floats (0: not eligible / too small to pay)
size_t wgw_plan(const WgradArgs* probs, int n, WgwArgs* out) {
    if (n < 1 || n > kW2MaxProb) return 0;
    WgwArgs w{};
    w.nprob = n;
    bool have_ab = false;   // every problem WITH an on-load affine shares (in_nstride, in_relu); the others ignore them
    for (int i = 0; i < n; ++i)
        if (probs[i].in_a) {
            if (!probs[i].in_b) return 0;
            if (have_ab && (probs[i].in_nstride != w.in_nstride || probs[i].in_relu != w.in_relu)) return 0;
            have_ab = true;
            w.in_nstride = probs[i].in_nstride;
            w.in_relu = probs[i].in_relu;
        }
    const int wgs_total = tune_int("FS_WGW_WGS", 256);
    const int per_lo = wgs_total / n < 1 ? 1 : wgs_total / n, extra = wgs_total >= n ? wgs_total - per_lo * n : 0;   // the first `extra` problems get one more
    long total_steps = 0;
    int wg = 0;
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        const WgradArgs& a = probs[i];
        if (!wgw_eligible(a)) return 0;
        WgwProb& p = w.prob[i];
        p.x = a.x;
        p.dy = a.dy;
        p.in_a = a.in_a;
        p.in_b = a.in_b;
        p.N = a.N;
        p.H = a.H;
        p.W = a.W;
        p.Ho = a.Ho;
        p.Wo = a.Wo;
        p.Ty = cdiv(a.Ho, 2);
        p.Tx = cdiv(a.Wo, 2);
        p.sps = cdiv(p.Ty * p.Tx, kST);
        p.steps = a.N * p.sps;
        const int per = per_lo + (i < extra ? 1 : 0);
        p.wg_count = p.steps < per ? p.steps : per;
        p.wg_begin = wg;
        p.slab_off = off;
        wg += p.wg_count;
        off += (size_t)p.wg_count * (16 * kC * kC);
        total_steps += p.steps;
    }
    if (total_steps < (long)tune_int("FS_WGW_MIN_STEPS", 64) * n) return 0;   // tiny problems: the slab traffic outweighs the matrix work
    w.n_wg = wg;
    *out = w;
    return off;
}

int wgw_run(const WgwArgs& planned, float* slabs, float* const* dw, float scale, hipStream_t s) {
    WgwArgs w = planned;
    w.slabs = slabs;
    Profiler* prof = Profiler::current();
    if (prof) {
        double fl = 0;
        for (int i = 0; i < w.nprob; ++i) fl += 2.0 * w.prob[i].N * w.prob[i].Ty * w.prob[i].Tx * 16.0 * kC * kC;   // executed: 16 products per 2x2 outputs
        prof->begin(PF_WGW, fl, s);
    }
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(wgw_kernel));
    hipLaunchKernelGGL(wgw_kernel, dim3((unsigned)w.n_wg), dim3(256), (size_t)(2 * kVF * 4), s, w);
    WgwReduce r{};
    r.scale = scale;
    r.n = w.nprob;
    for (int i = 0; i < w.nprob; ++i) {
        r.job[i].slabs = slabs + w.prob[i].slab_off;
        r.job[i].out = dw[i];
        r.job[i].n_slabs = w.prob[i].wg_count;
    }
    hipLaunchKernelGGL(wgw_reduce_kernel, dim3((unsigned)(kC * kC / 64), (unsigned)w.nprob), dim3(256), 0, s, r);
    if (prof) prof->end(s);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
