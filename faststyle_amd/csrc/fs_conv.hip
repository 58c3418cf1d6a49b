// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32: exact fp32, bit-identical to an fmaf chain).
//
// One kernel family serves every dense contraction with a sliding window on the path:
//   * transform-net convs (reference im_transf_net.py:91-119): 9x9/3x3, stride 1/2, SAME /
//     VALID / REFLECT-40 fused, instance-norm+ReLU of the producer applied on load
//     (im_transf_net.py:218-247), per-tile instance-norm statistics in the epilogue;
//   * the phase-collapsed resize-conv (im_transf_net.py:122-155) as a 2x2-tap conv with a
//     pixel-shuffle store;
//   * VGG16 3x3 convs + bias + ReLU (libs/vgg16.py:45-173);
//   * every dgrad (the same kernel on flipped/transposed filters; stride-2 dgrad through the
//     zero-dilated virtual input) and the Gram backward dF = F*S as a 1x1 conv with
//     per-sample filters (utils.py:78-81 adjoint).
//
// Mapping (wave64, 4 waves per workgroup): the GEMM M dimension is a TH x TW tile of output
// pixels flattened row-major (256 pixels per workgroup, 32 or 16 consecutive pixels per MFMA
// tile), N is a block of 64/32/16 output channels, K runs over (tap, input channel).  The
// input patch of the tile (with halo) is staged ONCE per channel chunk into LDS as
// [pixel][CC+1] (odd pitch: the 32 lanes of an A-fragment read hit 32 distinct banks) and
// re-used by all KHxKW taps; filters are staged as [k][BN] (B-fragment reads are contiguous).
#include "fs_kernels.h"

#include <cstdio>
#include <cstdlib>

namespace fs {

__device__ __forceinline__ bool src_coord(int mode, int refl, int v, int n_src, int& s) {
    if (mode == SRC_PLAIN) {
        s = v;
        return v >= 0 && v < n_src;
    } else if (mode == SRC_REFLECT) {
        if (v < 0 || v >= n_src + 2 * refl) return false;
        s = v - refl;
        if (s < 0) s = -s;
        if (s >= n_src) s = 2 * (n_src - 1) - s;
        return true;
    } else if (mode == SRC_DILATE2) {
        if (v < 0 || (v & 1)) return false;
        s = v >> 1;
        return s < n_src;
    } else {  // SRC_UP4
        if (v < 0) return false;
        s = v >> 2;
        return s < n_src;
    }
}

template <int MT>
struct Frag;
template <>
struct Frag<32> {
    typedef f32x16 acc_t;
    static constexpr int NACC = 16, KSTEP = 2;
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // accumulator register r of lane l holds row(r,l) of the 32x32 tile, column l&31
    __device__ static __forceinline__ int row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};
template <>
struct Frag<16> {
    typedef f32x4 acc_t;
    static constexpr int NACC = 4, KSTEP = 4;
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ int row(int r, int lane) { return 4 * (lane >> 4) + r; }
};

template <int MT, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    typedef Frag<MT> F;
    typedef typename F::acc_t acc_t;
    constexpr int KSTEP = F::KSTEP, NACC = F::NACC;
    constexpr int BN = WN * MT;
    HIP_DYNAMIC_SHARED(float, smem)
    const ConvPlan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = p.tiles_y * p.tiles_x;
    const int n = blockIdx.x / tiles;
    const int tr = blockIdx.x % tiles;
    const int ty0 = (tr / p.tiles_x) * p.TH, tx0 = (tr % p.tiles_x) * p.TW;
    const int co0 = blockIdx.y * BN;
    const int S = p.S, PW = p.PW, PH = p.PH, LG = p.LG, CC = p.CC;
    const int patch_floats = (PH * PW * S + 8 + 3) & ~3;
    float* patch = smem;
    float* wl = smem + patch_floats;

    const int lm = lane & (MT - 1), kq = lane / MT;
    const int tile_px = p.TH * p.TW;
    int laneA[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        int t = (wave * WM + m) * MT + lm;
        if (t >= tile_px) t = 0;
        const int py = t / p.TW, px = t - py * p.TW;
        laneA[m] = (py * a.stride * PW + px * a.stride) * S + kq;
    }
    const int laneB = kq * BN + lm;

    acc_t acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int nn = 0; nn < WN; ++nn)
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[m][nn][r] = 0.f;

    const int G = p.flat ? a.KH : a.KH * a.KW;
    const int nchunks = p.flat ? 1 : a.Cin / CC;
    const float* wbase = a.w + (size_t)n * a.w_nstride;
    const float* xn = a.x + (size_t)n * a.H * a.W * a.Cin;
    const int vy0 = ty0 * a.stride - a.pad_t, vx0 = tx0 * a.stride - a.pad_l;
    const bool has_ab = a.in_a != nullptr;
    const float* ia = has_ab ? a.in_a + (size_t)n * a.in_nstride : nullptr;
    const float* ib = has_ab ? a.in_b + (size_t)n * a.in_nstride : nullptr;

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int ci0 = chunk * CC;
        if (chunk) __syncthreads();
        // ---- stage the input patch (virtual image -> LDS [pixel][S]) ----
        if (p.flat) {
            for (int pix = tid; pix < PH * PW; pix += 256) {
                const int py = pix / PW, px = pix - py * PW;
                int sy, sx;
                const bool ok = src_coord(a.src_mode, a.refl, vy0 + py, a.H, sy) &&
                                src_coord(a.src_mode, a.refl, vx0 + px, a.W, sx);
                float v[3] = {0.f, 0.f, 0.f};
                if (ok) {
                    const float* src = xn + ((size_t)sy * a.W + sx) * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float t = src[c];
                        if (has_ab) t = fmaf(t, ia[c], ib[c]);
                        if (a.in_relu) t = fmaxf(t, 0.f);
                        v[c] = t;
                    }
                }
                patch[pix * 3 + 0] = v[0];
                patch[pix * 3 + 1] = v[1];
                patch[pix * 3 + 2] = v[2];
            }
        } else {
            const int c4n = CC >> 2;
            for (int e = tid; e < PH * PW * c4n; e += 256) {
                const int pix = e / c4n, c4 = e - pix * c4n;
                const int py = pix / PW, px = pix - py * PW;
                int sy, sx;
                const bool ok = src_coord(a.src_mode, a.refl, vy0 + py, a.H, sy) &&
                                src_coord(a.src_mode, a.refl, vx0 + px, a.W, sx);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const int c = ci0 + c4 * 4;
                    v = *reinterpret_cast<const float4*>(xn + ((size_t)sy * a.W + sx) * a.Cin + c);
                    if (has_ab) {
                        const float4 va = *reinterpret_cast<const float4*>(ia + c);
                        const float4 vb = *reinterpret_cast<const float4*>(ib + c);
                        v.x = fmaf(v.x, va.x, vb.x);
                        v.y = fmaf(v.y, va.y, vb.y);
                        v.z = fmaf(v.z, va.z, vb.z);
                        v.w = fmaf(v.w, va.w, vb.w);
                    }
                    if (a.in_relu) {
                        v.x = fmaxf(v.x, 0.f);
                        v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f);
                        v.w = fmaxf(v.w, 0.f);
                    }
                }
                float* d = patch + pix * S + c4 * 4;
                d[0] = v.x;
                d[1] = v.y;
                d[2] = v.z;
                d[3] = v.w;
            }
        }
        if (tid < 8) patch[PH * PW * S + tid] = 0.f;  // slack read by the zero-weight k padding
        // ---- stage the filter slice [G*LG][BN] ----
        {
            const int j4n = BN >> 2;
            const int rows = G * LG;
            const bool vec = (a.Cout & 3) == 0;
            for (int e = tid; e < rows * j4n; e += 256) {
                const int k = e / j4n, j4 = e - k * j4n;
                const int g = k / LG, r = k - g * LG;
                const int co = co0 + j4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                bool rowok;
                size_t grow;
                if (p.flat) {
                    rowok = r < a.KW * a.Cin;
                    grow = (size_t)g * a.KW * a.Cin + r;
                } else {
                    rowok = true;
                    grow = (size_t)g * a.Cin + ci0 + r;
                }
                if (rowok) {
                    const float* src = wbase + grow * a.Cout + co;
                    if (vec && co + 3 < a.Cout) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (co + 0 < a.Cout) v.x = src[0];
                        if (co + 1 < a.Cout) v.y = src[1];
                        if (co + 2 < a.Cout) v.z = src[2];
                        if (co + 3 < a.Cout) v.w = src[3];
                    }
                }
                *reinterpret_cast<float4*>(wl + k * BN + j4 * 4) = v;
            }
        }
        __syncthreads();
        // ---- MFMA sweep over taps x channels of this chunk ----
        for (int g = 0; g < G; ++g) {
            const int aoff = p.flat ? g * PW * S : ((g / a.KW) * PW + (g % a.KW)) * S;
            const float* pa = patch + aoff;
            const float* pb = wl + g * LG * BN + laneB;
#pragma unroll 4
            for (int kk = 0; kk < LG; kk += KSTEP) {
                float av[WM], bv[WN];
#pragma unroll
                for (int m = 0; m < WM; ++m) av[m] = pa[laneA[m] + kk];
#pragma unroll
                for (int nn = 0; nn < WN; ++nn) bv[nn] = pb[kk * BN + nn * MT];
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int nn = 0; nn < WN; ++nn) acc[m][nn] = F::mma(av[m], bv[nn], acc[m][nn]);
            }
        }
    }

    // ---- epilogue ----
    const int th_valid = min(p.TH, a.Ho - ty0), tw_valid = min(p.TW, a.Wo - tx0);
    // validity + coordinates of the rows this lane holds
    // (row index within the workgroup tile: t = (wave*WM+m)*MT + F::row(r,lane))
    if (a.stats) {
        float s1[WN];
#pragma unroll
        for (int nn = 0; nn < WN; ++nn) s1[nn] = 0.f;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                const int t = (wave * WM + m) * MT + F::row(r, lane);
                const int py = t / p.TW, px = t - py * p.TW;
                const bool ok = t < tile_px && py < th_valid && px < tw_valid;
#pragma unroll
                for (int nn = 0; nn < WN; ++nn) s1[nn] += ok ? acc[m][nn][r] : 0.f;
            }
#pragma unroll
        for (int nn = 0; nn < WN; ++nn) {
            s1[nn] += __shfl_xor(s1[nn], 32);
            if (MT == 16) s1[nn] += __shfl_xor(s1[nn], 16);
        }
        __syncthreads();
        float* red = smem;          // [4][BN]
        float* meanl = smem + 4 * BN;  // [BN]
        if (lane < MT)
#pragma unroll
            for (int nn = 0; nn < WN; ++nn) red[wave * BN + nn * MT + lane] = s1[nn];
        __syncthreads();
        const float cnt = (float)(th_valid * tw_valid);
        if (tid < BN) meanl[tid] = (red[tid] + red[BN + tid] + red[2 * BN + tid] + red[3 * BN + tid]) / cnt;
        __syncthreads();
        float mu[WN], s2[WN];
#pragma unroll
        for (int nn = 0; nn < WN; ++nn) {
            mu[nn] = meanl[nn * MT + lm];
            s2[nn] = 0.f;
        }
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                const int t = (wave * WM + m) * MT + F::row(r, lane);
                const int py = t / p.TW, px = t - py * p.TW;
                const bool ok = t < tile_px && py < th_valid && px < tw_valid;
#pragma unroll
                for (int nn = 0; nn < WN; ++nn) {
                    const float d = acc[m][nn][r] - mu[nn];
                    s2[nn] += ok ? d * d : 0.f;
                }
            }
#pragma unroll
        for (int nn = 0; nn < WN; ++nn) {
            s2[nn] += __shfl_xor(s2[nn], 32);
            if (MT == 16) s2[nn] += __shfl_xor(s2[nn], 16);
        }
        if (lane < MT)
#pragma unroll
            for (int nn = 0; nn < WN; ++nn) red[wave * BN + nn * MT + lane] = s2[nn];
        __syncthreads();
        if (tid < BN && co0 + tid < a.Cout) {
            float* st = a.stats + ((size_t)blockIdx.x * a.Cout + co0 + tid) * 3;
            st[0] = meanl[tid];
            st[1] = red[tid] + red[BN + tid] + red[2 * BN + tid] + red[3 * BN + tid];
            st[2] = cnt;
        }
    }

    const int Cr = a.shuffle ? a.Cout >> 2 : a.Cout;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            const int t = (wave * WM + m) * MT + F::row(r, lane);
            const int py = t / p.TW, px = t - py * p.TW;
            if (!(t < tile_px && py < th_valid && px < tw_valid)) continue;
            const int oy = ty0 + py, ox = tx0 + px;
#pragma unroll
            for (int nn = 0; nn < WN; ++nn) {
                const int co = co0 + nn * MT + lm;
                if (co >= a.Cout) continue;
                float v = acc[m][nn][r];
                if (a.bias) v += a.bias[co];
                if (a.out_relu) v = fmaxf(v, 0.f);
                if (a.add_src) {
                    const int ap = a.add_pad;
                    if (oy >= ap && oy < a.Ho - ap && ox >= ap && ox < a.Wo - ap)
                        v += a.add_src[(((size_t)n * (a.Ho - 2 * ap) + (oy - ap)) * (a.Wo - 2 * ap) + (ox - ap)) * a.Cout + co];
                }
                size_t o;
                if (a.shuffle) {
                    const int q = co / Cr, cr = co - q * Cr;
                    o = (((size_t)n * 2 * a.Ho + 2 * oy + (q >> 1)) * (2 * a.Wo) + 2 * ox + (q & 1)) * Cr + cr;
                } else {
                    o = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout + co;
                }
                a.y[o] = v;
            }
        }
}

// -------------------------------------------------------------------------------------- host
Profiler*& Profiler::current() {
    static thread_local Profiler* p = nullptr;
    return p;
}
void Profiler::begin(int fam, double flops, hipStream_t s) {
    if (n == cap) {
        const int ncap = cap ? cap * 2 : 1024;
        Rec* nr = (Rec*)realloc(recs, sizeof(Rec) * ncap);
        if (!nr) return;
        recs = nr;
        for (int i = cap; i < ncap; ++i) {
            (void)hipEventCreate(&recs[i].a);
            (void)hipEventCreate(&recs[i].b);
        }
        cap = ncap;
    }
    recs[n].fam = fam;
    recs[n].flops = flops;
    (void)hipEventRecord(recs[n].a, s);
}
void Profiler::end(hipStream_t s) {
    if (n < cap) (void)hipEventRecord(recs[n++].b, s);
}
int Profiler::collect(double out[kFamilies][3]) {
    for (int f = 0; f < kFamilies; ++f) out[f][0] = out[f][1] = out[f][2] = 0;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(recs[i].b) != hipSuccess) return -1;
        float ms = 0;
        if (hipEventElapsedTime(&ms, recs[i].a, recs[i].b) != hipSuccess) return -1;
        out[recs[i].fam][0] += 1;
        out[recs[i].fam][1] += recs[i].flops;
        out[recs[i].fam][2] += ms;
    }
    return 0;
}
void Profiler::reset() { n = 0; }
Profiler::~Profiler() {
    for (int i = 0; i < cap; ++i) {
        (void)hipEventDestroy(recs[i].a);
        (void)hipEventDestroy(recs[i].b);
    }
    free(recs);
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

static void plan_tile(int Ho, int Wo, int KH, int KW, int stride, int max_px, int* TH, int* TW) {
    double best = -1;
    int bth = 1, btw = 1;
    for (int tw = 1; tw <= Wo && tw <= max_px; ++tw) {
        // only widths that split Wo evenly-ish: tw = ceil(Wo / k)
        const int k = cdiv(Wo, tw);
        if (tw != cdiv(Wo, k)) continue;
        int th = max_px / tw;
        if (th > Ho) th = Ho;
        if (th < 1) continue;
        th = cdiv(Ho, cdiv(Ho, th));  // shrink to the even split
        const double eff = (double)Ho * Wo / ((double)cdiv(Ho, th) * cdiv(Wo, tw) * max_px);
        const double halo = (double)((th - 1) * stride + KH) * ((tw - 1) * stride + KW) / ((double)th * tw * stride * stride);
        const double score = eff / (1.0 + 0.1 * (halo - 1.0));
        if (score > best) {
            best = score;
            bth = th;
            btw = tw;
        }
    }
    *TH = bth;
    *TW = btw;
}

// variant -> (MFMA tile, m-tiles per wave, n-tiles per wave); pixels per workgroup = 4*WM*MT
static const int kVarMT[5] = {32, 32, 16, 32, 32};
static const int kVarWM[5] = {2, 2, 4, 1, 1};
static const int kVarWN[5] = {2, 1, 1, 2, 1};

static void plan_variant(const ConvArgs& a, int variant, ConvPlan* out) {
    ConvPlan p{};
    p.flat = a.Cin == 3;
    p.variant = variant;
    p.BN = kVarMT[variant] * kVarWN[variant];
    const int max_px = 4 * kVarWM[variant] * kVarMT[variant];
    const int kstep = kVarMT[variant] == 16 ? 4 : 2;
    plan_tile(a.Ho, a.Wo, a.KH, a.KW, a.stride, max_px, &p.TH, &p.TW);
    p.tiles_y = cdiv(a.Ho, p.TH);
    p.tiles_x = cdiv(a.Wo, p.TW);
    p.PH = (p.TH - 1) * a.stride + a.KH;
    p.PW = (p.TW - 1) * a.stride + a.KW;
    const int budget = env_int("FS_CONV_LDS_KB", 40) * 1024;
    if (p.flat) {
        p.CC = 3;
        p.S = 3;
        p.LG = cdiv(a.KW * 3, kstep) * kstep;
        const int G = a.KH;
        p.lds_bytes = 4 * (((p.PH * p.PW * p.S + 8 + 3) & ~3) + G * p.LG * p.BN);
    } else {
        const int G = a.KH * a.KW;
        int forced = env_int("FS_CONV_CC", 0);
        int chosen = 0, chosen_bytes = 0;
        for (int cc = 32; cc >= 4; cc >>= 1) {
            if (a.Cin % cc) continue;
            const int bytes = 4 * (((p.PH * p.PW * (cc + 1) + 8 + 3) & ~3) + G * cc * p.BN);
            if (forced == cc || (!forced && bytes <= budget) || cc == 4) {
                chosen = cc;
                chosen_bytes = bytes;
                break;
            }
        }
        p.CC = chosen;
        p.S = chosen + 1;
        p.LG = chosen;
        p.lds_bytes = chosen_bytes;
    }
    if (p.lds_bytes < 4 * 5 * p.BN) p.lds_bytes = 4 * 5 * p.BN;  // stats scratch
    *out = p;
}

ConvPlan conv_plan(const ConvArgs& a) {
    ConvPlan p;
    if (a.Cout <= 16) {
        plan_variant(a, 2, &p);
        return p;
    }
    // widest tile first; halve the workgroup tile while the launch cannot fill the chip
    // (256 CUs x >= 2 workgroups), e.g. VGG conv4_x at batch 4 or the 64-channel residual convs
    const int min_wgs = env_int("FS_CONV_MIN_WGS", 512);
    const int order_wide[3] = {0, 3, 4}, order_narrow[2] = {1, 4};
    const int* order = a.Cout > 32 ? order_wide : order_narrow;
    const int n = a.Cout > 32 ? 3 : 2;
    for (int i = 0; i < n; ++i) {
        plan_variant(a, order[i], &p);
        const long wgs = (long)a.N * p.tiles_y * p.tiles_x * cdiv(a.Cout, p.BN);
        if (wgs >= min_wgs) break;
    }
    return p;
}

int conv_launch(const ConvArgs& a, hipStream_t s) {
    const ConvPlan& p = a.p;
    if (p.CC <= 0 || (!p.flat && (a.Cin % 4 || a.Cin % p.CC))) return -1;
    if (p.lds_bytes > 64 * 1024) return -2;
    dim3 grid((unsigned)(a.N * p.tiles_y * p.tiles_x), (unsigned)cdiv(a.Cout, p.BN));
    Profiler* prof = Profiler::current();
    if (prof) {
        // algorithmic FLOPs: 2*M*K*N with the true extents; a zero-dilated dgrad only does 1/4 useful work
        double fl = 2.0 * a.N * a.Ho * a.Wo * (double)a.KH * a.KW * a.Cin * a.Cout;
        if (a.src_mode == SRC_DILATE2) fl *= 0.25;
        prof->begin(p.variant < 3 ? p.variant : p.variant + 1, fl, s);
    }
    if (p.variant == 0)
        hipLaunchKernelGGL((conv_igemm_kernel<32, 2, 2>), grid, dim3(256), (size_t)p.lds_bytes, s, a);
    else if (p.variant == 1)
        hipLaunchKernelGGL((conv_igemm_kernel<32, 2, 1>), grid, dim3(256), (size_t)p.lds_bytes, s, a);
    else if (p.variant == 2)
        hipLaunchKernelGGL((conv_igemm_kernel<16, 4, 1>), grid, dim3(256), (size_t)p.lds_bytes, s, a);
    else if (p.variant == 3)
        hipLaunchKernelGGL((conv_igemm_kernel<32, 1, 2>), grid, dim3(256), (size_t)p.lds_bytes, s, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<32, 1, 1>), grid, dim3(256), (size_t)p.lds_bytes, s, a);
    if (prof) prof->end(s);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
