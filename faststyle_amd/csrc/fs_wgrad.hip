// Pixel-reduction contractions on the fp32 matrix cores:
//   * filter gradients of the transform-net convs,  dW[k,co] = sum_p X[p+tap(k), ci(k)] * dY[p,co]
//     (the adjoint of tf.nn.conv2d wrt its filter, reference im_transf_net.py:115 / train.py:203);
//   * the per-sample Gram matrices  G = F^T F  (reference utils.py:76-82) as the 1x1 case with
//     X == dY == F and one result per sample.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32): the instruction's K dimension is a PAIR OF PIXELS, its
// M dimension 32 consecutive k = (tap, ci) and its N dimension 32 output channels.  A wave owns
// KWV k-blocks x (4/KWV) co-blocks of the result (64 accumulator registers); KWV is 1 for wide
// outputs (Gram, 128+ channels), 2 for 64 channels, 4 for <= 32 channels, so the four waves of a
// workgroup always cover 16 MFMA tiles per staged pixel tile.  A workgroup walks a strided list
// of pixel tiles, stages each tile's input patch as [pixel][C+1] and its dY tile as [pixel][DP] in
// LDS, and finally writes ONE partial slab; fs::reduce_slabs sums the slabs in a fixed order
// (deterministic, no atomics).
#include "fs_kernels.h"

#include <cstdlib>
#include <type_traits>

namespace fs {

__device__ __forceinline__ bool wsrc_coord(int mode, int refl, int v, int n_src, int& s) {
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
    } else {
        if (v < 0) return false;
        s = v >> 2;
        return s < n_src;
    }
}

#ifdef FS_CONV_TRACE
// debug build only (tools/conv_trace.py): per-workgroup phase cycle counts of the last launch
__device__ long long g_wgrad_trace[4096 * 8];
#define FS_WTRACE_NOW() ((long long)__builtin_readcyclecounter())
extern "C" int fs_debug_wgrad_trace(long long* out, int n_wg) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgrad_trace), sizeof(long long) * 8 * (size_t)n_wg, 0, hipMemcpyDeviceToHost);
}
extern "C" int fs_debug_wgrad_trace_reset() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wgrad_trace)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(long long) * 8 * 4096);
}
#endif

template <int KWV, int NWV>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
#ifdef FS_CONV_TRACE
    const long long tr_t0 = FS_WTRACE_NOW();
    long long tr_stage = 0, tr_sweep = 0, tr_bar = 0;
#endif
    const WgradPlan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 31, kq = lane >> 5;
    const int cogroups = cdiv(p.NB, NWV);
    const int kbg = blockIdx.y / cogroups, nbg = blockIdx.y % cogroups;
    // waves_k waves tile the k dimension; when K is short (Gram of 64 channels, 9x9x3 filters) the
    // remaining 4/waves_k waves split the tile's pixel rows and write their own partial slabs
    const int waves_k = p.waves_k, wsplit = 4 / waves_k;
    const int wk = wave % waves_k, ws = wave / waves_k;
    const int kb0 = (kbg * waves_k + wk) * KWV;       // this wave's first k-block
    const int nbw = min(NWV, p.NB - nbg * NWV);       // co-blocks of this workgroup
    const int DP = nbw * 32;                          // dY LDS pitch
    const int co_g0 = nbg * NWV * 32;
    // staged input-channel window
    const int CS = p.S - 1;
    const int cA = a.Cin <= 128 ? 0 : (kbg * 128) % a.Cin;
    const int S = p.S, PW = p.PW, PH = p.PH;
    const int patch_floats = (PH * PW * S + 4 + 3) & ~3;
    float* patch = smem;
    float* dyl = smem + patch_floats;

    // A-operand base of this lane per owned k-block: k -> (tap, ci)
    int abase[KWV], amul[KWV];
#pragma unroll
    for (int q = 0; q < KWV; ++q) {
        const int k = (kb0 + q) * 32 + lm;
        abase[q] = PH * PW * S;  // zero slack
        amul[q] = 0;
        if (kb0 + q < p.KB && k < p.K) {
            const int tap = k / a.Cin, ci = k - tap * a.Cin;
            const int kh = tap / a.KW, kw = tap - kh * a.KW;
            abase[q] = (kh * PW + kw * a.dil_x) * S + (ci - cA);
            amul[q] = 1;
        }
    }

    f32x16 acc[KWV][NWV];
#pragma unroll
    for (int q = 0; q < KWV; ++q)
#pragma unroll
        for (int j = 0; j < NWV; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][j][r] = 0.f;

    const int tiles = p.tiles_y * p.tiles_x;
    const int nsel = a.per_sample ? 1 : a.N;
    const int total = nsel * tiles;
    const bool has_ab = a.in_a != nullptr;
    const bool xvec = (a.Cin & 3) == 0;
    const int Cr = a.dy_unshuffle ? a.Cout >> 2 : a.Cout;

    // Tile loop.  Staging is BATCHED: 8 independent 16-byte global loads per thread are issued
    // back to back, then transformed and written to LDS -- a plain "load, use" loop serialises on
    // the ~1-2 us global latency per iteration (the vector paths below; the scalar paths only serve
    // the 3-channel first layer and ragged channel counts).
    constexpr int XB = 8;
    const int c4n = CS >> 2;
    const int ne_x = PH * PW * c4n;
    const int j4n = DP >> 2;
    const int ne_d = p.TH * p.TW * j4n;
    const bool dvec = (Cr & 3) == 0 && co_g0 + DP <= a.Cout;  // (the scalar dY path has no on-load transform)

    // x / d through a float reciprocal (exact for x < 2^22): the staging loops decompose ~30 element indices per
    // thread and tile, and an integer division costs ~35 instructions
    auto fdiv = [](int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); };
    const float inv_c4n = 1.0f / (float)(c4n > 0 ? c4n : 1), inv_pw = 1.0f / (float)PW, inv_j4n = 1.0f / (float)j4n;
    const float inv_tw = 1.0f / (float)p.TW, inv_cs = 1.0f / (float)CS, inv_cr = 1.0f / (float)Cr;
    const float inv_tiles = 1.0f / (float)tiles, inv_tx = 1.0f / (float)p.tiles_x;

    // fast staging paths (power-of-two channel-quad counts: every layer of this network)
    constexpr unsigned kOOB = 0x80000000u;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    const bool fastx = xvec && pow2(c4n) && c4n <= 256;
    const bool fastd = dvec && pow2(j4n) && j4n <= 256;
    auto ilog2 = [](int v) {
        int sh = 0;
        while ((1 << sh) < v) ++sh;
        return sh;
    };
    const int c4sh = ilog2(c4n > 0 ? c4n : 1), j4sh = ilog2(j4n);
    const int ppi = 256 >> c4sh, dpy = ppi / PW, dpx = ppi - dpy * PW;            // pixel step of the x elements
    const int ppi_d = 256 >> j4sh, dpy_d = ppi_d / p.TW, dpx_d = ppi_d - dpy_d * p.TW;
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * a.Cin) * 4u);
    const unsigned d_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * a.Cout) * 4u);
    auto uniform_ptr = [](const float* ptr) {  // a wave-uniform pointer, pinned to scalar registers
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };
    auto stage = [&](int t) {
        // (tile coordinates are wave-uniform but come out of the vector ALU: readfirstlane moves them, and the base
        // pointers derived from them, to scalar registers)
        const int tq = __builtin_amdgcn_readfirstlane(fdiv(t, inv_tiles));
        const int n = a.per_sample ? (int)blockIdx.z : tq;
        const int tr = t - tq * tiles;
        const int tyi = __builtin_amdgcn_readfirstlane(fdiv(tr, inv_tx));
        const int ty0 = tyi * p.TH, tx0 = (tr - tyi * p.tiles_x) * p.TW;
        const int vy0 = ty0 * a.stride - a.pad_t, vx0 = tx0 * a.stride - a.pad_l;
        const float* xn = uniform_ptr(a.x + (size_t)n * a.H * a.W * a.Cin);
        const float* ia = has_ab ? uniform_ptr(a.in_a + (size_t)n * a.in_nstride) : nullptr;
        const float* ib = has_ab ? uniform_ptr(a.in_b + (size_t)n * a.in_nstride) : nullptr;
        if (fastx) {
            // Thread <-> (pixel slot, channel quad) with the channel quad FIXED per thread (c4n is a power of two):
            // the pixel advances by 256/c4n per element, so (py, px) and the LDS position move by additions, the
            // instance-norm parameters of the thread's four channels are read once per tile, and the global reads
            // are buffer loads whose out-of-image / out-of-patch offset kOOB returns zeros (no branches at all).
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(xn), 0, x_bytes, 0x00020000);
            const int c4 = tid & (c4n - 1);
            const int ch = cA + c4 * 4;
            float4 va = make_float4(1.f, 1.f, 1.f, 1.f);
            uint4 vb = make_uint4(0u, 0u, 0u, 0u);
            if (has_ab) {
                va = *reinterpret_cast<const float4*>(ia + ch);
                vb = *reinterpret_cast<const uint4*>(ib + ch);
            }
            const int npix = PH * PW;
            int pix_i = tid >> c4sh;                                   // issue-side counters
            int py = fdiv(pix_i, inv_pw), px = pix_i - py * PW;
            int pix_c = pix_i, dst = pix_i * S + c4 * 4;              // commit-side counters
            for (int e0 = 0; e0 < ne_x; e0 += XB * 256) {
                float4 xv[XB];
                unsigned valid = 0;
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    int sy, sx;
                    const bool ok = pix_i < npix && wsrc_coord(a.src_mode, a.refl, vy0 + py, a.H, sy) &&
                                    wsrc_coord(a.src_mode, a.refl, vx0 + px, a.W, sx);
                    const unsigned vo = ok ? (unsigned)((sy * a.W + sx) * a.Cin + ch) * 4u : kOOB;
                    xv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, vo, 0, 0));
                    valid |= ok ? 1u << i : 0u;
                    pix_i += ppi;
                    px += dpx;
                    py += dpy;
                    if (px >= PW) {
                        px -= PW;
                        ++py;
                    }
                }
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    float4 v = xv[i];
                    if (has_ab) {  // padding arrives as 0 and stays 0: the shift is cleared by a bit mask
                        const unsigned okm = (valid >> i) & 1u ? 0xFFFFFFFFu : 0u;
                        v.x = fmaf(v.x, va.x, __uint_as_float(vb.x & okm));
                        v.y = fmaf(v.y, va.y, __uint_as_float(vb.y & okm));
                        v.z = fmaf(v.z, va.z, __uint_as_float(vb.z & okm));
                        v.w = fmaf(v.w, va.w, __uint_as_float(vb.w & okm));
                    }
                    if (a.in_relu) {
                        v.x = fmaxf(v.x, 0.f);
                        v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f);
                        v.w = fmaxf(v.w, 0.f);
                    }
                    float* d = patch + (pix_c < npix ? dst : npix * S);  // beyond the patch: the zero slack
                    d[0] = v.x;
                    d[1] = v.y;
                    d[2] = v.z;
                    d[3] = v.w;
                    pix_c += ppi;
                    dst += ppi * S;
                }
            }
        } else if (xvec) {
            for (int e0 = tid; e0 < ne_x; e0 += XB * 256) {
                float4 xv[XB];
                unsigned valid = 0;
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    const int e = e0 + i * 256;
                    xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < ne_x) {
                        const int pix = fdiv(e, inv_c4n), c4 = e - pix * c4n;
                        const int py = fdiv(pix, inv_pw), px = pix - py * PW;
                        int sy, sx;
                        if (wsrc_coord(a.src_mode, a.refl, vy0 + py, a.H, sy) &&
                            wsrc_coord(a.src_mode, a.refl, vx0 + px, a.W, sx)) {
                            xv[i] = *reinterpret_cast<const float4*>(xn + ((size_t)sy * a.W + sx) * a.Cin + cA + c4 * 4);
                            valid |= 1u << i;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    const int e = e0 + i * 256;
                    if (e >= ne_x) continue;
                    const int pix = fdiv(e, inv_c4n), c4 = e - pix * c4n;
                    float4 v = xv[i];
                    if (valid & (1u << i)) {
                        const int c = cA + c4 * 4;
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
        } else {  // Cin == 3 (the first layer): scalar
            for (int e = tid; e < PH * PW * CS; e += 256) {
                const int pix = fdiv(e, inv_cs), c = e - pix * CS;
                const int py = fdiv(pix, inv_pw), px = pix - py * PW;
                int sy, sx;
                const bool ok = wsrc_coord(a.src_mode, a.refl, vy0 + py, a.H, sy) &&
                                wsrc_coord(a.src_mode, a.refl, vx0 + px, a.W, sx);
                float v = 0.f;
                if (ok) {
                    v = xn[((size_t)sy * a.W + sx) * a.Cin + cA + c];
                    if (has_ab) v = fmaf(v, ia[cA + c], ib[cA + c]);
                    if (a.in_relu) v = fmaxf(v, 0.f);
                }
                patch[pix * S + c] = v;
            }
        }
        if (tid < 4) patch[PH * PW * S + tid] = 0.f;
        if (a.same_xy) {
            // Gram matrix of <= 128 channels: dY IS the staged input tile (x == dy, 1x1, every channel in the patch) --
            // nothing more to stage; the B operand is read from the patch (half the HBM traffic of the 64/128-channel Grams,
            // which are bandwidth-bound: 537 MB of conv1_2 features per batch of 32)
        } else if (fastd) {
            // same scheme for the dY tile: the thread's channel quad (and with it the pixel-unshuffle phase and the
            // on-load affine of the deconv units) is fixed, pixels advance by 256/j4n
            const float* dyn = uniform_ptr(a.dy + (size_t)n * a.Ho * a.Wo * a.Cout);
            const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(dyn), 0, d_bytes, 0x00020000);
            const int j4 = tid & (j4n - 1);
            const int co = co_g0 + j4 * 4;
            int cbase = co, ystep = a.Wo * a.Cout, xstep = a.Cout;   // element offset = oy*ystep + ox*xstep + cbase
            if (a.dy_unshuffle) {
                const int q = fdiv(co, inv_cr), cr = co - q * Cr;
                cbase = ((q >> 1) * 2 * a.Wo + (q & 1)) * Cr + cr;
                ystep = 4 * a.Wo * Cr;
                xstep = 2 * Cr;
            }
            float4 va = make_float4(1.f, 1.f, 1.f, 1.f);
            uint4 vb = make_uint4(0u, 0u, 0u, 0u);
            if (a.dy_a) {
                va = *reinterpret_cast<const float4*>(a.dy_a + (size_t)n * a.dy_nstride + co);
                vb = *reinterpret_cast<const uint4*>(a.dy_b + (size_t)n * a.dy_nstride + co);
            }
            const int npix = p.TH * p.TW;
            int pix_i = tid >> j4sh;
            int py = fdiv(pix_i, inv_tw), px = pix_i - py * p.TW;
            int e_c = tid;
            for (int e0 = 0; e0 < ne_d; e0 += XB * 256) {
                float4 dv[XB];
                unsigned valid = 0;
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    const int oy = ty0 + py, ox = tx0 + px;
                    const bool ok = pix_i < npix && oy < a.Ho && ox < a.Wo;
                    const unsigned vo = ok ? (unsigned)(oy * ystep + ox * xstep + cbase) * 4u : kOOB;
                    dv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(dr, vo, 0, 0));
                    valid |= ok ? 1u << i : 0u;
                    pix_i += ppi_d;
                    px += dpx_d;
                    py += dpy_d;
                    if (px >= p.TW) {
                        px -= p.TW;
                        ++py;
                    }
                }
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    float4 v = dv[i];
                    if (a.dy_a) {
                        const unsigned okm = (valid >> i) & 1u ? 0xFFFFFFFFu : 0u;
                        v.x = fmaf(v.x, va.x, __uint_as_float(vb.x & okm));
                        v.y = fmaf(v.y, va.y, __uint_as_float(vb.y & okm));
                        v.z = fmaf(v.z, va.z, __uint_as_float(vb.z & okm));
                        v.w = fmaf(v.w, va.w, __uint_as_float(vb.w & okm));
                    }
                    if (a.dy_relu) {
                        v.x = fmaxf(v.x, 0.f);
                        v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f);
                        v.w = fmaxf(v.w, 0.f);
                    }
                    if (e_c < ne_d) *reinterpret_cast<float4*>(dyl + e_c * 4) = v;
                    e_c += 256;
                }
            }
        } else if (dvec) {
            for (int e0 = tid; e0 < ne_d; e0 += XB * 256) {
                float4 dv[XB];
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    const int e = e0 + i * 256;
                    dv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < ne_d) {
                        const int pix = fdiv(e, inv_j4n), j4 = e - pix * j4n;
                        const int py = fdiv(pix, inv_tw), px = pix - py * p.TW;
                        const int oy = ty0 + py, ox = tx0 + px;
                        const int co = co_g0 + j4 * 4;
                        if (oy < a.Ho && ox < a.Wo) {
                            const float* src;
                            if (a.dy_unshuffle) {
                                const int q = fdiv(co, inv_cr), cr = co - q * Cr;
                                src = a.dy + (((size_t)n * 2 * a.Ho + 2 * oy + (q >> 1)) * (2 * a.Wo) + 2 * ox + (q & 1)) * Cr + cr;
                            } else {
                                src = a.dy + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout + co;
                            }
                            float4 v = *reinterpret_cast<const float4*>(src);
                            if (a.dy_a) {
                                const float4 va = *reinterpret_cast<const float4*>(a.dy_a + (size_t)n * a.dy_nstride + co);
                                const float4 vb = *reinterpret_cast<const float4*>(a.dy_b + (size_t)n * a.dy_nstride + co);
                                v.x = fmaf(v.x, va.x, vb.x);
                                v.y = fmaf(v.y, va.y, vb.y);
                                v.z = fmaf(v.z, va.z, vb.z);
                                v.w = fmaf(v.w, va.w, vb.w);
                            }
                            if (a.dy_relu) {
                                v.x = fmaxf(v.x, 0.f);
                                v.y = fmaxf(v.y, 0.f);
                                v.z = fmaxf(v.z, 0.f);
                                v.w = fmaxf(v.w, 0.f);
                            }
                            dv[i] = v;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    const int e = e0 + i * 256;
                    if (e < ne_d) *reinterpret_cast<float4*>(dyl + e * 4) = dv[i];
                }
            }
        } else {  // ragged channel counts: scalar
            for (int e = tid; e < ne_d; e += 256) {
                const int pix = fdiv(e, inv_j4n), j4 = e - pix * j4n;
                const int py = fdiv(pix, inv_tw), px = pix - py * p.TW;
                const int oy = ty0 + py, ox = tx0 + px;
                const int co = co_g0 + j4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (oy < a.Ho && ox < a.Wo && co < a.Cout) {
                    const float* src;
                    if (a.dy_unshuffle) {
                        const int q = fdiv(co, inv_cr), cr = co - q * Cr;
                        src = a.dy + (((size_t)n * 2 * a.Ho + 2 * oy + (q >> 1)) * (2 * a.Wo) + 2 * ox + (q & 1)) * Cr + cr;
                    } else {
                        src = a.dy + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout + co;
                    }
                    v.x = src[0];
                    if (co + 1 < a.Cout) v.y = src[1];
                    if (co + 2 < a.Cout) v.z = src[2];
                    if (co + 3 < a.Cout) v.w = src[3];
                    if (a.dy_a || a.dy_relu) {
                        float* vv = reinterpret_cast<float*>(&v);
                        for (int k = 0; k < 4 && co + k < a.Cout; ++k) {
                            if (a.dy_a) vv[k] = fmaf(vv[k], a.dy_a[(size_t)n * a.dy_nstride + co + k], a.dy_b[(size_t)n * a.dy_nstride + co + k]);
                            if (a.dy_relu) vv[k] = fmaxf(vv[k], 0.f);
                        }
                    }
                }
                *reinterpret_cast<float4*>(dyl + e * 4) = v;
            }
        }
    };

    bool first = true;
    for (int t = blockIdx.x; t < total; t += p.n_wg) {
#ifdef FS_CONV_TRACE
        const long long q0 = FS_WTRACE_NOW();
#endif
        if (!first) __syncthreads();
        first = false;
#ifdef FS_CONV_TRACE
        const long long q1 = FS_WTRACE_NOW();
#endif
        stage(t);
#ifdef FS_CONV_TRACE
        const long long q2 = FS_WTRACE_NOW();
#endif
        __syncthreads();
#ifdef FS_CONV_TRACE
        const long long q3 = FS_WTRACE_NOW();
        tr_bar += (q1 - q0) + (q3 - q2);
        tr_stage += q2 - q1;
#endif
        // ---- MFMA sweep over pixel pairs ----
        // (the common case -- every co-block of the workgroup present -- is compiled WITHOUT the per-MFMA `j < nbw`
        // guard: with the guard each matrix instruction sits behind its own branch and s_waitcnt)
        if (kb0 < p.KB) {
            auto sweep = [&](auto FULL) {
                constexpr bool full = decltype(FULL)::value;
                for (int py = ws; py < p.TH; py += wsplit) {
                    const int rowA = py * a.stride * PW;
#pragma unroll 2
                    for (int px0 = 0; px0 < p.TW; px0 += 2) {
                        const int px = px0 + kq;
                        const int poff = (rowA + px * a.stride) * S;
                        float av[KWV], bv[NWV];
#pragma unroll
                        for (int q = 0; q < KWV; ++q) av[q] = patch[abase[q] + amul[q] * poff];
                        const float* pb = a.same_xy ? patch + (py * PW + px) * S + co_g0 + lm : dyl + (py * p.TW + px) * DP + lm;
#pragma unroll
                        for (int j = 0; j < NWV; ++j) bv[j] = (full || j < nbw) ? pb[j * 32] : 0.f;
#pragma unroll
                        for (int q = 0; q < KWV; ++q)
#pragma unroll
                            for (int j = 0; j < NWV; ++j)
                                if (full || j < nbw)
                                    acc[q][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[j], acc[q][j], 0, 0, 0);
                    }
                }
            };
            if (nbw == NWV)
                sweep(std::true_type{});
            else
                sweep(std::false_type{});
        }
#ifdef FS_CONV_TRACE
        tr_sweep += FS_WTRACE_NOW() - q3;
#endif
    }
#ifdef FS_CONV_TRACE
    const long long tr_main = FS_WTRACE_NOW();
#endif
    // ---- write this workgroup's partial slab ----
    float* slab = a.slabs + (((size_t)blockIdx.z * p.n_wg + blockIdx.x) * wsplit + ws) * (size_t)p.K * a.Cout;
    // everything this wave owns inside K x Cout (the usual case): no per-element predicate, 32-bit offsets
    const bool whole = (kb0 + KWV) * 32 <= p.K && co_g0 + nbw * 32 <= a.Cout && nbw == NWV;
    if (whole) {
        const int base = (kb0 * 32 + 4 * (lane >> 5)) * a.Cout + co_g0 + lm;
#pragma unroll
        for (int q = 0; q < KWV; ++q)
#pragma unroll
            for (int j = 0; j < NWV; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[base + (q * 32 + (r & 3) + 8 * (r >> 2)) * a.Cout + j * 32] = acc[q][j][r];
    } else {
#pragma unroll
        for (int q = 0; q < KWV; ++q) {
            if (kb0 + q >= p.KB) continue;
#pragma unroll
            for (int j = 0; j < NWV; ++j) {
                if (j >= nbw) continue;
                const int co = co_g0 + j * 32 + lm;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kk = (kb0 + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (kk < p.K && co < a.Cout) slab[(size_t)kk * a.Cout + co] = acc[q][j][r];
                }
            }
        }
    }
#ifdef FS_CONV_TRACE
    {
        const long long tr_end = FS_WTRACE_NOW();
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (tid == 0 && lin < 4096) {
            long long* t = g_wgrad_trace + lin * 8;
            t[0] = tr_t0;
            t[1] = 0;
            t[2] = tr_sweep;
            t[3] = tr_stage;
            t[4] = tr_bar;
            t[5] = tr_end - tr_main;
            t[6] = tr_end;
            t[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
#endif
}

static int env_int2(const char* name, int dflt) { return tune_int(name, dflt); }

WgradPlan wgrad_plan(const WgradArgs& a) {
    WgradPlan p{};
    p.K = a.KH * a.KW * a.Cin;
    p.KB = cdiv(p.K, 32);
    p.NB = cdiv(a.Cout, 32);
    // a wave owns KWV k-blocks x NWV co-blocks of MFMA tiles: <1,4> wide outputs / Gram, <4,1> narrow
    // outputs, <2,2> 64-channel outputs with short K, <5,2> the 3x3 64->64 filters (K = 576 = 18 k-blocks:
    // one workgroup covers all of K, so the tile is staged once instead of once per k-group)
    if (p.NB >= 4 || a.Cin > 128) {
        p.KWV = 1;
        p.NWV = 4;
    } else if (p.NB >= 2) {
        p.NWV = 2;
        p.KWV = p.KB > 8 ? 5 : 2;
    } else {
        p.KWV = 4;
        p.NWV = 1;
    }
    const int NWV = p.NWV;
    // pixel tile: the largest of 256/128/64/32 pixels (even width) that keeps the staged patch + dY
    // tile within ~80 KiB of LDS (two workgroups per CU) and a few batches of loads per thread
    const int CS = a.Cin <= 128 ? a.Cin : 128;
    const int DP = 32 * (p.NB < NWV ? p.NB : NWV);
    const int dil = a.dil_x > 0 ? a.dil_x : 1;
    const int kblocks_w0 = cdiv(p.KB, p.KWV);
    const int waves_k0 = kblocks_w0 >= 3 ? 4 : (kblocks_w0 >= 2 ? 2 : 1);
    const int groups0 = cdiv(p.KB, waves_k0 * p.KWV) * cdiv(p.NB, NWV) * (a.per_sample ? a.N : 1);
    const int start_px = env_int2("FS_WGRAD_MAXPX", 256);
    for (int max_px = start_px; max_px >= 32; max_px >>= 1) {
        int tw = a.Wo >= 16 ? 16 : ((a.Wo + 1) & ~1);
        tw = cdiv(cdiv(a.Wo, cdiv(a.Wo, tw)), 2) * 2;
        int th = max_px / tw;
        if (th < 1) th = 1;
        if (th > a.Ho) th = a.Ho;
        th = cdiv(a.Ho, cdiv(a.Ho, th));
        p.TH = th;
        p.TW = tw;
        p.tiles_y = cdiv(a.Ho, th);
        p.tiles_x = cdiv(a.Wo, tw);
        p.PH = (th - 1) * a.stride + a.KH;
        p.PW = (tw - 1) * a.stride + (a.KW - 1) * dil + 1;
        p.S = CS + 1;
        p.lds_bytes = 4 * (((p.PH * p.PW * p.S + 4 + 3) & ~3) + th * tw * DP);
        const bool xfit = (a.Cin & 3) || p.PH * p.PW * (CS / 4) <= 16 * 256;
        const bool dfit = th * tw * (DP / 4) <= 16 * 256;
        if (!((xfit && dfit && p.lds_bytes <= 80 * 1024) || max_px == 32)) continue;
        // CU balance: every workgroup gets the same number of tiles, but with n_wg between 256 and 512 some CUs run two
        // workgroups and the others one (e.g. 441 -> 86 % of the chip).  If halving the tile lands the workgroup count
        // just under a multiple of 256, take the smaller tile.
        if (max_px > 64 && env_int2("FS_WGRAD_BALANCE", 1)) {
            auto eff = [&](long total_tiles) {
                int want = env_int2("FS_WGRAD_WGS", 512) / groups0;
                if (want < 1) want = 1;
                const long t = total_tiles <= want ? 1 : cdiv((int)total_tiles, want);  // tiles per workgroup
                const long nwg = total_tiles <= want ? total_tiles : cdiv((int)total_tiles, (int)t);
                const long per_cu = cdiv((int)(nwg * groups0), 256);                     // workgroups on the busiest CU
                return (double)total_tiles / (double)(per_cu * t * 256 / groups0 > 0 ? per_cu * t * 256.0 / groups0 : 1.0);
            };
            const long tot_here = (long)(a.per_sample ? 1 : a.N) * p.tiles_y * p.tiles_x;
            // the half-size tile has ~2x the tiles
            if (eff(tot_here) < 0.9 && eff(2 * tot_here) > eff(tot_here) + 0.08) continue;
        }
        break;
    }
    const int kblocks_w = cdiv(p.KB, p.KWV);  // k-blocks in units of one wave's share
    p.waves_k = kblocks_w >= 3 ? 4 : (kblocks_w >= 2 ? 2 : 1);
    const int total = (a.per_sample ? 1 : a.N) * p.tiles_y * p.tiles_x;
    const int groups = cdiv(p.KB, p.waves_k * p.KWV) * cdiv(p.NB, NWV) * (a.per_sample ? a.N : 1);
    int want = env_int2("FS_WGRAD_WGS", 512) / groups;  // aim for ~2 workgroups per CU in flight
    if (want < 1) want = 1;
    // equal number of tiles per workgroup (no straggler round)
    p.n_wg = total <= want ? total : cdiv(total, cdiv(total, want));
    p.n_slabs = p.n_wg * (4 / p.waves_k);
    return p;
}

template <int KWV, int NWV>
static void launch_wgrad(const WgradArgs& a, dim3 grid, hipStream_t s) {
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(conv_wgrad_kernel<KWV, NWV>));
    hipLaunchKernelGGL((conv_wgrad_kernel<KWV, NWV>), grid, dim3(256), (size_t)a.p.lds_bytes, s, a);
}

int wgrad_launch(const WgradArgs& a_in, hipStream_t s) {
    WgradArgs a = a_in;
    if (a.dil_x < 1) a.dil_x = 1;
    const WgradPlan& p = a.p;
    a.same_xy = a.per_sample && a.x == a.dy && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.Cin == a.Cout && a.Cin <= 128 &&
                a.Cin % 32 == 0 && !a.in_a && !a.dy_a && !a.dy_unshuffle && a.src_mode == SRC_PLAIN && a.pad_t == 0 && a.pad_l == 0 &&
                tune_int("FS_GRAM_SAME", 1);
    if (a.Cin > 128 && a.Cin % 128) return -1;
    if (p.lds_bytes > 160 * 1024) return -2;
    const int NWV = p.NWV;
    dim3 grid((unsigned)p.n_wg, (unsigned)(cdiv(p.KB, p.waves_k * p.KWV) * cdiv(p.NB, NWV)), (unsigned)(a.per_sample ? a.N : 1));
    Profiler* prof = Profiler::current();
    if (prof) prof->begin(a.per_sample ? PF_GRAM_WGRAD : PF_WGRAD, 2.0 * a.N * a.Ho * a.Wo * (double)p.K * a.Cout, s);
    if (p.KWV == 1)
        launch_wgrad<1, 4>(a, grid, s);
    else if (p.KWV == 2)
        launch_wgrad<2, 2>(a, grid, s);
    else if (p.KWV == 5)
        launch_wgrad<5, 2>(a, grid, s);
    else
        launch_wgrad<4, 1>(a, grid, s);
    if (prof) prof->end(s);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
