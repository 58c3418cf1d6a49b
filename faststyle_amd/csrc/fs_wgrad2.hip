// Filter gradients of the transform-net convolutions, second generation:
//     dW[k, co] = sum over pixels p of  X[p + tap(k), ci(k)] * dY[p, co]
// (the adjoint of tf.nn.conv2d wrt its filter behind AdamOptimizer.minimize, reference train.py:203 /
// im_transf_net.py:115).  The result is tiny (<= 576 x 64) and the reduction runs over every output pixel of the
// batch, so the shape of the problem is "a few persistent workgroups, each streaming a long list of pixel tiles
// through the matrix cores into ONE resident accumulator tile":
//
//   * one workgroup per CU (4 waves, one per SIMD, up to 512 registers each), a CONTIGUOUS range of pixel tiles per
//     workgroup, at most 256 partial slabs per problem (fs::reduce_slabs_batch sums them in a fixed order:
//     deterministic, no atomics);
//   * v_mfma_f32_16x16x4_f32: M = 16 consecutive k = (tap, ci), N = 16 output channels, K = 4 consecutive pixels of a
//     tile row -- 16-wide blocks fit the 16/32-channel full-resolution layers without padding half the tile;
//     a wave owns KM k-blocks x KN channel blocks (<= 144 accumulator registers) and needs KM + KN LDS dwords per
//     KM*KN matrix instructions;
//   * two LDS stages: the global loads of tile t+1 (whole tile, in registers) are issued before the sweep of tile t
//     and committed into the other stage after it -- one barrier per tile, global latency hidden behind MFMAs;
//   * LDS pitches chosen per layer so that the 4 pixels x 16 rows of an operand fragment fall into distinct banks;
//   * several problems of one conv geometry (the ten 3x3 64->64 residual filters) run as ONE launch: workgroups are
//     dealt to problems in proportion to their tile counts.
//
// The virtual-input functor (reflect padding, zero dilation, x4 nearest upsample), the producer instance-norm + ReLU
// applied on load, the pixel-unshuffled dY of the phase-collapsed resize-conv and the on-load affine of the
// conv2d_transpose units are those of the first-generation kernel (fs_wgrad.hip), which keeps the Gram matrices,
// wide filters (K x Cout beyond one workgroup's registers) and ragged channel counts.
#include "fs_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace fs {

// an opaque identity on a vector register: the value is materialised there (no rematerialisation from its parts at the uses)
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_PIN_VGPR(x) asm volatile("" : "+v"(x))
// LDS byte address of a pointer into shared memory / a float at LDS byte address `addr` (an integer, so that base registers
// hold COMPLETE addresses: with pointer arithmetic on the shared array the backend adds the array's base in front of the reads)
#define FS_LDS_ADDR(p) ((int)(size_t)(const __attribute__((address_space(3))) char*)(p))
#define FS_LDS_F32(base, addr) (*(const __attribute__((address_space(3))) float*)(size_t)(unsigned)(addr))
#else
#define FS_PIN_VGPR(x) ((void)0)
#define FS_LDS_ADDR(p) 0
#define FS_LDS_F32(base, addr) (*reinterpret_cast<const float*>((base) + (addr)))
#endif

namespace {
constexpr int kXN = 12;   // float4 (for 3-channel inputs: pixels) of patch per thread and tile
constexpr int kDN = 8;    // float4 of dY per thread and tile
constexpr unsigned kOOB = 0x80000000u;

__device__ __forceinline__ bool w2_coord(int mode, int refl, int v, int n_src, int& s) {
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
}  // namespace

// XVEC: Cin is a multiple of 4 and the input is read as it lies in memory (SRC_PLAIN) -- 16-byte loads, every transform-net
// layer but the first; !XVEC: 3-channel inputs through any virtual-input mode (reflect padding of the image), scalar loads.
// SA > 0: STATIC TILE GEOMETRY -- SA = stride * patch pixel pitch S and SD = dY pixel pitch DP (floats), SPR = tile width / 4,
// THc = tile rows, PWc = patch width, WPc = waves over the pixel rows, all known at compile time: the sweep of a tile is one
// straight line in which every operand read is a lane-constant base register + an IMMEDIATE offset -- no vector-ALU
// instruction in the matrix-instruction slots (tools/mfma16_slots.hip: beside v_mfma_f32_16x16x4_f32 a ds_read_b32 with an
// immediate offset is free, 32.6 cycles per slot; the address add in front of it makes the slot 58.9).  Instantiated for
// the tiles the planner picks for the 9x9 layers and the residual 3x3 batch; SA = 0: any geometry, one add per read.
template <int KM, int KN, bool XVEC, int SA = 0, int SD = 0, int SPR = 0, int THc = 0, int PWc = 0, int WPc = 0>
__global__ __launch_bounds__(256) void wgrad2_kernel(Wg2Args a) {
    HIP_DYNAMIC_SHARED(float, smem)
    const Wg2Plan& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: keeps the wave's loop bounds in scalar registers)
    const int m16 = lane & 15, k4 = lane >> 4;
    // ---- problem and tile range of this workgroup (wave-uniform)
    // (the problem table is read through a pointer into the kernel-argument segment: indexing the by-value struct makes
    // the compiler preload and spill all ten descriptors)
    const Wg2Args* ka = FS_KERNARG_PTR(Wg2Args, a);
    int pi = 0;
    for (int i = 1; i < a.nprob; ++i)
        if ((int)blockIdx.x >= ka->prob[i].wg_begin) pi = i;
    const Wg2Prob& P = ka->prob[pi];
    const int H = P.H, W = P.W, Ho = P.Ho, Wo = P.Wo;
    const int tiles = P.tiles_y * P.tiles_x;
    const int total = P.N * tiles;
    const int wl = (int)blockIdx.x - P.wg_begin;
    const int t_beg = __builtin_amdgcn_readfirstlane((int)((long long)total * wl / P.wg_count));
    const int t_end = __builtin_amdgcn_readfirstlane((int)((long long)total * (wl + 1) / P.wg_count));

    const int S = p.S, DP = p.DP, PW = p.PW, PH = p.PH, TW = p.TW, TH = p.TH;
    const int waves_k = p.waves_k, waves_p = 4 / waves_k;
    const int wk = wave % waves_k, wp = wave / waves_k;
    const int kb0 = wk * KM;
    const int zero_off = PH * PW * S;   // 16 zero floats behind the patch of each stage
    auto fdiv = [](int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); };   // exact for x < 2^22

    // ---- A-operand row of this lane in every owned k-block: k -> (tap, ci) -> offset inside the patch
    // (rows beyond K -- the padding of the last 16-row block -- read element 0 of the patch: whatever they accumulate is
    // never stored, and a row of A only reaches the same row of the result)
    int abase[KM];
#pragma unroll
    for (int q = 0; q < KM; ++q) {
        const int k = (kb0 + q) * 16 + m16;
        abase[q] = 0;
        if (kb0 + q < p.KB && k < p.K) {
            const int tap = k / a.Cin, ci = k - tap * a.Cin;
            const int kh = tap / a.KW, kw = tap - kh * a.KW;
            abase[q] = (kh * PW + kw * a.dil_x) * S + ci;
        }
    }
    f32x4 acc[KM][KN];
#pragma unroll
    for (int q = 0; q < KM; ++q)
#pragma unroll
        for (int j = 0; j < KN; ++j) acc[q][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging descriptors (tile-invariant): element i of this thread is patch / tile pixel (py, px), LDS offset dst
    constexpr bool xvec = XVEC;
    const int c4n = xvec ? a.Cin >> 2 : 1;
    const int npx = PH * PW;
    const int ne_x = npx * c4n;   // 3-channel inputs (!XVEC): one element per patch pixel, its three floats loaded separately
    // element i of this thread: patch pixel (py << 16 | px) and the thread's fixed channel quad; kNone: not owned
    constexpr int kNone = 0x40000000;
    int xq[kXN];
    const int xch = xvec ? (tid % c4n) * 4 : 0;                     // (c4n is a power of two <= 64: fixed per thread)
#pragma unroll
    for (int i = 0; i < kXN; ++i) {
        const int e = tid + i * 256;
        xq[i] = kNone;   // fails every bounds test -> zeros into the sink
        if (i < p.xn && e < ne_x) {
            const int pix = e / c4n;
            const int py = pix / PW;
            xq[i] = (py << 16) | (pix - py * PW);
        }
    }
    const int j4n = a.Cout >> 2;
    const int ne_d = TH * TW * j4n;
    const int Cr = a.dy_unshuffle ? a.Cout >> 2 : a.Cout;
    int dq[kDN];
    const int dco = (tid % j4n) * 4;                                 // (j4n is a power of two <= 64)
    int d_cbase = dco, d_ystep = Wo * a.Cout, d_xstep = a.Cout;      // element offset = oy*ystep + ox*xstep + cbase
    if (a.dy_unshuffle) {
        const int q = dco / Cr, cr = dco - q * Cr;
        d_cbase = ((q >> 1) * 2 * Wo + (q & 1)) * Cr + cr;
        d_ystep = 4 * Wo * Cr;
        d_xstep = 2 * Cr;
    }
#pragma unroll
    for (int i = 0; i < kDN; ++i) {
        const int e = tid + i * 256;
        dq[i] = kNone;
        if (i < p.dn && e < ne_d) {
            const int pix = e / j4n;
            const int py = pix / TW;
            dq[i] = (py << 16) | (pix - py * TW);
        }
    }
    const bool has_ab = P.in_a != nullptr, has_dab = P.dy_a != nullptr;
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(H * W * a.Cin) * 4u);
    const unsigned d_bytes = __builtin_amdgcn_readfirstlane((unsigned)(Ho * Wo * a.Cout) * 4u);
    auto uniform_ptr = [](const float* ptr) {  // a wave-uniform pointer, pinned to scalar registers
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };
    const float inv_tiles = 1.0f / (float)tiles, inv_tx = 1.0f / (float)P.tiles_x;

    float4 xv[kXN], dv[kDN];
    unsigned xok = 0, dok = 0;
    float4 va = make_float4(1.f, 1.f, 1.f, 1.f), da = make_float4(1.f, 1.f, 1.f, 1.f);
    uint4 vb = make_uint4(0u, 0u, 0u, 0u), db = make_uint4(0u, 0u, 0u, 0u);

    // global -> registers: the whole tile (patch with halo + dY), every load independent of the others
    auto issue = [&](int t) {
        const int n = __builtin_amdgcn_readfirstlane(fdiv(t, inv_tiles));
        const int tr = t - n * tiles;
        const int tyi = __builtin_amdgcn_readfirstlane(fdiv(tr, inv_tx));
        const int ty0 = tyi * TH, tx0 = (tr - tyi * P.tiles_x) * TW;
        const int vy0 = ty0 * a.stride - a.pad_t, vx0 = tx0 * a.stride - a.pad_l;
        const float* xn = uniform_ptr(P.x + (size_t)n * H * W * a.Cin);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, x_bytes, 0x00020000);
        xok = 0;
        if (xvec) {
#pragma unroll
            for (int i = 0; i < kXN; ++i) {
                if (i >= p.xn) break;
                int sy, sx;
                const bool ok = w2_coord(SRC_PLAIN, 0, vy0 + (xq[i] >> 16), H, sy) && w2_coord(SRC_PLAIN, 0, vx0 + (xq[i] & 0xFFFF), W, sx);
                const unsigned vo = ok ? (unsigned)((sy * W + sx) * a.Cin + xch) * 4u : kOOB;
                xv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, vo, 0, 0));
                xok |= ok ? 1u << i : 0u;
            }
            if (has_ab) {
                va = *reinterpret_cast<const float4*>(P.in_a + (size_t)n * a.in_nstride + xch);
                vb = *reinterpret_cast<const uint4*>(P.in_b + (size_t)n * a.in_nstride + xch);
            }
        } else {  // 3-channel inputs (the image itself, or dz of the last conv2d_transpose): three dwords per pixel, no affine
            // mirror padding by `refl` pixels (tf.pad REFLECT, im_transf_net.py:78-88); refl = 0 is the plain image
            const int refl = a.src_mode == SRC_REFLECT ? a.refl : 0;
#pragma unroll
            for (int i = 0; i < kXN; ++i) {
                if (i >= p.xn) break;
                const int vy = vy0 + (xq[i] >> 16), vx = vx0 + (xq[i] & 0xFFFF);
                const bool ok = (unsigned)vy < (unsigned)(H + 2 * refl) && (unsigned)vx < (unsigned)(W + 2 * refl);   // (kNone fails)
                int sy = vy - refl, sx = vx - refl;
                sy = sy < 0 ? -sy : sy;
                sx = sx < 0 ? -sx : sx;
                sy = sy >= H ? 2 * (H - 1) - sy : sy;
                sx = sx >= W ? 2 * (W - 1) - sx : sx;
                const unsigned vo = ok ? (unsigned)((sy * W + sx) * 3) * 4u : kOOB;
                xv[i].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, vo, 0, 0));
                xv[i].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, vo, 4, 0));
                xv[i].z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, vo, 8, 0));
            }
        }
        const float* dyn = uniform_ptr(P.dy + (size_t)n * Ho * Wo * a.Cout);
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dyn), 0, d_bytes, 0x00020000);
        dok = 0;
#pragma unroll
        for (int i = 0; i < kDN; ++i) {
            if (i >= p.dn) break;
            const int oy = ty0 + (dq[i] >> 16), ox = tx0 + (dq[i] & 0xFFFF);
            const bool ok = oy < Ho && ox < Wo;
            const unsigned vo = ok ? (unsigned)(oy * d_ystep + ox * d_xstep + d_cbase) * 4u : kOOB;
            dv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(dr, vo, 0, 0));
            dok |= ok ? 1u << i : 0u;
        }
        if (has_dab) {
            da = *reinterpret_cast<const float4*>(P.dy_a + (size_t)n * a.dy_nstride + dco);
            db = *reinterpret_cast<const uint4*>(P.dy_b + (size_t)n * a.dy_nstride + dco);
        }
    };
    // registers -> LDS stage (producer instance norm + ReLU applied here; padding arrives as 0 and stays 0)
    auto commit = [&](float* stage) {
        float* patch = stage;
        float* dyl = stage + p.patch_floats;
        if (xvec) {
#pragma unroll
            for (int i = 0; i < kXN; ++i) {
                if (i >= p.xn) break;
                float4 v = xv[i];
                if (has_ab) {
                    const unsigned okm = (xok >> i) & 1u ? 0xFFFFFFFFu : 0u;
                    v.x = fmaf(v.x, va.x, __uint_as_float(vb.x & okm));
                    v.y = fmaf(v.y, va.y, __uint_as_float(vb.y & okm));
                    v.z = fmaf(v.z, va.z, __uint_as_float(vb.z & okm));
                    v.w = fmaf(v.w, va.w, __uint_as_float(vb.w & okm));
                }
                if (has_ab && a.in_relu) {   // (the ReLU belongs to the producer's instance norm: problems without one skip both)
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f);
                    v.w = fmaxf(v.w, 0.f);
                }
                // (S and the channel quad are multiples of 4: 16-byte aligned; unowned elements carry zeros into the sink)
                const int dst = xq[i] != kNone ? ((xq[i] >> 16) * PW + (xq[i] & 0xFFFF)) * S + xch : zero_off;
                *reinterpret_cast<float4*>(patch + dst) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kXN; ++i) {
                if (i >= p.xn) break;
                // (pixels the thread does not own were loaded as zeros and go to the zero sink)
                const int dst = xq[i] != kNone ? ((xq[i] >> 16) * PW + (xq[i] & 0xFFFF)) * 3 : zero_off;
                patch[dst] = xv[i].x;
                patch[dst + 1] = xv[i].y;
                patch[dst + 2] = xv[i].z;
            }
        }
#pragma unroll
        for (int i = 0; i < kDN; ++i) {
            if (i >= p.dn) break;
            float4 v = dv[i];
            if (has_dab) {
                const unsigned okm = (dok >> i) & 1u ? 0xFFFFFFFFu : 0u;
                v.x = fmaf(v.x, da.x, __uint_as_float(db.x & okm));
                v.y = fmaf(v.y, da.y, __uint_as_float(db.y & okm));
                v.z = fmaf(v.z, da.z, __uint_as_float(db.z & okm));
                v.w = fmaf(v.w, da.w, __uint_as_float(db.w & okm));
            }
            if (has_dab && a.dy_relu) {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
            }
            if (dq[i] != kNone) *reinterpret_cast<float4*>(dyl + ((dq[i] >> 16) * TW + (dq[i] & 0xFFFF)) * DP + dco) = v;
        }
    };

    // ---- MFMA sweep of one staged tile: 4 consecutive pixels of a tile row per instruction; operands of the next step
    // are read while the matrix instructions of the current one issue (two register sets)
    const int spr = TW >> 2;                                  // steps per tile row
    const int rows_w = (TH - wp + waves_p - 1) / waves_p;     // rows of this wave: wp, wp + waves_p, ...
    const int nsteps = rows_w * spr;
    const bool active = kb0 < p.KB;
    // (k-blocks beyond K read the zero slack and are multiplied like the others: a per-instruction guard would put every
    // matrix instruction behind its own branch)
    // Every operand read of step s+1 is PINNED into a matrix-instruction slot of step s (one address add + one ds_read_b32 per
    // slot, sched_barrier after every slot), two register sets.  Round 2 left the placement to sched_group_barrier hints: the
    // backend then sank most reads of a step to just before their own matrix instruction (ds_read, s_waitcnt lgkmcnt(0),
    // v_mfma -- the LDS latency exposed on every instruction: 107 cycles per matrix instruction in the 9x9 instances against
    // 32 of issue time; ablation FS_WGRAD2_DEBUG=2).  The lane's share of every operand address is folded into fixed byte
    // offsets (ab[], bb), so a step's offset is two scalars.
    int ab[KM];
#pragma unroll
    for (int q = 0; q < KM; ++q) ab[q] = (abase[q] + k4 * a.stride * S) * 4;
    const int bb = (p.patch_floats + m16 + k4 * DP) * 4;
    const char* const lds0 = reinterpret_cast<const char*>(smem);
    auto sweep = [&](const float* stage) __attribute__((always_inline)) {
        const int st = (int)(stage - smem) * 4;               // byte offset of the stage (wave-uniform)
        constexpr int NM = KM * KN, NL = KM + KN;
        float a0[KM], b0[KN], a1[KM], b1[KN];
        if constexpr (SA > 0) {
            constexpr int STEP_A = 16 * SA, STEP_B = 16 * SD;                              // bytes per step (4 pixels)
            constexpr int ROW_A = PWc * SA * 4 * WPc, ROW_B = 4 * SPR * SD * 4 * WPc;      // bytes between two rows of a wave
            constexpr int RW = THc / WPc;                                                  // rows per tile and wave
            const int l0 = FS_LDS_ADDR(lds0);   // (0 on the CPU emulator, where FS_LDS_F32 adds the array's address instead)
            int rb[KM];
#pragma unroll
            for (int q = 0; q < KM; ++q) {
                rb[q] = ab[q] + (l0 + st + wp * (ROW_A / WPc));
                FS_PIN_VGPR(rb[q]);   // (or the backend keeps "lane part + scalar row offset" apart and re-adds them in front of every read)
            }
            int rbB = bb + (l0 + st + wp * (ROW_B / WPc));
            FS_PIN_VGPR(rbB);
#pragma unroll
            for (int j = 0; j < KN; ++j) b0[j] = FS_LDS_F32(lds0, rbB + j * 64);
#pragma unroll
            for (int q = 0; q < KM; ++q) a0[q] = FS_LDS_F32(lds0, rb[q]);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_s_waitcnt(0xC07F);   // (nothing may be pending on the way into the row loop: see the generic path)
#endif
            // one step: the matrix instructions of (ca, cb); read r of the NEXT step (bases + off_a / off_b) rides in slot
            // r * NM / NL
            auto step = [&](int off_a, int off_b, const float (&ca)[KM], const float (&cb)[KN], float (&na)[KM], float (&nb)[KN]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < NM; ++i) {
#pragma unroll
                    for (int r = 0; r < NL; ++r) {
                        if (r * NM / NL != i) continue;
                        if (r < KN)
                            nb[r] = FS_LDS_F32(lds0, rbB + (off_b + r * 64));
                        else
                            na[r - KN] = FS_LDS_F32(lds0, rb[r - KN] + off_a);
                    }
                    acc[i / KN][i % KN] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[i / KN], cb[i % KN], acc[i / KN][i % KN], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // a ROLLED loop over the wave's rows (the accumulators stay in place across its back edge; a fully unrolled tile
            // made the register allocator alternate between two accumulator sets), the SPR steps of a row unrolled.  The bases
            // move to the next row in front of the row's last step, which prefetches that row's first (past the last row: the
            // first row again -- a prefetch nobody uses).  SPR is even: a row starts on register set 0.
#pragma unroll 1
            for (int r = 0; r < RW; ++r) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cx = 0; cx + 2 < SPR; cx += 2) {
                    step((cx + 1) * STEP_A, (cx + 1) * STEP_B, a0, b0, a1, b1);
                    step((cx + 2) * STEP_A, (cx + 2) * STEP_B, a1, b1, a0, b0);
                }
                step((SPR - 1) * STEP_A, (SPR - 1) * STEP_B, a0, b0, a1, b1);
                const int d_a = r + 1 == RW ? -(RW - 1) * ROW_A : ROW_A, d_b = r + 1 == RW ? -(RW - 1) * ROW_B : ROW_B;
#pragma unroll
                for (int q = 0; q < KM; ++q) {
                    rb[q] += d_a;
                    FS_PIN_VGPR(rb[q]);
                }
                rbB += d_b;
                FS_PIN_VGPR(rbB);
                __builtin_amdgcn_sched_barrier(0);
                step(0, 0, a1, b1, a0, b0);
            }
            return;
        }
        int cy = wp, cx = 0;                                  // load cursor
        int sa = 0, sb = 0;
        auto cursor = [&]() __attribute__((always_inline)) {  // byte offsets of the cursor's step, then advance it
            sa = st + (cy * a.stride * PW + cx * 4 * a.stride) * S * 4;
            sb = st + (cy * TW + cx * 4) * DP * 4;
            if (++cx == spr) {
                cx = 0;
                cy += waves_p;
                if (cy >= TH) cy = wp;   // past the last step: wrap (the prefetch of a step that does not exist re-reads step 0)
            }
        };
        cursor();
#pragma unroll
        for (int j = 0; j < KN; ++j) b0[j] = *reinterpret_cast<const float*>(lds0 + (bb + sb) + j * 64);
#pragma unroll
        for (int q = 0; q < KM; ++q) a0[q] = *reinterpret_cast<const float*>(lds0 + (ab[q] + sa));
#if defined(__HIP_DEVICE_COMPILE__)
        // (waits are placed statically: anything still pending on the way INTO the loop -- these reads, a scalar load --
        // becomes an lgkmcnt(0) in front of the first matrix instruction of EVERY iteration; drain once here instead)
        __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
        // one step: the matrix instructions of (ca, cb); read r of the next step rides in slot r * NM / NL (B operands first:
        // the first matrix instruction of the next step needs them)
        auto step = [&](const float (&ca)[KM], const float (&cb)[KN], float (&na)[KM], float (&nb)[KN]) __attribute__((always_inline)) {
            cursor();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
#pragma unroll
                for (int r = 0; r < NL; ++r) {
                    if (r * NM / NL != i) continue;
                    if (r < KN)
                        nb[r] = *reinterpret_cast<const float*>(lds0 + (bb + sb) + r * 64);
                    else
                        na[r - KN] = *reinterpret_cast<const float*>(lds0 + (ab[r - KN] + sa));
                }
                acc[i / KN][i % KN] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[i / KN], cb[i % KN], acc[i / KN][i % KN], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // nsteps is even (the planner picks tile widths that are multiples of 8), so the body is branch-free
        for (int s = 0; s < nsteps; s += 2) {
            step(a0, b0, a1, b1);
            step(a1, b1, a0, b0);
        }
    };

    // ---- pipeline over the tile range
    float* st0 = smem;
    float* st1 = smem + p.stage_floats;
    if (tid < 32) {   // zero slack / sink of both stages (never written with anything but zeros afterwards)
        (tid < 16 ? st0 : st1)[zero_off + (tid & 15)] = 0.f;
    }
    __syncthreads();
    // iteration t sweeps tile t (none at t = t_beg - 1) while tile t+1 travels: global -> registers before the sweep,
    // registers -> the other LDS stage after it; one barrier per tile
    for (int t = t_beg - 1; t < t_end; ++t) {
        float* cur = ((t - t_beg) & 1) ? st1 : st0;
        float* nxt = ((t - t_beg) & 1) ? st0 : st1;
        const bool more = t + 1 < t_end;
        if (more && !(a.debug & 2 && t >= t_beg)) issue(t + 1);
        if (t >= t_beg && active && nsteps > 0 && !(a.debug & 1)) sweep(cur);
        if (more && !(a.debug & 2 && t >= t_beg)) commit(nxt);
        FS_WAIT_VMEM_FENCED();   // (free: commit consumed every load of this iteration -- tells the wait-count pass so; fs_kernels.h)
        __syncthreads();
    }

    // ---- the pixel-row groups of the workgroup summed through LDS (round 5; both stages are free, the loop ended on a barrier): group w = 1,
    // 2, .. hands its accumulators to group 0 lane for lane -- the groups of one wk hold the same elements in the same (register, lane) places --
    // in that fixed order.  One slab per workgroup instead of 4 / waves_k: a quarter / half of the slab bytes written here and read by the reduction.
    if (p.combine) {
        float* cb = smem + (size_t)wk * (KM * KN * 4 * 64) + lane;
        for (int w = 1; w < waves_p; ++w) {
            if (wp == w) {
#pragma unroll
                for (int q = 0; q < KM; ++q)
#pragma unroll
                    for (int j = 0; j < KN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) cb[((q * KN + j) * 4 + r) * 64] = acc[q][j][r];
            }
            __syncthreads();
            if (wp == 0) {
#pragma unroll
                for (int q = 0; q < KM; ++q)
#pragma unroll
                    for (int j = 0; j < KN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[q][j][r] += cb[((q * KN + j) * 4 + r) * 64];
            }
            __syncthreads();
        }
        if (wp != 0) return;
    }
    // ---- this wave's part of the workgroup's partial slab(s): [K][Cout]; !combine: one slab per pixel-row group wp
    float* slab = P.slabs + (p.combine ? (size_t)wl : (size_t)wl * waves_p + wp) * (size_t)p.K * a.Cout;
#pragma unroll
    for (int q = 0; q < KM; ++q) {
        if (kb0 + q >= p.KB) continue;
#pragma unroll
        for (int j = 0; j < KN; ++j) {
            if (j >= p.NB) continue;
            const int co = j * 16 + m16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = (kb0 + q) * 16 + 4 * k4 + r;
                if (kk < p.K) slab[(size_t)kk * a.Cout + co] = acc[q][j][r];
            }
        }
    }
}

// out[job][i] = scale * sum over that job's slabs, in a fixed order (deterministic).  A thread owns 4 consecutive elements
// (one float4 per slab), 8 threads cover a 128-byte line, the 32 slab groups of a workgroup keep <= 8 independent loads
// per thread in flight; the groups are combined through LDS.
__global__ __launch_bounds__(256) void reduce_slabs_batch_kernel(Wg2Reduce r) {
    constexpr int EL4 = 8, SG = 32;
    __shared__ float4 sh[256];
    const Wg2Reduce* kr = FS_KERNARG_PTR(Wg2Reduce, r);
    const Wg2Reduce::Job& jb = kr->job[blockIdx.y];
    const size_t count = jb.count;
    const int n_slabs = jb.n_slabs;
    const int el = threadIdx.x & (EL4 - 1), sg = threadIdx.x >> 3;
    const size_t i = ((size_t)blockIdx.x * EL4 + el) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < count) {
        const float* p = jb.slabs + i;
        int w = sg;
        for (; w + 7 * SG < n_slabs; w += 8 * SG) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(w + u * SG) * count);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc.x += v[u].x;
                acc.y += v[u].y;
                acc.z += v[u].z;
                acc.w += v[u].w;
            }
        }
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {   // tail: clamped slab index, contribution masked (keeps the loads unconditional)
            const int ww = w + u * SG;
            v[u] = *reinterpret_cast<const float4*>(p + (size_t)(ww < n_slabs ? ww : 0) * count);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float m = (w + u * SG) < n_slabs ? 1.f : 0.f;
            acc.x = fmaf(v[u].x, m, acc.x);
            acc.y = fmaf(v[u].y, m, acc.y);
            acc.z = fmaf(v[u].z, m, acc.z);
            acc.w = fmaf(v[u].w, m, acc.w);
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (sg == 0 && i < count) {
        float4 t = sh[el];
#pragma unroll
        for (int k = 1; k < SG; ++k) {
            const float4 q = sh[k * EL4 + el];
            t.x += q.x;
            t.y += q.y;
            t.z += q.z;
            t.w += q.w;
        }
        const float sc = r.scale;
        *reinterpret_cast<float4*>(jb.out + i) = make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc);
    }
}

// ------------------------------------------------------------------------------------------------ host
static int w2_pitch(int c, int step) {
    // smallest pitch >= c (multiple of 4) with  pitch * step == 16 (mod 32): the four pixels of an operand fragment
    // (16 consecutive floats each, `step` pixels apart) then occupy distinct LDS banks
    for (int s = (c + 3) & ~3; s < c + 64; s += 4)
        if ((s * step) % 32 == 16) return s;
    return (c + 3) & ~3;   // (step a multiple of 32: no conflict-free pitch; never the case for stride 1 / 2)
}

bool wgrad2_eligible(const WgradArgs& a) {
    if (a.per_sample) return false;                                   // Gram matrices: first-generation kernel
    if ((a.in_relu && !a.in_a) || (a.dy_relu && !a.dy_a)) return false;   // (a ReLU on load only comes with its affine here)
    if (a.Cout % 16 || a.Cout > 128) return false;
    const bool vec_ok = a.Cin % 4 == 0 && a.Cin <= 256 && ((a.Cin / 4) & (a.Cin / 4 - 1)) == 0 && a.src_mode == SRC_PLAIN;
    const bool scalar_ok = a.Cin == 3 && !a.in_a && a.Cout == 16 && (a.src_mode == SRC_PLAIN || a.src_mode == SRC_REFLECT);
    if (!vec_ok && !scalar_ok) return false;
    if (((a.Cout / 4) & (a.Cout / 4 - 1)) != 0) return false;
    if (a.stride != 1 && a.stride != 2) return false;
    const int K = a.KH * a.KW * a.Cin, KB = cdiv(K, 16), NB = a.Cout / 16;
    // accumulator tile of one workgroup: 4 waves x (KM x KN) blocks
    if (NB > 8) return false;
    const int KM = NB > 4 ? 4 : (NB >= 2 ? 9 : 18);
    return KB <= 4 * KM && tune_int("FS_WGRAD2", 1) != 0;
}

static void w2_variant(int KB, int NB, int* KM, int* KN, int* waves_k) {
    if (NB > 4) {
        *KM = 4;
        *KN = 8;
    } else if (NB > 2) {
        *KM = 9;
        *KN = 4;
    } else if (NB == 2) {
        *KM = 9;
        *KN = 2;
    } else {
        *KM = 18;
        *KN = 1;
    }
    const int need = cdiv(KB, *KM);
    *waves_k = need >= 3 ? 4 : need;
}

// Plans `n` problems that share the conv geometry of probs[0] (image shapes and pointers may differ): tile, pitches, wave
// roles, and the deal of workgroups to problems.  The on-load affine of x (in_a/in_b: the producer's instance norm + ReLU)
// and of dy may be present in some problems and absent in others -- the kernel gates it on the problem's pointer -- but
// every problem that HAS one must agree on its flags (in_nstride / in_relu, dy_nstride / dy_relu), which are launch-wide.
// Returns the number of slab floats the launch needs (0: not eligible -- the caller launches per problem).
size_t wgrad2_plan(const WgradArgs* probs, int n, Wg2Args* out) {
    if (n < 1 || n > kW2MaxProb || !wgrad2_eligible(probs[0])) return 0;
    const WgradArgs& g = probs[0];
    const WgradArgs* with_in = nullptr;   // first problem with an x affine / a dy affine: the source of the launch-wide flags
    const WgradArgs* with_dy = nullptr;
    for (int i = 0; i < n; ++i) {
        const WgradArgs& q = probs[i];
        const int gd = g.dil_x > 0 ? g.dil_x : 1, qd = q.dil_x > 0 ? q.dil_x : 1;
        if (i > 0 && (!wgrad2_eligible(q) || q.Cin != g.Cin || q.Cout != g.Cout || q.KH != g.KH || q.KW != g.KW || q.stride != g.stride ||
                      q.pad_t != g.pad_t || q.pad_l != g.pad_l || qd != gd || q.src_mode != g.src_mode || q.refl != g.refl ||
                      q.dy_unshuffle != g.dy_unshuffle || q.per_sample != g.per_sample))
            return 0;
        if (q.in_a) {
            if (!with_in) with_in = &q;
            if (q.in_nstride != with_in->in_nstride || q.in_relu != with_in->in_relu) return 0;
        }
        if (q.dy_a) {
            if (!with_dy) with_dy = &q;
            if (q.dy_nstride != with_dy->dy_nstride || q.dy_relu != with_dy->dy_relu) return 0;
        }
    }
    Wg2Args A{};
    Wg2Plan& p = A.p;
    A.Cin = g.Cin;
    A.Cout = g.Cout;
    A.KH = g.KH;
    A.KW = g.KW;
    A.stride = g.stride;
    A.pad_t = g.pad_t;
    A.pad_l = g.pad_l;
    A.dil_x = g.dil_x > 0 ? g.dil_x : 1;
    A.src_mode = g.src_mode;
    A.refl = g.refl;
    A.in_nstride = with_in ? with_in->in_nstride : 0;
    A.in_relu = with_in ? with_in->in_relu : 0;
    A.dy_nstride = with_dy ? with_dy->dy_nstride : 0;
    A.dy_relu = with_dy ? with_dy->dy_relu : 0;
    A.dy_unshuffle = g.dy_unshuffle;
    A.debug = tune_int("FS_WGRAD2_DEBUG", 0);   // timing experiments only: 1 skips the sweeps, 2 stages only the first tile
    p.K = g.KH * g.KW * g.Cin;
    p.KB = cdiv(p.K, 16);
    p.NB = g.Cout / 16;
    w2_variant(p.KB, p.NB, &p.KM, &p.KN, &p.waves_k);
    p.S = g.Cin == 3 ? 3 : w2_pitch(g.Cin, g.stride);
    p.DP = w2_pitch(g.Cout, 1);
    const int waves_p = 4 / p.waves_k;
    // pixel tile: rows x (multiple of 4) columns; both LDS stages within 160 KiB, the whole tile within the per-thread
    // register prefetch (kXN / kDN float4), as many pixels as that allows but at least ~8 tiles per workgroup when the
    // problem is large enough, and the best fill of the image among the candidates
    int maxHo = 0, maxWo = 0;
    long px_total = 0;
    for (int i = 0; i < n; ++i) {
        if (probs[i].Ho > maxHo) maxHo = probs[i].Ho;
        if (probs[i].Wo > maxWo) maxWo = probs[i].Wo;
        px_total += (long)probs[i].N * probs[i].Ho * probs[i].Wo;
    }
    const int n_wg_max = tune_int("FS_WGRAD2_WGS", 256);
    const long want_px = px_total / ((long)n_wg_max * 8) > 32 ? px_total / ((long)n_wg_max * 8) : 32;   // pixels per tile for >= 8 tiles/workgroup
    double best = -1;
    for (int tw = 8; tw <= 32 && tw <= ((maxWo + 7) & ~7); tw += 8)
        for (int th = waves_p; th <= 16 && th <= ((maxHo + waves_p - 1) / waves_p) * waves_p; th += waves_p) {
            const int PH = (th - 1) * g.stride + g.KH, PW = (tw - 1) * g.stride + (g.KW - 1) * A.dil_x + 1;
            const int patch_floats = ((PH * PW * p.S + 16 + 3) & ~3);
            const int stage = patch_floats + th * tw * p.DP;
            const int ne_x = g.Cin == 3 ? PH * PW : PH * PW * (g.Cin / 4);
            const int ne_d = th * tw * (g.Cout / 4);
            if (2 * stage * 4 > 160 * 1024 || ne_x > kXN * 256 || ne_d > kDN * 256) continue;
            double fill = 0, tot = 0;
            for (int i = 0; i < n; ++i) {
                const double img = (double)probs[i].Ho * probs[i].Wo;
                fill += img / ((double)cdiv(probs[i].Ho, th) * th * cdiv(probs[i].Wo, tw) * tw) * img;
                tot += img;
            }
            fill /= tot;
            const double halo = (double)(PH * PW) / ((double)th * tw * g.stride * g.stride);
            const double size_pen = th * tw > 2 * want_px ? 0.85 : 1.0;       // too few tiles per workgroup: imbalance
            const double score = fill * size_pen / (1.0 + 0.15 * (halo - 1.0)) * (th * tw >= 32 ? 1.0 : 0.8);
            if (score > best) {
                best = score;
                p.TH = th;
                p.TW = tw;
                p.PH = PH;
                p.PW = PW;
                p.patch_floats = patch_floats;
                p.stage_floats = stage;
                p.xn = cdiv(ne_x, 256);
                p.dn = cdiv(ne_d, 256);
            }
        }
    if (best < 0) return 0;
    p.lds_bytes = 2 * p.stage_floats * 4;
    p.combine = waves_p > 1 && tune_int("FS_WGRAD2_COMBINE", 1) != 0;
    if (p.combine && p.lds_bytes < p.waves_k * p.KM * p.KN * 4 * 64 * 4) p.lds_bytes = p.waves_k * p.KM * p.KN * 4 * 64 * 4;   // the hand-over buffer of one group
    // deal workgroups to problems in proportion to their tile counts (same geometry: same cost per tile)
    long tiles_total = 0;
    long tiles[kW2MaxProb];
    for (int i = 0; i < n; ++i) {
        tiles[i] = (long)probs[i].N * cdiv(probs[i].Ho, p.TH) * cdiv(probs[i].Wo, p.TW);
        tiles_total += tiles[i];
    }
    int wg_left = (int)(tiles_total < n_wg_max ? tiles_total : n_wg_max);
    if (wg_left < n) wg_left = n;
    long tiles_left = tiles_total;
    int wg_begin = 0;
    size_t slab_floats = 0;
    A.nprob = n;
    for (int i = 0; i < n; ++i) {
        Wg2Prob& P = A.prob[i];
        const WgradArgs& q = probs[i];
        int cnt = (int)((tiles[i] * wg_left + tiles_left - 1) / tiles_left);
        if (cnt < 1) cnt = 1;
        if (cnt > wg_left - (n - 1 - i)) cnt = wg_left - (n - 1 - i);
        if (cnt > tiles[i]) cnt = (int)tiles[i];
        if (cnt < 1) cnt = 1;
        P.x = q.x;
        P.dy = q.dy;
        P.in_a = q.in_a;
        P.in_b = q.in_b;
        P.dy_a = q.dy_a;
        P.dy_b = q.dy_b;
        P.N = q.N;
        P.H = q.H;
        P.W = q.W;
        P.Ho = q.Ho;
        P.Wo = q.Wo;
        P.tiles_y = cdiv(q.Ho, p.TH);
        P.tiles_x = cdiv(q.Wo, p.TW);
        P.wg_begin = wg_begin;
        P.wg_count = cnt;
        P.slab_off = slab_floats;
        wg_begin += cnt;
        wg_left -= cnt;
        tiles_left -= tiles[i];
        slab_floats += (size_t)cnt * (p.combine ? 1 : waves_p) * p.K * g.Cout;
    }
    A.n_wg = wg_begin;
    *out = A;
    return slab_floats;
}

template <int KM, int KN, bool XVEC, int SA = 0, int SD = 0, int SPR = 0, int THc = 0, int PWc = 0, int WPc = 0>
static void w2_launch(const Wg2Args& a, hipStream_t s) {
    static BigLds lds_attr;
    lds_attr.ensure(reinterpret_cast<const void*>(wgrad2_kernel<KM, KN, XVEC, SA, SD, SPR, THc, PWc, WPc>));
    hipLaunchKernelGGL((wgrad2_kernel<KM, KN, XVEC, SA, SD, SPR, THc, PWc, WPc>), dim3((unsigned)a.n_wg), dim3(256), (size_t)a.p.lds_bytes, s, a);
}

// the slab reduction of one or several wgrad2_run calls (same scale) as ONE launch
int wgrad2_reduce(const Wg2Reduce& r, hipStream_t s) {
    if (r.n <= 0) return 0;
    if (r.n > kW2MaxProb) return -1;
    size_t max4 = 0;
    for (int i = 0; i < r.n; ++i)
        if (r.job[i].count / 4 > max4) max4 = r.job[i].count / 4;
    hipLaunchKernelGGL(reduce_slabs_batch_kernel, dim3((unsigned)((max4 + 7) / 8), (unsigned)r.n), dim3(256), 0, s, r);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Launches a planned batch (slab pointers are bound here) and the reduction into dw[i] (scale applied).  defer != nullptr: the reduction jobs
// are APPENDED to *defer instead (same scale; the caller launches wgrad2_reduce once for several runs -- the slabs must stay until then).
int wgrad2_run(const Wg2Args& planned, float* slabs, float* const* dw, float scale, hipStream_t s, Wg2Reduce* defer) {
    Wg2Args a = planned;
    const int waves_p = 4 / a.p.waves_k;
    Wg2Reduce r{};
    r.scale = scale;
    r.n = a.nprob;
    size_t max4 = 0;
    double flops = 0;
    for (int i = 0; i < a.nprob; ++i) {
        a.prob[i].slabs = slabs + a.prob[i].slab_off;
        r.job[i].slabs = a.prob[i].slabs;
        r.job[i].n_slabs = a.prob[i].wg_count * (a.p.combine ? 1 : waves_p);
        r.job[i].count = (size_t)a.p.K * a.Cout;
        r.job[i].out = dw[i];
        if (r.job[i].count / 4 > max4) max4 = r.job[i].count / 4;
        flops += 2.0 * a.prob[i].N * a.prob[i].Ho * a.prob[i].Wo * (double)a.p.K * a.Cout;
    }
    if (((size_t)a.p.K * a.Cout) % 4) return -1;
    Profiler* prof = Profiler::current();
    if (prof) prof->begin(PF_WGRAD2, flops, s);
    // instances with static tile geometry (immediate-offset operand reads) for the tiles the planner picks on the transform
    // net's 9x9 layers (256x256 at batch 32 / batch 4 per GPU) and residual batch; anything else -- and everything under
    // FS_WGRAD2_STATIC=0 -- takes the any-geometry instance of its (KM, KN)
    const Wg2Plan& p = a.p;
    const int wvp = 4 / p.waves_k;
    auto geo = [&](int sa, int sd, int spr, int th, int pw, int wp) {
        return tune_int("FS_WGRAD2_STATIC", 1) != 0 && a.stride * p.S == sa && p.DP == sd && p.TW == 4 * spr && p.TH == th && p.PW == pw && wvp == wp;
    };
    if (tune_int("FS_CONV_DEBUG", 0))
        fprintf(stderr, "wgrad2: KM %d KN %d Cin %d Cout %d K %d stride %d S %d DP %d tile %dx%d PW %d waves_p %d wgs %d nprob %d\n", p.KM, p.KN, a.Cin,
                a.Cout, p.K, a.stride, p.S, p.DP, p.TH, p.TW, p.PW, wvp, a.n_wg, a.nprob);
    // (geometry: stride * S, DP, tile width / 4, tile rows, patch width, waves over the pixel rows)
#define FS_W2_GEO(KM, KN, XV, SA, SD, SPR, TH, PW, WP) \
    if (geo(SA, SD, SPR, TH, PW, WP)) {                                                        \
        if (tune_int("FS_CONV_DEBUG", 0)) fprintf(stderr, "wgrad2: static instance %d %d | %d %d %d %d %d %d\n", KM, KN, SA, SD, SPR, TH, PW, WP); \
        w2_launch<KM, KN, XV, SA, SD, SPR, TH, PW, WP>(a, s);                                  \
    } else
    if (a.Cin == 3) {
        FS_W2_GEO(18, 1, false, 3, 16, 6, 16, 32, 4)      // image layer 9x9, 256 + 80 = 336-pixel maps: 16 x 24 tiles
        FS_W2_GEO(18, 1, false, 3, 16, 4, 16, 24, 4)      // ... 16 x 16 tiles (maps that are multiples of 16 only)
        w2_launch<18, 1, false>(a, s);
    } else if (p.KN == 8) {
        w2_launch<4, 8, true>(a, s);   // (first resize-conv, 64 -> 4 x 32: a static instance measured SLOWER, 176 vs 124 us at batch 32)
    } else if (p.KN == 4) {
        FS_W2_GEO(9, 4, true, 80, 80, 2, 8, 10, 1)        // the ten residual 3x3 filters (one launch)
        FS_W2_GEO(9, 4, true, 80, 80, 2, 6, 17, 2)        // second stride-2 conv (32 -> 64)
        FS_W2_GEO(9, 4, true, 48, 80, 2, 16, 9, 4)        // second resize-conv (32 -> 4 x 16), batch 32
        FS_W2_GEO(9, 4, true, 48, 80, 2, 8, 9, 4)         // ... batch 4 per GPU
        w2_launch<9, 4, true>(a, s);
    } else if (p.KN == 2) {
        FS_W2_GEO(9, 2, true, 48, 48, 2, 12, 17, 4)       // first stride-2 conv (16 -> 32)
        w2_launch<9, 2, true>(a, s);
    } else {
        FS_W2_GEO(18, 1, true, 16, 16, 6, 16, 29, 4)      // kw-folded output layer (9x2 taps, 16 -> 15), batch 32: 16 x 24 tiles
        FS_W2_GEO(18, 1, true, 16, 16, 4, 16, 21, 4)      // ... batch 4 per GPU: 16 x 16 tiles
        w2_launch<18, 1, true>(a, s);
    }
#undef FS_W2_GEO
    if (prof) prof->end(s);
    if (hipGetLastError() != hipSuccess) return -3;
    if (defer && defer->n + r.n <= kW2MaxProb && (defer->n == 0 || defer->scale == scale)) {
        defer->scale = scale;
        for (int i = 0; i < r.n; ++i) defer->job[defer->n++] = r.job[i];
        return 0;
    }
    (void)max4;
    return wgrad2_reduce(r, s);
}

}  // namespace fs
