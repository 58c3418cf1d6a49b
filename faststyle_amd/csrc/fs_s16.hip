// Streaming convolution with SIXTEEN output channels: the layers next to the image (reference im_transf_net.py:37, 69-70) --
//   * the 9x9 image layer 3 -> 16 (REFLECT-40 fused) and the input gradient of the 9x9 output layer (3 -> 16, zero padding);
//   * the kw-folded output layer 16 -> 16 "virtual" channels (9x2 taps, horizontal spacing 5; fs_fold.hip).
//
// Round 1-2 ran them through conv_igemm_kernel<16,4,1>: one tile per workgroup (load, wait, multiply, store), the whole filter
// re-staged through LDS for every tile, and an inner loop in which the backend shuffles the accumulators through
// v_accvgpr_read / mov / write every iteration (two register sets behind a conditional prefetch).  This is the streaming
// recipe of fs_cstream.hip for v_mfma_f32_16x16x4_f32 (16 pixels x 16 channels x 4 k):
//   * persistent workgroups (two per CU: ~150 registers per lane) walk a strided list of 16x16-pixel tiles; ONE patch stage:
//     the loads of tile t+1 travel in registers during the sweep of tile t;
//   * the whole filter lives in REGISTERS as B fragments (63 / 72 registers), no filter traffic through LDS;
//   * the sweep is one straight line of 252 / 288 matrix instructions per wave; every A operand is ONE lane-constant base
//     register + an immediate offset, read one step ahead and pinned into the slot of the previous step's matrix instruction
//     (tools/mfma16_slots.hip: beside a 32-cycle 16x16x4 instruction an immediate-offset ds_read_b32 is free, every vector-ALU
//     instruction costs 13+ cycles);
//   * K runs over (kw, ci) contiguously per kernel row for the 3-channel inputs (27 -> 28: 7 k-steps per row, the 28th
//     multiplies a zero of the filter), over the 16 channels of a tap for the folded layer (LDS pixel pitch 17: the 16 pixels
//     of an A fragment fall into distinct banks);
//   * epilogue options these layers use: per-tile instance-norm records {mean, M2, count}, plain stores.
#include "fs_kernels.h"

#include <type_traits>

namespace fs {

namespace {
constexpr unsigned kOOB = 0x80000000u;
constexpr int kT = 16;   // tile side
}  // namespace

// NB: 16-channel output blocks (Cout = 16 NB).  NB = 4 is VGG16's conv1_1 (3 -> 64, 3x3, image mean folded into the load,
// bias + ReLU): every A fragment feeds four matrix instructions.
template <int CIN, int KH, int KW, int DILX, int NB = 1>
__global__ __launch_bounds__(256, 2) void conv_s16_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    const ConvPlan& p = a.p;
    constexpr bool C3 = CIN == 3;
    constexpr int COUT = 16 * NB;
    constexpr int S = C3 ? 3 : CIN + 1;                                   // LDS floats per patch pixel
    constexpr int PH = kT - 1 + KH, PW = kT - 1 + (KW - 1) * DILX + 1, NPX = PH * PW;
    constexpr int SPR = C3 ? (KW * 3 + 3) / 4 : CIN / 4;                  // k-steps per kernel row (C3) / per tap
    constexpr int KSTEPS = (C3 ? KH : KH * KW) * SPR;
    constexpr int C4 = C3 ? 1 : CIN / 4;                                  // staged elements per pixel (a pixel's 3 floats / float4s)
    constexpr int NE = NPX * C4, SX = (NE + 255) / 256;
    constexpr int PATCH_F = (NPX * S + 8 + 3) & ~3;                       // + slack: the sink of unowned elements, the k = 27 overrun
    constexpr int REDF = 4 * 3 * 16;                                      // one statistics buffer: [wave][s1, s2, shift][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, k4 = lane >> 4;
    float* const red = smem + PATCH_F;   // [2][REDF]
    auto fdiv = [](int x, float inv_d) __attribute__((always_inline)) { return (int)(((float)x + 0.5f) * inv_d); };   // exact for x < 2^22
    auto uniform_ptr = [](const float* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };

    // ---- the filter, once, into registers: lane (channel m16, k4) holds row 4 j + k4 of k-step j
    float breg[KSTEPS][NB];
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (C3) {
                const int kh = j / SPR, k = 4 * (j % SPR) + k4;   // k = kw * 3 + ci
                breg[j][nb] = k < KW * 3 ? a.w[(kh * KW * 3 + k) * COUT + nb * 16 + m16] : 0.f;
            } else {
                const int tap = j / SPR, ci = 4 * (j % SPR) + k4;
                breg[j][nb] = a.w[(tap * CIN + ci) * COUT + nb * 16 + m16];
            }
        }
    // ---- A operands: block m of this wave is tile row 4 wave + m, lane (m16, k4) feeds pixel column m16, row k4 of the k-step
    const int laneA = ((4 * wave) * PW + m16) * S + k4;
    auto aoff = [](int j, int m) __attribute__((always_inline)) {   // compile-time float offset of k-step j, block m
        if (C3) return ((j / SPR) * PW + m * PW) * S + 4 * (j % SPR);
        const int tap = j / SPR;
        return ((tap / KW + m) * PW + (tap % KW) * DILX) * S + 4 * (j % SPR);
    };

    // ---- staging descriptors (tile-invariant): element e = tid + i*256 is float4 c4 of patch pixel e / C4 (C3: the pixel's 3 floats)
    const int c4 = C3 ? 0 : (tid & (C4 - 1));
    int pq[SX], pdst[SX];
    unsigned poffb[SX];
#pragma unroll
    for (int i = 0; i < SX; ++i) {
        const int e = tid + i * 256;
        pq[i] = -1;
        pdst[i] = NPX * S;   // slack
        poffb[i] = kOOB;
        if (e < NE) {
            const int pix = C3 ? e : e / C4;
            const int py = fdiv(pix, 1.0f / (float)PW), px = pix - py * PW;
            pq[i] = (py << 8) | px;
            pdst[i] = pix * S + c4 * 4;
            poffb[i] = (unsigned)((py * a.W + px) * CIN + c4 * 4) * 4u;
        }
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * CIN) * 4u);
    const bool has_ab = !C3 && a.in_a != nullptr;
    const bool has_ab3 = C3 && a.in_a != nullptr;   // per-channel affine of a 3-channel input (VGG: image - mean)
    const bool in_relu = !C3 && a.in_relu != 0;
    const int refl = a.src_mode == SRC_REFLECT ? a.refl : 0;

    // ---- items: tile lin = blockIdx.x + it * gridDim.x over (sample, tile row, tile column)
    const int tiles = p.tiles_y * p.tiles_x;
    const int total = a.N * tiles;
    const int GX = (int)gridDim.x;
    const int my_items = ((int)blockIdx.x < total) ? (total - 1 - (int)blockIdx.x) / GX + 1 : 0;
    const float inv_tiles = 1.0f / (float)tiles, inv_tx = 1.0f / (float)p.tiles_x;
    struct Item {
        int n, ty0, tx0, lin;
    };
    auto decode = [&](int it) __attribute__((always_inline)) {
        Item r;
        r.lin = (int)blockIdx.x + it * GX;
        r.n = fdiv(r.lin, inv_tiles);
        const int tr = r.lin - r.n * tiles;
        const int tyi = fdiv(tr, inv_tx);
        r.ty0 = tyi * kT;
        r.tx0 = (tr - tyi * p.tiles_x) * kT;
        r.lin = __builtin_amdgcn_readfirstlane(r.lin);
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.ty0 = __builtin_amdgcn_readfirstlane(r.ty0);
        r.tx0 = __builtin_amdgcn_readfirstlane(r.tx0);
        return r;
    };
    float pv[SX][4];
    unsigned pok = 0;   // bit i: element i came from inside the image; bit 31: the whole patch did (interior tile)
    float4 va = make_float4(1.f, 1.f, 1.f, 1.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
    auto issue = [&](const Item& I) __attribute__((always_inline)) {
        // (vy0, vx0): the patch's first pixel in the coordinates of the (mirror-padded) source
        const int vy0 = I.ty0 - a.pad_t, vx0 = I.tx0 - a.pad_l;
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * CIN);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, x_bytes, 0x00020000);
        const int iy0 = vy0 - refl, ix0 = vx0 - refl;   // ... in the coordinates of the stored image
        if (iy0 >= 0 && ix0 >= 0 && iy0 + PH <= a.H && ix0 + PW <= a.W) {
            // interior tile (the vast majority): tile-invariant per-thread offsets, the origin rides in the scalar offset operand
            pok = 0xFFFFFFFFu;
            const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)((iy0 * a.W + ix0) * CIN) * 4u);
#pragma unroll
            for (int i = 0; i < SX; ++i) {
                if (C3) {
                    const auto t = __builtin_amdgcn_raw_buffer_load_b96(xr, poffb[i], base, 0);
                    __builtin_memcpy(pv[i], &t, 12);
                } else {
                    const auto t = __builtin_amdgcn_raw_buffer_load_b128(xr, poffb[i], base, 0);
                    __builtin_memcpy(pv[i], &t, 16);
                }
            }
        } else {
            pok = 0;
#pragma unroll
            for (int i = 0; i < SX; ++i) {
                const int vy = vy0 + (pq[i] >> 8), vx = vx0 + (pq[i] & 255);
                // zero padding outside the (mirror-padded) source; inside it, mirror padding by `refl` pixels (tf.pad REFLECT)
                const bool ok = pq[i] >= 0 && (unsigned)vy < (unsigned)(a.H + 2 * refl) && (unsigned)vx < (unsigned)(a.W + 2 * refl);
                int sy = vy - refl, sx = vx - refl;
                sy = sy < 0 ? -sy : sy;
                sx = sx < 0 ? -sx : sx;
                sy = sy >= a.H ? 2 * (a.H - 1) - sy : sy;
                sx = sx >= a.W ? 2 * (a.W - 1) - sx : sx;
                pok |= ok ? (1u << i) : 0u;
                const unsigned vo = ok ? (unsigned)((sy * a.W + sx) * CIN + c4 * 4) * 4u : kOOB;
                if (C3) {
                    const auto t = __builtin_amdgcn_raw_buffer_load_b96(xr, vo, 0, 0);
                    __builtin_memcpy(pv[i], &t, 12);
                } else {
                    const auto t = __builtin_amdgcn_raw_buffer_load_b128(xr, vo, 0, 0);
                    __builtin_memcpy(pv[i], &t, 16);
                }
            }
        }
        if (has_ab) {
            va = *reinterpret_cast<const float4*>(a.in_a + (size_t)I.n * a.in_nstride + c4 * 4);
            vb = *reinterpret_cast<const float4*>(a.in_b + (size_t)I.n * a.in_nstride + c4 * 4);
        }
        if (has_ab3) {
            const float* pa = a.in_a + (size_t)I.n * a.in_nstride;
            const float* pb = a.in_b + (size_t)I.n * a.in_nstride;
            va = make_float4(pa[0], pa[1], pa[2], 0.f);
            vb = make_float4(pb[0], pb[1], pb[2], 0.f);
        }
    };
    auto relu1 = [](float x) __attribute__((always_inline)) {   // ONE v_max_f32
#if defined(__HIP_DEVICE_COMPILE__)
        float r;
        asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
        return r;
#else
        return x > 0.f ? x : 0.f;
#endif
    };
    auto commit_as = [&](auto MASKED) __attribute__((always_inline)) {
        constexpr bool masked = decltype(MASKED)::value;
#pragma unroll
        for (int i = 0; i < SX; ++i) {
            float* d = smem + pdst[i];
            if (C3) {
                float v0 = pv[i][0], v1 = pv[i][1], v2 = pv[i][2];
                if (has_ab3) {   // padding arrives as 0 and must stay 0 (tf.nn.conv2d pads the mean-subtracted image with zeros)
                    const unsigned okm = (!masked || ((pok >> i) & 1u)) ? 0xFFFFFFFFu : 0u;
                    v0 = fmaf(v0, va.x, __uint_as_float(__float_as_uint(vb.x) & okm));
                    v1 = fmaf(v1, va.y, __uint_as_float(__float_as_uint(vb.y) & okm));
                    v2 = fmaf(v2, va.z, __uint_as_float(__float_as_uint(vb.z) & okm));
                }
                d[0] = v0;
                d[1] = v1;
                d[2] = v2;
            } else {
                float v[4] = {pv[i][0], pv[i][1], pv[i][2], pv[i][3]};
                if (has_ab) {   // producer instance norm folded into the load; padding arrives as 0 and must stay 0
                    const unsigned okm = (!masked || ((pok >> i) & 1u)) ? 0xFFFFFFFFu : 0u;
                    v[0] = fmaf(v[0], va.x, __uint_as_float(__float_as_uint(vb.x) & okm));
                    v[1] = fmaf(v[1], va.y, __uint_as_float(__float_as_uint(vb.y) & okm));
                    v[2] = fmaf(v[2], va.z, __uint_as_float(__float_as_uint(vb.z) & okm));
                    v[3] = fmaf(v[3], va.w, __uint_as_float(__float_as_uint(vb.w) & okm));
                }
                if (in_relu) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = relu1(v[c]);
                }
                d[0] = v[0];
                d[1] = v[1];
                d[2] = v[2];
                d[3] = v[3];
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
        if (pok >> 31)   // (wave-uniform) interior tile: no masking
            commit_as(std::false_type{});
        else
            commit_as(std::true_type{});
    };

    f32x4 acc[4][NB];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[m][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    // one straight line: matrix instruction (j, m) with the read of (j + 1, m) in its slot
    auto sweep = [&]() __attribute__((always_inline)) {
        float av[2][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) av[0][m] = smem[laneA + aoff(0, m)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (j + 1 < KSTEPS) av[(j + 1) & 1][m] = smem[laneA + aoff(j + 1, m)];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j & 1][m], breg[j][nb], acc[m][nb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };

    // ---- epilogue of one item: accumulator register r of block m, lane (m16, k4) = pixel (row 4 wave + m, column 4 k4 + r), channel m16
    const unsigned y_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * COUT) * 4u);
    float bias[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bias[nb] = a.bias ? a.bias[nb * 16 + m16] : 0.f;
    const bool out_relu = a.out_relu != 0;
    auto epilogue = [&](const Item& I, float* rbuf) __attribute__((always_inline)) {
        const int th_valid = min(kT, a.Ho - I.ty0), tw_valid = min(kT, a.Wo - I.tx0);
        if (NB == 1 && a.stats) {
            // per-WAVE partial sums of (x - c), (x - c)^2 over the wave's four tile rows, c = the wave's own first pixel of the
            // channel; the four records of a tile are merged when the next pipeline step starts (finalize below): no barrier here
            const float cs = __shfl(acc[0][0][0], m16);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = 4 * wave + m < th_valid && 4 * k4 + r < tw_valid;
                    const float d = ok ? acc[m][0][r] - cs : 0.f;
                    s1 += d;
                    s2 = fmaf(d, d, s2);
                }
            s1 += __shfl_xor(s1, 16);
            s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lane < 16) {
                rbuf[(wave * 3 + 0) * 16 + lane] = s1;
                rbuf[(wave * 3 + 1) * 16 + lane] = s2;
                rbuf[(wave * 3 + 2) * 16 + lane] = cs;
            }
        }
        float* yn = a.y + (size_t)I.n * a.Ho * a.Wo * COUT;
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yn)), 0, y_bytes, 0x00020000);
        const int lane_off = ((I.tx0 + 4 * k4) * COUT + m16) * 4;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = 4 * wave + m;
            const int row_off = (I.ty0 + row) * a.Wo * COUT * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = row < th_valid && 4 * k4 + r < tw_valid;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float v = acc[m][nb][r] + bias[nb];
                    if (out_relu) v = relu1(v);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, ok ? (unsigned)(lane_off + row_off + r * COUT * 4 + nb * 64) : kOOB, 0, 0);
                }
            }
        }
        zero_acc();
    };
    // merge of the four per-wave records of one tile (Chan's update, fixed order) -> {mean, M2, count} of the tile
    auto finalize = [&](const Item& I, const float* rbuf) __attribute__((always_inline)) {
        const int th_valid = min(kT, a.Ho - I.ty0), tw_valid = min(kT, a.Wo - I.tx0);
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int rows = min(4, max(0, th_valid - 4 * w));
            const float cb = (float)(rows * tw_valid);
            if (cb > 0.f) {
                const float S1 = rbuf[(w * 3 + 0) * 16 + tid], S2 = rbuf[(w * 3 + 1) * 16 + tid], sh = rbuf[(w * 3 + 2) * 16 + tid];
                const float mb = sh + S1 / cb, qb2 = fmaxf(S2 - S1 * S1 / cb, 0.f);
                const float nn_ = cnt + cb, d = mb - mean, rr = cb / nn_;
                mean += d * rr;
                m2 += qb2 + d * d * cnt * rr;
                cnt = nn_;
            }
        }
        float* st = a.stats + ((size_t)I.lin * 16 + tid) * 3;
        st[0] = mean;
        st[1] = m2;
        st[2] = cnt;
    };

    // ---- the pipeline (fs_cstream.hip): ONE patch stage.  While tile t is multiplied the loads of tile t+1 are in flight
    // (registers); after the sweep (barrier A) they are committed over the patch, the epilogue of tile t follows (its stores
    // drain during the next sweep), barrier B, next tile.
    if (my_items == 0) return;
    if (tid < 8) smem[NPX * S + tid] = 0.f;   // the slack is read by the k = 27 overrun of the last patch pixel (times a zero weight)
    Item cur = decode(0), prev = cur;
    issue(cur);
    commit();
    FS_TOUCH_F4(va);   // (... and on the path into the loop)
    FS_TOUCH_F4(vb);
    __syncthreads();
    for (int it = 0; it < my_items; ++it) {
        const bool more = it + 1 < my_items;
        if (NB == 1 && it > 0 && a.stats && tid < 16) finalize(prev, red + ((it - 1) & 1) * REDF);
        Item nxt = cur;
        if (more) {
            nxt = decode(it + 1);
            issue(nxt);
        }
        sweep();
        FS_LDS_BARRIER();   // A: every wave is done reading the patch
        if (more) commit();
        FS_TOUCH_F4(va);   // (the conditional affine loads of `issue` are known complete at the loop header: fs_kernels.h -- the next issue phase waited
        FS_TOUCH_F4(vb);   //  vmcnt(0) for its own patch loads AND the previous tile's stores before the sweep)
        epilogue(cur, red + (it & 1) * REDF);
        FS_LDS_BARRIER();   // B: next patch and this tile's statistics records (LDS) visible; the stores drain during the next sweep
        prev = cur;
        cur = nxt;
    }
    if (NB == 1 && a.stats && tid < 16) finalize(prev, red + ((my_items - 1) & 1) * REDF);
}


// ---------------------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the kw-folded output layer (16 -> 16 virtual channels, 9 x 2 taps, spacing 5) with its products on the bf16 matrix cores as SIX EXACT
// products of bf16 pieces (the arithmetic of fs_wino6.hip: x = h + m + l exactly, three bf16 pieces by truncation; hh, hm, mh, hl, lh, mm issued, the
// three terms below 2^-24 of the leading one dropped; fp32 accumulation, the five small products in an accumulator of their own).
// v_mfma_f32_16x16x32_bf16 multiplies 16 pixels x 16 channels x 32 k in 16 cycles where eight v_mfma_f32_16x16x4_f32 take 256: six products cost 96.
// And unlike the fp32 matrix instruction (which shares the fp32 lanes with the vector ALU) it runs beside the other resident workgroup's commit /
// epilogue arithmetic.  What it costs: the commit splits every staged value (4 + 1.5 vector instructions per element on top of the affine + ReLU),
// the patch holds three bf16 pieces (6 bytes per element instead of 4) and the filter three pieces in registers (108 instead of 72).
//   * one k-step = the two taps of a kernel row x 16 channels: lane (pixel m16, k group kg) reads channels 8 (kg & 1) .. + 7 of tap kw = kg >> 1 -- the
//     tap's column offset is part of the lane's base address, every operand read is lane base + immediate, one ds_read_b128 per piece;
//   * LDS patch [half = channel / 8][pixel][piece][8 channels] bf16: 48 bytes per pixel and half (an odd number of 16-byte slots: the 16 pixels of a
//     fragment fall on 16 distinct slots), the second half a multiple of 256 bytes behind the first (the two halves of a ds_read_b128 lane group hit
//     the same slots a whole bank row apart: conflict-free);
//   * the pipeline, the statistics records and the stores are conv_s16_kernel's.
typedef __bf16 s16_bf16x8 __attribute__((ext_vector_type(8)));
#ifndef FS_S16X_ABL
#define FS_S16X_ABL 0   /* timing experiments (results wrong): 1 no sweep, 2 no split (h only), 4 no stores */
#endif
__host__ __device__ __forceinline__ void s16x_split(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    h = u & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, h);
    m = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, m);
    l = __builtin_bit_cast(unsigned, r2);   // (<= 8 significant bits: its low half is zero)
}
template <int KH, int DILX>
__global__ __launch_bounds__(256, 2) void conv_s16x_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    char* const lds = reinterpret_cast<char*>(smem);
    const ConvPlan& p = a.p;
    constexpr int CIN = 16, COUT = 16, KW = 2;
    constexpr int PH = kT - 1 + KH, PW = kT - 1 + (KW - 1) * DILX + 1, NPX = PH * PW;
    constexpr int PXB = 48;                                               // bytes per pixel and half: [piece 3][8 channels bf16]
    constexpr int HPB = ((NPX + 1) * PXB + 255) & ~255;                   // half plane (+ one pixel: the sink of unowned elements), a multiple of 256 bytes
    constexpr int KSTEPS = KH;                                            // one kernel row = two taps x 16 channels = 32 k
    constexpr int NE = NPX * 4, SX = (NE + 255) / 256;
    constexpr int REDF = 4 * 3 * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, kg = lane >> 4;
    float* const red = reinterpret_cast<float*>(lds + 2 * HPB);   // [2][REDF]
    auto fdiv = [](int x, float inv_d) __attribute__((always_inline)) { return (int)(((float)x + 0.5f) * inv_d); };
    auto uniform_ptr = [](const float* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };

    // ---- the filter, once, into registers as three pieces: lane (channel m16, kg) holds k = 8 kg .. + 7 of k-step j = (tap 2 j + (kg >> 1), ci 8 (kg & 1) ..)
    s16_bf16x8 breg[KSTEPS][3];
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) {
        unsigned hh[8], mm[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s16x_split(a.w[((2 * j + (kg >> 1)) * CIN + 8 * (kg & 1) + e) * COUT + m16], hh[e], mm[e], ll[e]);
        const uint4 H = make_uint4((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3], (hh[4] >> 16) | hh[5], (hh[6] >> 16) | hh[7]);
        const uint4 M = make_uint4((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3], (mm[4] >> 16) | mm[5], (mm[6] >> 16) | mm[7]);
        const uint4 L = make_uint4((ll[0] >> 16) | (ll[1] & 0xffff0000u), (ll[2] >> 16) | (ll[3] & 0xffff0000u), (ll[4] >> 16) | (ll[5] & 0xffff0000u), (ll[6] >> 16) | (ll[7] & 0xffff0000u));
        breg[j][0] = __builtin_bit_cast(s16_bf16x8, H);
        breg[j][1] = __builtin_bit_cast(s16_bf16x8, M);
        breg[j][2] = __builtin_bit_cast(s16_bf16x8, L);
    }
    // ---- A operands: block m of this wave is tile row 4 wave + m; lane (m16, kg) feeds pixel column m16 with tap column kg >> 1, channel half kg & 1
    const int laneA = (kg & 1) * HPB + ((4 * wave) * PW + m16 + (kg >> 1) * DILX) * PXB;
    auto aoff = [](int j, int m, int pc) __attribute__((always_inline)) { return (j + m) * PW * PXB + pc * 16; };   // compile-time byte offset

    // ---- staging descriptors (tile-invariant): element e = tid + i*256 is float4 c4 of patch pixel e / 4
    const int c4 = tid & 3;
    int pq[SX], pdst[SX];
    unsigned poffb[SX];
#pragma unroll
    for (int i = 0; i < SX; ++i) {
        const int e = tid + i * 256;
        pq[i] = -1;
        pdst[i] = NPX * PXB;   // the sink pixel
        poffb[i] = kOOB;
        if (e < NE) {
            const int pix = e >> 2;
            const int py = fdiv(pix, 1.0f / (float)PW), px = pix - py * PW;
            pq[i] = (py << 8) | px;
            pdst[i] = (c4 >> 1) * HPB + pix * PXB + (c4 & 1) * 8;
            poffb[i] = (unsigned)((py * a.W + px) * CIN + c4 * 4) * 4u;
        }
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * CIN) * 4u);
    const bool has_ab = a.in_a != nullptr;
    const bool in_relu = a.in_relu != 0;

    const int tiles = p.tiles_y * p.tiles_x;
    const int total = a.N * tiles;
    const int GX = (int)gridDim.x;
    const int my_items = ((int)blockIdx.x < total) ? (total - 1 - (int)blockIdx.x) / GX + 1 : 0;
    const float inv_tiles = 1.0f / (float)tiles, inv_tx = 1.0f / (float)p.tiles_x;
    struct Item {
        int n, ty0, tx0, lin;
    };
    auto decode = [&](int it) __attribute__((always_inline)) {
        Item r;
        r.lin = (int)blockIdx.x + it * GX;
        r.n = fdiv(r.lin, inv_tiles);
        const int tr = r.lin - r.n * tiles;
        const int tyi = fdiv(tr, inv_tx);
        r.ty0 = tyi * kT;
        r.tx0 = (tr - tyi * p.tiles_x) * kT;
        r.lin = __builtin_amdgcn_readfirstlane(r.lin);
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.ty0 = __builtin_amdgcn_readfirstlane(r.ty0);
        r.tx0 = __builtin_amdgcn_readfirstlane(r.tx0);
        return r;
    };
    float4 pv[SX];
    unsigned pok = 0;   // bit i: element i came from inside the image; bit 31: the whole patch did (interior tile)
    float4 va = make_float4(1.f, 1.f, 1.f, 1.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
    auto issue = [&](const Item& I) __attribute__((always_inline)) {
        const int vy0 = I.ty0 - a.pad_t, vx0 = I.tx0 - a.pad_l;
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * CIN);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, x_bytes, 0x00020000);
        if (vy0 >= 0 && vx0 >= 0 && vy0 + PH <= a.H && vx0 + PW <= a.W) {
            pok = 0xFFFFFFFFu;
            const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)((vy0 * a.W + vx0) * CIN) * 4u);
#pragma unroll
            for (int i = 0; i < SX; ++i) pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, poffb[i], base, 0));
        } else {
            pok = 0;
#pragma unroll
            for (int i = 0; i < SX; ++i) {
                const int sy = vy0 + (pq[i] >> 8), sx = vx0 + (pq[i] & 255);
                const bool ok = pq[i] >= 0 && (unsigned)sy < (unsigned)a.H && (unsigned)sx < (unsigned)a.W;
                pok |= ok ? (1u << i) : 0u;
                pv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((sy * a.W + sx) * CIN + c4 * 4) * 4u : kOOB, 0, 0));
            }
        }
        if (has_ab) {
            va = *reinterpret_cast<const float4*>(a.in_a + (size_t)I.n * a.in_nstride + c4 * 4);
            vb = *reinterpret_cast<const float4*>(a.in_b + (size_t)I.n * a.in_nstride + c4 * 4);
        }
    };
    auto relu1 = [](float x) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        float r;
        asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
        return r;
#else
        return x > 0.f ? x : 0.f;
#endif
    };
    auto commit_as = [&](auto MASKED) __attribute__((always_inline)) {
        constexpr bool masked = decltype(MASKED)::value;
#pragma unroll
        for (int i = 0; i < SX; ++i) {
            float v[4] = {pv[i].x, pv[i].y, pv[i].z, pv[i].w};
            if (has_ab) {   // producer instance norm folded into the load; padding arrives as 0 and must stay 0
                const unsigned okm = (!masked || ((pok >> i) & 1u)) ? 0xFFFFFFFFu : 0u;
                v[0] = fmaf(v[0], va.x, __uint_as_float(__float_as_uint(vb.x) & okm));
                v[1] = fmaf(v[1], va.y, __uint_as_float(__float_as_uint(vb.y) & okm));
                v[2] = fmaf(v[2], va.z, __uint_as_float(__float_as_uint(vb.z) & okm));
                v[3] = fmaf(v[3], va.w, __uint_as_float(__float_as_uint(vb.w) & okm));
            }
            if (in_relu) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = relu1(v[c]);
            }
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (FS_S16X_ABL & 2) {
                    h[c] = __float_as_uint(v[c]) & 0xffff0000u;
                    m[c] = l[c] = 0u;
                } else {
                    s16x_split(v[c], h[c], m[c], l[c]);
                }
            }
            char* d = lds + pdst[i];
            *reinterpret_cast<uint2*>(d) = make_uint2((h[0] >> 16) | h[1], (h[2] >> 16) | h[3]);
            *reinterpret_cast<uint2*>(d + 16) = make_uint2((m[0] >> 16) | m[1], (m[2] >> 16) | m[3]);
            *reinterpret_cast<uint2*>(d + 32) = make_uint2((l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u));
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
        if (pok >> 31)
            commit_as(std::false_type{});
        else
            commit_as(std::true_type{});
    };

    f32x4 acc[4], acs[4];   // the leading product / the five small ones
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = acs[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    // one straight line: the six matrix instructions of (j, m) with the three operand reads of the next (j, m) in front of them
    auto sweep = [&]() __attribute__((always_inline)) {
        s16_bf16x8 av[2][3];
        auto rd = [&](int u, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) av[buf][pc] = __builtin_bit_cast(s16_bf16x8, *reinterpret_cast<const uint4*>(lds + laneA + aoff(u >> 2, u & 3, pc)));
        };
        rd(0, 0);
#pragma unroll
        for (int u = 0; u < KSTEPS * 4; ++u) {
            const int j = u >> 2, m = u & 3, b = u & 1;
            if (u + 1 < KSTEPS * 4) rd(u + 1, b ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            // smallest first: h l, l h, m m, h m, m h into the small accumulator, h h into the leading one
            acs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][0], breg[j][2], acs[m], 0, 0, 0);
            acs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][2], breg[j][0], acs[m], 0, 0, 0);
            acs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][1], breg[j][1], acs[m], 0, 0, 0);
            acs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][0], breg[j][1], acs[m], 0, 0, 0);
            acs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][1], breg[j][0], acs[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][0], breg[j][0], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] += acs[m];
    };

    // ---- epilogue of one item: accumulator register r of block m, lane (m16, kg) = pixel (row 4 wave + m, column 4 kg + r), channel m16
    const unsigned y_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * COUT) * 4u);
    auto epilogue = [&](const Item& I, float* rbuf) __attribute__((always_inline)) {
        const int th_valid = min(kT, a.Ho - I.ty0), tw_valid = min(kT, a.Wo - I.tx0);
        if (a.stats) {
            const float cs = __shfl(acc[0][0], m16);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = 4 * wave + m < th_valid && 4 * kg + r < tw_valid;
                    const float d = ok ? acc[m][r] - cs : 0.f;
                    s1 += d;
                    s2 = fmaf(d, d, s2);
                }
            s1 += __shfl_xor(s1, 16);
            s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lane < 16) {
                rbuf[(wave * 3 + 0) * 16 + lane] = s1;
                rbuf[(wave * 3 + 1) * 16 + lane] = s2;
                rbuf[(wave * 3 + 2) * 16 + lane] = cs;
            }
        }
        float* yn = a.y + (size_t)I.n * a.Ho * a.Wo * COUT;
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yn)), 0, y_bytes, 0x00020000);
        const int lane_off = ((I.tx0 + 4 * kg) * COUT + m16) * 4;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = 4 * wave + m;
            const int row_off = (I.ty0 + row) * a.Wo * COUT * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = row < th_valid && 4 * kg + r < tw_valid;
                if (!(FS_S16X_ABL & 4) || acc[m][r] == 12345.678f)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[m][r]), yr, ok ? (unsigned)(lane_off + row_off + r * COUT * 4) : kOOB, 0, 0);
            }
        }
        zero_acc();
    };
    auto finalize = [&](const Item& I, const float* rbuf) __attribute__((always_inline)) {
        const int th_valid = min(kT, a.Ho - I.ty0), tw_valid = min(kT, a.Wo - I.tx0);
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int rows = min(4, max(0, th_valid - 4 * w));
            const float cb = (float)(rows * tw_valid);
            if (cb > 0.f) {
                const float S1 = rbuf[(w * 3 + 0) * 16 + tid], S2 = rbuf[(w * 3 + 1) * 16 + tid], sh = rbuf[(w * 3 + 2) * 16 + tid];
                const float mb = sh + S1 / cb, qb2 = fmaxf(S2 - S1 * S1 / cb, 0.f);
                const float nn_ = cnt + cb, d = mb - mean, rr = cb / nn_;
                mean += d * rr;
                m2 += qb2 + d * d * cnt * rr;
                cnt = nn_;
            }
        }
        float* st = a.stats + ((size_t)I.lin * 16 + tid) * 3;
        st[0] = mean;
        st[1] = m2;
        st[2] = cnt;
    };

    if (my_items == 0) return;
    Item cur = decode(0), prev = cur;
    issue(cur);
    commit();
    FS_TOUCH_F4(va);
    FS_TOUCH_F4(vb);
    __syncthreads();
    for (int it = 0; it < my_items; ++it) {
        const bool more = it + 1 < my_items;
        if (it > 0 && a.stats && tid < 16) finalize(prev, red + ((it - 1) & 1) * REDF);
        Item nxt = cur;
        if (more) {
            nxt = decode(it + 1);
            issue(nxt);
        }
        if (!(FS_S16X_ABL & 1)) sweep();
        FS_LDS_BARRIER();   // A: every wave is done reading the patch
        if (more) commit();
        FS_TOUCH_F4(va);
        FS_TOUCH_F4(vb);
        epilogue(cur, red + (it & 1) * REDF);
        FS_LDS_BARRIER();   // B: next patch and this tile's statistics records (LDS) visible; the stores drain during the next sweep
        prev = cur;
        cur = nxt;
    }
    if (a.stats && tid < 16) finalize(prev, red + ((my_items - 1) & 1) * REDF);
}


// The 9 x 9 layers with THREE input channels (the image layer 3 -> 16 with REFLECT-40 fused; the input gradient of the output layer, zero padding) in
// the same arithmetic.  A k-step of v_mfma_f32_16x16x32_bf16 takes 8 consecutive k per lane, i.e. 16 bytes of one LDS address: with 3-channel pixels a
// kernel row's 27 contiguous values start at 6 x bytes -- no alignment --, so
//   * the patch pads a pixel to FOUR channels (8 bytes per piece; the fourth is zero) and a lane's 8 values are a PAIR of horizontally adjacent taps;
//     k runs over the 45 groups (kernel row kh, tap pair pr: taps 2 pr, 2 pr + 1; the tenth tap has zero weights) = 12 k-steps of 4 groups (63 % of the
//     matrix work is useful, against 96 % of the fp32 form's: 72 x 6 x 16 = 1152 cycles per 16 x 16 block against 63 x 32 = 2016);
//   * every piece plane is stored TWICE, the second copy one pixel further: a pair that starts at an odd column is 16-byte aligned there.  The lane picks
//     the copy by the parity of its own pixel column (patch width, pair offsets and row offsets are even), the copies lie 7 slots apart modulo 16:
//     one ds_read_b128 per piece, conflict-free inside a kernel row;
//   * a group's offset (kh PW + 2 pr) depends on the lane's k group: twelve per-lane address registers, the block row and the piece are immediates;
//   * the filter as B fragments would be 144 registers: it lives in LDS ([step][piece][lane] x 16 bytes = 36 KB, written once per workgroup), one
//     ds_read_b128 per piece and k-step shared by the four block rows.  ~150 registers: two workgroups per CU.
template <int KH, int KW>
__global__ __launch_bounds__(256, 2) void conv_s16c3x_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    char* const lds = reinterpret_cast<char*>(smem);
    const ConvPlan& p = a.p;
    constexpr int CIN = 3, COUT = 16;
    constexpr int PH = kT - 1 + KH, PW = kT - 1 + KW, NPX = PH * PW;
    static_assert(!(PW & 1), "even patch width: the copy a lane reads is fixed by the parity of its column");
    constexpr int NPR = (KW + 1) / 2, NG = KH * NPR, KST = (NG + 3) / 4;   // tap pairs per kernel row, groups, k-steps
    constexpr int PL0 = (((NPX + 2) * 8) + 255) & ~255;                    // one piece plane of one copy
    constexpr int C1B = 3 * PL0 + 7 * 16;                                  // copy 1 (pixel i at (i + 1) * 8): 7 slots further modulo 16
    constexpr int FB = (C1B + 3 * PL0 + 255) & ~255;                       // the filter: [KST][3][64 lanes] x 16 bytes
    constexpr int RB = FB + KST * 3 * 64 * 16;                             // statistics records
    constexpr int NE = NPX, SX = (NE + 255) / 256;
    constexpr int REDF = 4 * 3 * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, kg = lane >> 4;
    float* const red = reinterpret_cast<float*>(lds + RB);   // [2][REDF]
    auto fdiv = [](int x, float inv_d) __attribute__((always_inline)) { return (int)(((float)x + 0.5f) * inv_d); };
    auto uniform_ptr = [](const float* ptr) __attribute__((always_inline)) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };

    // ---- the filter, once, into LDS: wave w splits the k-steps w, w + 4, ...; lane (channel m16, kg) of step j holds group q = 4 j + kg = (kh, pr):
    // element e = tap 2 pr + (e >> 2), channel e & 3 (zero for the padded channel, the tenth tap and the groups past the last)
    for (int j = wave; j < KST; j += 4) {
        const int q = 4 * j + kg, kh = q / NPR, pr = q - kh * NPR;
        unsigned hh[8], mm[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kw = 2 * pr + (e >> 2), ci = e & 3;
            const float w = (q < NG && kw < KW && ci < 3) ? a.w[((kh * KW + kw) * CIN + ci) * COUT + m16] : 0.f;
            s16x_split(w, hh[e], mm[e], ll[e]);
        }
        uint4* d = reinterpret_cast<uint4*>(lds + FB + (j * 3 * 64 + lane) * 16);
        d[0] = make_uint4((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3], (hh[4] >> 16) | hh[5], (hh[6] >> 16) | hh[7]);
        d[64] = make_uint4((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3], (mm[4] >> 16) | mm[5], (mm[6] >> 16) | mm[7]);
        d[128] = make_uint4((ll[0] >> 16) | (ll[1] & 0xffff0000u), (ll[2] >> 16) | (ll[3] & 0xffff0000u), (ll[4] >> 16) | (ll[5] & 0xffff0000u), (ll[6] >> 16) | (ll[7] & 0xffff0000u));
    }
    // ---- A operands: block m of this wave is tile row 4 wave + m; lane (pixel column m16, kg) reads the pixel pair of its group from the copy of its parity
    const int cp = m16 & 1;
    int addrA[KST];
#pragma unroll
    for (int j = 0; j < KST; ++j) {
        int q = 4 * j + kg;
        q = q < NG ? q : NG - 1;
        const int kh = q / NPR, pr = q - kh * NPR;
        addrA[j] = cp * C1B + ((4 * wave + kh) * PW + m16 + 2 * pr + cp) * 8;
    }

    // ---- staging descriptors (tile-invariant): element e = tid + i*256 is patch pixel e (its 3 floats)
    int pq[SX], pdst[SX];
    unsigned poffb[SX];
#pragma unroll
    for (int i = 0; i < SX; ++i) {
        const int e = tid + i * 256;
        pq[i] = -1;
        pdst[i] = NPX * 8;   // the sink pixel (read by the tenth tap of the last patch row: times a zero weight)
        poffb[i] = kOOB;
        if (e < NE) {
            const int py = fdiv(e, 1.0f / (float)PW), px = e - py * PW;
            pq[i] = (py << 8) | px;
            pdst[i] = e * 8;
            poffb[i] = (unsigned)((py * a.W + px) * CIN) * 4u;
        }
    }
    const unsigned x_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.H * a.W * CIN) * 4u);
    const bool has_ab3 = a.in_a != nullptr;   // per-channel affine of a 3-channel input
    const int refl = a.src_mode == SRC_REFLECT ? a.refl : 0;

    const int tiles = p.tiles_y * p.tiles_x;
    const int total = a.N * tiles;
    const int GX = (int)gridDim.x;
    const int my_items = ((int)blockIdx.x < total) ? (total - 1 - (int)blockIdx.x) / GX + 1 : 0;
    const float inv_tiles = 1.0f / (float)tiles, inv_tx = 1.0f / (float)p.tiles_x;
    struct Item {
        int n, ty0, tx0, lin;
    };
    auto decode = [&](int it) __attribute__((always_inline)) {
        Item r;
        r.lin = (int)blockIdx.x + it * GX;
        r.n = fdiv(r.lin, inv_tiles);
        const int tr = r.lin - r.n * tiles;
        const int tyi = fdiv(tr, inv_tx);
        r.ty0 = tyi * kT;
        r.tx0 = (tr - tyi * p.tiles_x) * kT;
        r.lin = __builtin_amdgcn_readfirstlane(r.lin);
        r.n = __builtin_amdgcn_readfirstlane(r.n);
        r.ty0 = __builtin_amdgcn_readfirstlane(r.ty0);
        r.tx0 = __builtin_amdgcn_readfirstlane(r.tx0);
        return r;
    };
    // TWO tiles of loads in flight (a pixel is 12 bytes: 9 registers per tile): with the sweep at a third of the fp32 form's length one tile of
    // look-ahead no longer covers the latency of the loads
    struct Stage {
        float pv[SX][3];
        unsigned pok;   // bit i: element i came from inside the image; bit 31: the whole patch did (interior tile)
        float4 va, vb;
    };
    Stage st0, st1;
    st0.pok = st1.pok = 0;
    st0.va = st1.va = make_float4(1.f, 1.f, 1.f, 1.f);
    st0.vb = st1.vb = make_float4(0.f, 0.f, 0.f, 0.f);
    // live = 0: a tile beyond the list -- every load against an empty buffer (zeros, no traffic), so that the number of loads in flight is the same on every path
    auto issue = [&](Stage& T, const Item& I, int live) __attribute__((always_inline)) {
        float (&pv)[SX][3] = T.pv;
        unsigned& pok = T.pok;
        float4 &va = T.va, &vb = T.vb;
        const int vy0 = I.ty0 - a.pad_t, vx0 = I.tx0 - a.pad_l;   // the patch's first pixel in the coordinates of the (mirror-padded) source
        const float* xn = uniform_ptr(a.x + (size_t)I.n * a.H * a.W * CIN);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, live ? x_bytes : 0u, 0x00020000);
        const int iy0 = vy0 - refl, ix0 = vx0 - refl;             // ... in the coordinates of the stored image
        if (iy0 >= 0 && ix0 >= 0 && iy0 + PH <= a.H && ix0 + PW <= a.W) {
            pok = 0xFFFFFFFFu;
            const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)((iy0 * a.W + ix0) * CIN) * 4u);
#pragma unroll
            for (int i = 0; i < SX; ++i) {
                const auto t = __builtin_amdgcn_raw_buffer_load_b96(xr, poffb[i], base, 0);
                __builtin_memcpy(pv[i], &t, 12);
            }
        } else {
            pok = 0;
#pragma unroll
            for (int i = 0; i < SX; ++i) {
                const int vy = vy0 + (pq[i] >> 8), vx = vx0 + (pq[i] & 255);
                const bool ok = pq[i] >= 0 && (unsigned)vy < (unsigned)(a.H + 2 * refl) && (unsigned)vx < (unsigned)(a.W + 2 * refl);
                int sy = vy - refl, sx = vx - refl;
                sy = sy < 0 ? -sy : sy;
                sx = sx < 0 ? -sx : sx;
                sy = sy >= a.H ? 2 * (a.H - 1) - sy : sy;
                sx = sx >= a.W ? 2 * (a.W - 1) - sx : sx;
                pok |= ok ? (1u << i) : 0u;
                const auto t = __builtin_amdgcn_raw_buffer_load_b96(xr, ok ? (unsigned)((sy * a.W + sx) * CIN) * 4u : kOOB, 0, 0);
                __builtin_memcpy(pv[i], &t, 12);
            }
        }
        if (has_ab3) {
            const float* pa = a.in_a + (size_t)I.n * a.in_nstride;
            const float* pb = a.in_b + (size_t)I.n * a.in_nstride;
            va = make_float4(pa[0], pa[1], pa[2], 0.f);
            vb = make_float4(pb[0], pb[1], pb[2], 0.f);
        }
    };
    auto commit_as = [&](Stage& T, auto MASKED) __attribute__((always_inline)) {
        constexpr bool masked = decltype(MASKED)::value;
        float (&pv)[SX][3] = T.pv;
        const unsigned pok = T.pok;
        const float4 va = T.va, vb = T.vb;
#pragma unroll
        for (int i = 0; i < SX; ++i) {
            float v[3] = {pv[i][0], pv[i][1], pv[i][2]};
            if (has_ab3) {   // padding arrives as 0 and must stay 0
                const unsigned okm = (!masked || ((pok >> i) & 1u)) ? 0xFFFFFFFFu : 0u;
                v[0] = fmaf(v[0], va.x, __uint_as_float(__float_as_uint(vb.x) & okm));
                v[1] = fmaf(v[1], va.y, __uint_as_float(__float_as_uint(vb.y) & okm));
                v[2] = fmaf(v[2], va.z, __uint_as_float(__float_as_uint(vb.z) & okm));
            }
            unsigned h[3], m[3], l[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) s16x_split(v[c], h[c], m[c], l[c]);
            const uint2 H = make_uint2((h[0] >> 16) | h[1], h[2] >> 16), M = make_uint2((m[0] >> 16) | m[1], m[2] >> 16),
                        L = make_uint2((l[0] >> 16) | (l[1] & 0xffff0000u), l[2] >> 16);
            char* d0 = lds + pdst[i];
            char* d1 = lds + C1B + 8 + pdst[i];
            *reinterpret_cast<uint2*>(d0) = H;
            *reinterpret_cast<uint2*>(d0 + PL0) = M;
            *reinterpret_cast<uint2*>(d0 + 2 * PL0) = L;
            *reinterpret_cast<uint2*>(d1) = H;
            *reinterpret_cast<uint2*>(d1 + PL0) = M;
            *reinterpret_cast<uint2*>(d1 + 2 * PL0) = L;
        }
    };
    auto commit = [&](Stage& T) __attribute__((always_inline)) {
        if (T.pok >> 31)
            commit_as(T, std::false_type{});
        else
            commit_as(T, std::true_type{});
    };

    f32x4 acc[4], acs[4];   // the leading product / the five small ones
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = acs[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    // one straight line: per k-step the three filter pieces, per (step, block row) the three operand pieces, each read one unit ahead
    auto sweep = [&]() __attribute__((always_inline)) {
        s16_bf16x8 av[2][3], bv[2][3];
        auto rda = [&](int u, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) av[buf][pc] = __builtin_bit_cast(s16_bf16x8, *reinterpret_cast<const uint4*>(lds + addrA[u >> 2] + (u & 3) * PW * 8 + pc * PL0));
        };
        auto rdb = [&](int j, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bv[buf][pc] = __builtin_bit_cast(s16_bf16x8, *reinterpret_cast<const uint4*>(lds + FB + ((j * 3 + pc) * 64 + lane) * 16));
        };
        rdb(0, 0);
        rda(0, 0);
#pragma unroll
        for (int u = 0; u < KST * 4; ++u) {
            const int j = u >> 2, m = u & 3, b = u & 1, bb = j & 1;
            if (u + 1 < KST * 4) rda(u + 1, b ^ 1);
            if (m == 0 && j + 1 < KST) rdb(j + 1, bb ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            acs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][0], bv[bb][2], acs[m], 0, 0, 0);
            acs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][2], bv[bb][0], acs[m], 0, 0, 0);
            acs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][1], bv[bb][1], acs[m], 0, 0, 0);
            acs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][0], bv[bb][1], acs[m], 0, 0, 0);
            acs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][1], bv[bb][0], acs[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[b][0], bv[bb][0], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] += acs[m];
    };

    // ---- epilogue of one item: accumulator register r of block m, lane (m16, kg) = pixel (row 4 wave + m, column 4 kg + r), channel m16
    const unsigned y_bytes = __builtin_amdgcn_readfirstlane((unsigned)(a.Ho * a.Wo * COUT) * 4u);
    auto epilogue = [&](const Item& I, float* rbuf) __attribute__((always_inline)) {
        const int th_valid = min(kT, a.Ho - I.ty0), tw_valid = min(kT, a.Wo - I.tx0);
        if (a.stats) {
            const float cs = __shfl(acc[0][0], m16);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = 4 * wave + m < th_valid && 4 * kg + r < tw_valid;
                    const float d = ok ? acc[m][r] - cs : 0.f;
                    s1 += d;
                    s2 = fmaf(d, d, s2);
                }
            s1 += __shfl_xor(s1, 16);
            s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lane < 16) {
                rbuf[(wave * 3 + 0) * 16 + lane] = s1;
                rbuf[(wave * 3 + 1) * 16 + lane] = s2;
                rbuf[(wave * 3 + 2) * 16 + lane] = cs;
            }
        }
        float* yn = a.y + (size_t)I.n * a.Ho * a.Wo * COUT;
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(yn)), 0, y_bytes, 0x00020000);
        const int lane_off = ((I.tx0 + 4 * kg) * COUT + m16) * 4;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = 4 * wave + m;
            const int row_off = (I.ty0 + row) * a.Wo * COUT * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = row < th_valid && 4 * kg + r < tw_valid;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[m][r]), yr, ok ? (unsigned)(lane_off + row_off + r * COUT * 4) : kOOB, 0, 0);
            }
        }
        zero_acc();
    };
    auto finalize = [&](const Item& I, const float* rbuf) __attribute__((always_inline)) {
        const int th_valid = min(kT, a.Ho - I.ty0), tw_valid = min(kT, a.Wo - I.tx0);
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int rows = min(4, max(0, th_valid - 4 * w));
            const float cb = (float)(rows * tw_valid);
            if (cb > 0.f) {
                const float S1 = rbuf[(w * 3 + 0) * 16 + tid], S2 = rbuf[(w * 3 + 1) * 16 + tid], sh = rbuf[(w * 3 + 2) * 16 + tid];
                const float mb = sh + S1 / cb, qb2 = fmaxf(S2 - S1 * S1 / cb, 0.f);
                const float nn_ = cnt + cb, d = mb - mean, rr = cb / nn_;
                mean += d * rr;
                m2 += qb2 + d * d * cnt * rr;
                cnt = nn_;
            }
        }
        float* st = a.stats + ((size_t)I.lin * 16 + tid) * 3;
        st[0] = mean;
        st[1] = m2;
        st[2] = cnt;
    };

    if (my_items == 0) return;
    Item cur = decode(0), prev = cur;
    issue(st0, cur, 1);
    commit(st0);
    __syncthreads();   // (the filter and the first patch)
    Item nxt = my_items > 1 ? decode(1) : cur;
    issue(st1, nxt, my_items > 1);
    // one pipeline step: tile `it` is multiplied; the loads of tile it + 1 (stage C) were issued a step ago, those of tile it + 2 go out now (stage I)
    auto step = [&](int it, Stage& C, Stage& I2) __attribute__((always_inline)) {
        if (it > 0 && a.stats && tid < 16) finalize(prev, red + ((it - 1) & 1) * REDF);
        const int live2 = it + 2 < my_items;
        const Item nn = live2 ? decode(it + 2) : cur;
        issue(I2, nn, live2);
        sweep();
        FS_LDS_BARRIER();   // A: every wave is done reading the patch
        if (it + 1 < my_items) commit(C);
        epilogue(cur, red + (it & 1) * REDF);
        FS_LDS_BARRIER();   // B: next patch and this tile's statistics records (LDS) visible; the stores drain during the next sweep
        prev = cur;
        cur = nxt;
        nxt = nn;
    };
    for (int it = 0; it < my_items; it += 2) {
        step(it, st1, st0);
        if (it + 1 < my_items) step(it + 1, st0, st1);
    }
    if (a.stats && tid < 16) finalize(prev, red + ((my_items - 1) & 1) * REDF);
}

// ------------------------------------------------------------------------------------------------------------ host
// FS_S16_SPLIT (default 1): the folded output layer on the bf16 matrix cores as six exact bf16-piece products (conv_s16x_kernel); 0: fp32 matrix instructions
static bool s16_split_on() { return tune_int("FS_S16_SPLIT", 1) != 0; }
static int s16_instance(const ConvArgs& a) {
    if (a.stride != 1) return 0;
    if (a.Cout == 64) {   // VGG16 conv1_1: 3 -> 64, 3x3 (mean on load, bias + ReLU)
        return (a.Cin == 3 && a.KH == 3 && a.KW == 3 && a.dil_x <= 1 && a.src_mode == SRC_PLAIN && !a.in_relu && !a.stats && tune_int("FS_S16_VGG", 1)) ? 3 : 0;
    }
    if (a.Cout != 16 || a.bias || a.out_relu) return 0;
    if (a.Cin == 3 && a.KH == 9 && a.KW == 9 && a.dil_x <= 1 && (a.src_mode == SRC_PLAIN || a.src_mode == SRC_REFLECT) && !a.in_relu) return 1;
    if (a.Cin == 16 && a.KH == 9 && a.KW == 2 && a.dil_x == 5 && a.src_mode == SRC_PLAIN) return 2;
    return 0;
}

bool s16_eligible(const ConvArgs& a) {
    const int inst = s16_instance(a);
    if (!tune_int("FS_S16", 1) || !inst) return false;
    if (a.mask_src || a.route_src || a.pool_out || a.w_nstride || a.w_wino || a.w_wino2 || a.shuffle || a.add_src || a.fin.counter) return false;
    if (a.in_a && !a.in_b) return false;
    if (a.in_relu && !a.in_a) return false;
    if (a.pad_t < 0 || a.pad_l < 0) return false;
    const long tiles = (long)a.N * cdiv(a.Ho, kT) * cdiv(a.Wo, kT);
    return tiles >= tune_int("FS_S16_MIN_TILES", 64);
}

void s16_plan(const ConvArgs& a, ConvPlan* out) {
    ConvPlan p{};
    const int inst = s16_instance(a);
    p.variant = 9;
    p.BN = a.Cout;
    p.CC = a.Cin;
    p.flat = a.Cin == 3;   // (K runs over (kw, ci) contiguously per kernel row)
    p.TH = p.TW = kT;
    p.tiles_y = cdiv(a.Ho, kT);
    p.tiles_x = cdiv(a.Wo, kT);
    p.PH = kT - 1 + a.KH;
    p.PW = inst == 2 ? kT - 1 + 5 + 1 : kT - 1 + a.KW;
    p.S = inst == 2 ? 17 : 3;
    const int patch_f = (p.PH * p.PW * p.S + 8 + 3) & ~3;
    p.lds_bytes = 4 * (patch_f + 2 * 4 * 3 * 16);
    if (inst == 1 && s16_split_on()) {   // conv_s16c3x_kernel: two copies of three piece planes, the filter, the statistics records
        p.S = 8;
        const int pl0 = (((p.PH * p.PW + 2) * 8) + 255) & ~255, c1b = 3 * pl0 + 7 * 16, fb = (c1b + 3 * pl0 + 255) & ~255;
        p.lds_bytes = fb + ((a.KH * ((a.KW + 1) / 2) + 3) / 4) * 3 * 64 * 16 + 4 * (2 * 4 * 3 * 16);
    }
    if (inst == 2 && s16_split_on()) {   // conv_s16x_kernel: two half planes of (pixels + 1) x 48 bytes, each rounded up to 256 bytes
        p.S = 48;
        p.lds_bytes = 2 * (((p.PH * p.PW + 1) * 48 + 255) & ~255) + 4 * (2 * 4 * 3 * 16);
    }
    p.ksplit = 1;
    *out = p;
}

int s16_launch(const ConvArgs& a, hipStream_t s) {
    const ConvPlan& p = a.p;
    const long total = (long)a.N * p.tiles_y * p.tiles_x;
    const int wgs = tune_int("FS_S16_WGS", 512);
    const unsigned grid = (unsigned)(total < wgs ? total : wgs);
    switch (s16_instance(a)) {
        case 1:
            if (p.S == 8) {
                static BigLds lds_attr;
                lds_attr.ensure(reinterpret_cast<const void*>(conv_s16c3x_kernel<9, 9>));
                hipLaunchKernelGGL((conv_s16c3x_kernel<9, 9>), dim3(grid), dim3(256), (size_t)p.lds_bytes, s, a);
            } else {
                hipLaunchKernelGGL((conv_s16_kernel<3, 9, 9, 1>), dim3(grid), dim3(256), (size_t)p.lds_bytes, s, a);
            }
            break;
        case 2:
            if (p.S == 48)
                hipLaunchKernelGGL((conv_s16x_kernel<9, 5>), dim3(grid), dim3(256), (size_t)p.lds_bytes, s, a);
            else
                hipLaunchKernelGGL((conv_s16_kernel<16, 9, 2, 5>), dim3(grid), dim3(256), (size_t)p.lds_bytes, s, a);
            break;
        case 3: hipLaunchKernelGGL((conv_s16_kernel<3, 3, 3, 1, 4>), dim3(grid), dim3(256), (size_t)p.lds_bytes, s, a); break;
        default: return -4;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs
