// Winograd F(4x4, 3x3) with the FILTER OPERAND STREAMED GLOBAL -> REGISTERS (round 4, second half): the transform net's residual
// convolutions (reference im_transf_net.py:250-276: ten 3x3 64 -> 64 VALID convs, each behind an instance norm; in the training
// step also their input gradients, 3x3 'full' convs of dz) and, with 32-tile items, the VGG16 3x3 convs of fs_perceptual_loss
// (reference libs/vgg16.py:45-173) and their input gradients.  Same algorithm as fs_wino4.hip (36 products per 4x4 outputs, fp32
// matrix cores, one wave per SIMD, persistent workgroups over a strided item list); what differs:
//
//   * item = TB x 16 tiles (TB = 1: 4 x 4 tiles = 16 x 16 pixels; TB = 2: 4 x 8 tiles = 16 x 32 pixels) x 64 output channels x 36
//     positions.  Wave w owns the 16 channels 16 w .. of ALL tiles and positions = 144 TB accumulator registers (channels on the
//     matrix instruction's row side, so a lane holds four consecutive channels of ONE tile per tile block at every position: output
//     transform in registers, 16-byte stores).  A 720p frame is 180 x 320 pixels at the residual depth = 143 items of 16 x 32 pixels
//     for 256 CUs, a batch-4 training step 72: those launches take TB = 1 (240 and 100-144 items);
//   * input channels in steps of 8 = two matrix instructions per position and tile block: 72 TB per step and wave;
//   * every wave needs a DIFFERENT 16-channel block of the filter, so nothing of it is shared between waves: the A operand goes
//     global -> registers directly, pre-arranged by wt_wino4t in the matrix instruction's own lane layout (lane = k * 16 + m holds 4
//     consecutive slots per 16-byte load: 18 loads per step and lane, no LDS write, no LDS read -- fs_wino4.hip moves the same
//     bytes per step through 9 loads + 9 LDS writes + 36 LDS reads for HALF the matrix instructions).  A filter quad is
//     reloaded for the next step right behind the last matrix instruction that reads it (one register set);
//   * LDS holds only the transformed input V (72 blocks of [k][tile], skewed: conflict-free for the transform's writes and the
//     operand reads) and the raw 18 x (16 TB + 2) x 8 patch (channel-planar) -- 33 KB (TB = 1) / 67 KB (TB = 2) per stage;
//   * input transform of the tiles x 8 channels of a step on lane pairs with v_permlane32_swap, as in fs_wino4.hip (TB = 2: two passes);
//   * on load: the producer's instance norm + ReLU (AFF; padding 0 only); epilogue forms: raw (also split-K partials), raw +
//     per-item instance-norm partials {mean, M2, count}, + the residual gradient added in the interior, bias + ReLU (+ the fused
//     2x2 max-pool), the consumer's ReLU mask.
#include "fs_wino4t_kernel.h"

namespace fs {

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

// The 128-channel item form (M = 3 of the kernel): U4u[ci/8][co/32][q (36)][lane = (ci%4) * 16 + co%16][e], quad q = slot / 2, e = (slot % 2) * 2 + (co/16) % 2
// -- the wave that owns channels co/32 * 32 .. +31 reads, per slot, the A operands of its two 16-channel blocks from one quad.
__global__ __launch_bounds__(256) void wt_wino4u_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t cc = (size_t)Cin * Cout;
    if (i >= cc) return;
    const int ci = (int)(i / Cout), co = (int)(i - (size_t)ci * Cout);
    double o[36];
    wino4_filter_transform(w, cc, i, o);
    float* dst = U + ((size_t)(ci >> 3) * (Cout >> 5) + (co >> 5)) * (36 * 256) + ((ci & 3) * 16 + (co & 15)) * 4 + ((co >> 4) & 1);
    const int sub = (ci >> 2) & 1;
#pragma unroll
    for (int pos = 0; pos < 36; ++pos) {
        const int slot = sub * 36 + pos;
        dst[(slot >> 1) * 256 + (slot & 1) * 2] = (float)o[pos];
    }
}

int wt_wino4u(const float* w, float* U, int Cin, int Cout, hipStream_t s) {
    if (Cin % kCC || Cout % (2 * kBN)) return -1;
    const size_t cc = (size_t)Cin * Cout;
    hipLaunchKernelGGL(wt_wino4u_kernel, dim3((unsigned)((cc + 255) / 256)), dim3(256), 0, s, w, U, Cin, Cout);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int wt_wino4t_batch(const WinoBatch& b, int Cin, int Cout, hipStream_t s) {
    if (Cin % kCC || Cout % kBN || b.n < 0 || b.n > 24) return -1;
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
extern "C" int fs_debug_wino4t_trace_1a(long long* out, int n_wg) { return wino4t_trace_read(out, n_wg); }
#endif

// which epilogue form a launch takes (-1: none fits)
static int wino4t_epi(const ConvArgs& a, int ksplit) {
    const bool act = a.bias || a.out_relu || a.pool_out;
    if (ksplit > 1) return a.inb_rec ? -1 : 0;
    if (a.inb_rec) return (act || a.mask_src || a.stats || a.in_a || !a.inb_z || !a.inb_mean || !a.inb_rstd || (a.inb_relu && (!a.inb_a || !a.inb_b))) ? -1 : (a.add_src ? 6 : 5);
    if (a.stats) return (act || a.mask_src || a.add_src) ? -1 : 1;
    if (a.add_src) return (act || a.mask_src) ? -1 : 2;
    if (a.mask_src) return act ? -1 : 4;
    return act ? 3 : 0;
}

bool wino4t_eligible(const ConvArgs& a) {
    // 3x3 stride 1 with padding 0 (the residual convs), 2 (their input gradients) or 1 (the VGG16 convs); plain source
    const bool pad_ok = a.pad_t == a.pad_l && a.pad_t >= 0 && a.pad_t <= 2 && a.Ho == a.H + 2 * a.pad_t - 2 && a.Wo == a.W + 2 * a.pad_l - 2;
    // byte offsets inside one sample are 32-bit with the top bit reserved for "out of range"; the filter likewise
    const bool fits = (double)a.H * a.W * a.Cin * 4.0 < 2147483648.0 && (double)a.Ho * a.Wo * a.Cout * 4.0 < 2147483648.0 && 36.0 * a.Cin * a.Cout * 4.0 < 2147483648.0;
    const bool aff_ok = !a.in_a || (a.pad_t == 0 && a.in_b && a.in_relu && (a.in_nstride == 0 || a.in_nstride == a.Cin) && !a.add_src && !a.mask_src && !a.bias &&
                                    !a.out_relu && !a.pool_out);
    // item / step indices are decoded with float reciprocals in the kernel ((int)((x + 0.5f) * inv_d)): exact below 2^22 -- larger launches take the other kernels
    const bool count_ok = (double)a.N * cdiv(a.Ho, kBH) * cdiv(a.Wo, 16) * (a.Cout / kBN) * 4.0 /* split-K */ < 4194304.0;
    return fits && count_ok && a.w_wino4t && a.KH == 3 && a.KW == 3 && a.stride == 1 && pad_ok && a.src_mode == SRC_PLAIN && a.Cin % kCC == 0 && a.Cout % kBN == 0 &&
           !a.shuffle && aff_ok && wino4t_epi(a, 1) >= 0 && !a.route_src && a.w_nstride == 0 && a.dil_x <= 1 && !a.fin.counter && a.Ho > 0 && a.Wo > 0 &&
           (!a.pool_out || (!(a.Ho & 1) && !(a.Wo & 1)));
}

static long wino4t_items_tb(const ConvArgs& a, int tb) { return (long)a.N * cdiv(a.Ho, kBH) * cdiv(a.Wo, 16 * tb) * (a.Cout / kBN); }

// tile blocks per item: 32-tile items amortise the filter loads over twice the products (a step costs ~1.8x that of a 16-tile item), 16-tile
// items fill small grids.  FS_WINO4T_TB = 1 | 2 pins it.
static int wino4t_pick_tb(const ConvArgs& a) {
    const int forced = tune_int("FS_WINO4T_TB", 0);
    if (forced == 1 || forced == 2) return forced;
    if (a.tnet_plan == 1) return 1;   // the transform net's 8-step items: measured level at batch 32 (800 16-tile / 480 32-tile items), 16-tile items ahead everywhere else
    const int wgs = tune_int("FS_WINO4T_WGS", 256);
    const double c1 = (double)cdiv((int)wino4t_items_tb(a, 1), wgs), c2 = 1.8 * (double)cdiv((int)wino4t_items_tb(a, 2), wgs);
    return c2 <= c1 ? 2 : 1;
}

long wino4t_items(const ConvArgs& a) { return wino4t_items_tb(a, wino4t_pick_tb(a)); }

// The 128-channel item form (16 tiles x 128 channels; the filter in the layout of wt_wino4u): where 32-tile items would be taken and the
// layer has the channels for it -- same item count, but ONE input-transform pass per 144 matrix instructions instead of two.
// FS_WINO4T_CB = 1 disables it.
static bool wino4t_use_cb2(const ConvArgs& a) {
    const int e = wino4t_epi(a, 1);
    return a.w_wino4u && a.Cout % (2 * kBN) == 0 && !a.in_a && (e == 0 || e == 3 || e == 4) && tune_int("FS_WINO4T_CB", 2) >= 2 && wino4t_pick_tb(a) == 2;
}

// The flattened form (M = 4 of the kernel, round 5): 16 consecutive tiles of the sample's row-major tile list per item -- ceil(tiles / 16) items per
// sample instead of ceil(Ho / 16) ceil(Wo / 16), at the price of a 6 x 6 patch per tile (1.8x the patch loads).  Taken where it saves a whole ROUND
// of the persistent grid (batch 32 at 256 x 256: 736 against 800 items = three rounds instead of four); the transform net's launches only (its
// epilogue forms).  FS_WINO4T_FLAT = 0 never, 2 always (tests).
static bool wino4t_use_flat(const ConvArgs& a, int tb, bool cb2) {
    const int mode = tune_int("FS_WINO4T_FLAT", 1);
    const int e = wino4t_epi(a, 1);
    if (!mode || tb != 1 || cb2 || !(e == 0 || e == 1 || e == 2 || e == 5 || e == 6) || a.split_ws) return false;
    if (mode == 2) return true;
    if (!a.tnet_plan) return false;
    const int wgs = tune_int("FS_WINO4T_WGS", 256);
    const long ncob = a.Cout / kBN;
    const long rect = (long)a.N * cdiv(a.Ho, kBH) * cdiv(a.Wo, 16) * ncob, flat = (long)a.N * cdiv(cdiv(a.Ho, 4) * cdiv(a.Wo, 4), 16) * ncob;
    return cdiv((int)flat, wgs) < cdiv((int)rect, wgs);
}

void wino4t_plan(const ConvArgs& a, ConvPlan* out) {
    ConvPlan p{};
    const bool cb2 = wino4t_use_cb2(a);
    const int tb = cb2 ? 1 : wino4t_pick_tb(a);
    p.variant = 11;
    p.BN = cb2 ? 2 * kBN : kBN;
    p.CC = kCC;
    p.TH = kBH;
    p.TW = 16 * tb;
    p.tiles_y = cdiv(a.Ho, kBH);
    p.tiles_x = cdiv(a.Wo, p.TW);
    p.lds_bytes = 4 * 2 * (tb == 1 ? Geo<1>::kStageF : Geo<2>::kStageF);
    if (wino4t_use_flat(a, tb, cb2)) {
        p.flat_tiles = 1;
        p.tiles_y = 1;
        p.tiles_x = cdiv(cdiv(a.Ho, 4) * cdiv(a.Wo, 4), 16);   // items (= statistics / partial-sum records) per sample
        p.lds_bytes = 4 * 2 * Geo<1, true>::kStageF;
    }
    p.ksplit = 1;
    const long items = (long)a.N * p.tiles_y * p.tiles_x * (a.Cout / p.BN);
    const int nchunks = a.Cin / kCC;
    const int max_ks = tune_int("FS_WINO_KSPLIT", 4);
    if (a.split_ws && !a.pool_out && !a.stats && !a.in_a && !a.inb_rec && !p.flat_tiles) {   // split-K where the launch cannot fill the chip (the rule of fs_wino4.hip; a step here is 8 channels)
        int ks = 1;
        const int min_steps = tune_int("FS_WINO4_KSPLIT_MINSTEPS", 16) / 2;
        while (ks < max_ks && items * ks < 256 && nchunks / (ks * 2) >= min_steps && (size_t)(ks * 2) * a.N * a.Ho * a.Wo * a.Cout <= a.split_ws_floats) ks *= 2;
        p.ksplit = ks;
    }
    *out = p;
}

// (the other instantiations: fs_wino4t1b.hip, fs_wino4t2.hip, fs_wino4t2b.hip)
int wino4t_launch_1b(const ConvArgs& a, int epi, long grid, hipStream_t s);
int wino4t_launch_1c(const ConvArgs& a, int epi, long grid, hipStream_t s);
int wino4t_launch_1d(const ConvArgs& a, int epi, long grid, hipStream_t s);
int wino4t_launch_4a(const ConvArgs& a, int epi, long grid, hipStream_t s);
int wino4t_launch_4b(const ConvArgs& a, int epi, long grid, hipStream_t s);
int wino4t_launch_2a(const ConvArgs& a, int epi, long grid, hipStream_t s);
int wino4t_launch_2b(const ConvArgs& a, int epi, long grid, hipStream_t s);

int wino4t_launch(const ConvArgs& a, hipStream_t s) {
    const ConvPlan& p = a.p;
    const int tb = p.TW / 16;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int epi = wino4t_epi(a, ks);
    if (epi < 0 || (tb != 1 && tb != 2) || (a.in_a && epi > 1)) return -7;
    const long items = (long)a.N * p.tiles_y * p.tiles_x * (a.Cout / p.BN) * ks;
    const int wgs = tune_int("FS_WINO4T_WGS", 256);
    const long grid = items < wgs ? items : wgs;
    if (p.flat_tiles) {
        if (tb != 1 || p.BN != kBN || ks != 1) return -7;
        return (a.in_a || epi <= 1) ? wino4t_launch_4a(a, epi, grid, s) : wino4t_launch_4b(a, epi, grid, s);
    }
    if (p.BN == 2 * kBN) return (tb == 1 && a.w_wino4u) ? wino4t_launch_1c(a, epi, grid, s) : -7;
    const bool part_a = a.in_a || epi <= 1;
    if (epi >= 5) return tb == 1 ? wino4t_launch_1d(a, epi, grid, s) : -7;   // (instance-norm-backward partial sums: 16-tile items only)
    if (tb == 1) return part_a ? wino4t_launch_part_a<1>(a, epi, grid, s) : wino4t_launch_1b(a, epi, grid, s);
    return part_a ? wino4t_launch_2a(a, epi, grid, s) : wino4t_launch_2b(a, epi, grid, s);
}

}  // namespace fs
