// Transform-net host orchestration (layout + launch sequences); see fs_tnet.hip.
#pragma once
#include "fs_kernels.h"

#include <cstdio>

namespace fs {

struct ParamInfo {
    char name[48];
    int offset, count, ndim, dims[4];
};
const ParamInfo* param_table();  // 48 entries, checkpoint key order

// One conv + instance-norm unit of the net.
struct Unit {
    int kind;  // 0: conv, 1: phase-collapsed resize-conv (2x2 taps, pixel-shuffle store),
               // 2: kw-folded 9x9 -> 3-channel output layer (fs_fold.hip)
               // 3: conv2d_transpose (--upsample_method deconv, im_transf_net.py:158-190): the forward is the
               //    input-gradient kernel call of a stride-`dstride` conv with filter [K,K,Cout,Cin]
    int K, stride, Cin, Cout;
    int KWx, dil_x;    // horizontal taps / tap spacing the conv kernel sees (K / 1 except kind 2: 2 / 5)
    int dstride, dpad_t, dpad_l;  // kind 3: stride and SAME padding of the conv whose transpose this is
    int Cc;            // channels the conv kernel produces (4*Cout for kind 1)
    int Hsrc, Wsrc;    // tensor the conv kernel reads (the unpadded image for initconv_0)
    int Hin, Win;      // conv input extent (after reflect padding)
    int Hc, Wc;        // conv kernel output extent
    int Hout, Wout;    // unit output extent (2x Hc for kind 1)
    int pad_t, pad_l, src_mode, refl;
    int w_off, g_off, b_off;  // offsets into the flat parameter buffer
    size_t z, stats, mean, rstd, a, b;  // workspace offsets (floats)
    int tiles;
    int wino;        // forward through a Winograd kernel: 1 wino(2)_conv_kernel (3x3 VALID residual convs on grids that fill the
                     // chip with 64-tile items), 2 wino2h_conv_kernel (half items: smaller grids, batch 4 per GPU), 3 wino4t_conv_kernel (F(4x4,3x3))
    int x6;          // forward through the split-bf16 direct kernel (conv_r64x_kernel, round 6): the residual convs wherever the launch has enough 8 x 16-pixel tiles
    size_t wino_u;   // its transformed filter ([16][Cin][Cout]; 36 * Cin * Cout floats for 3) in the workspace
    ConvPlan plan;
    WgradPlan wplan;
};

struct TnetLayout {
    int N, H, W, Hy, Wy;
    int deconv;       // upsample_method == 'deconv'
    int wino_mode;    // FS_TNET_WINO at layout time (0 off, 1 auto, 2 forced): part of the layout's identity
    Unit u[16];
    size_t h[5];      // residual block outputs
    size_t weff[2];   // collapsed resize-conv filters
    size_t zfold, wfold, dwfold;  // kw-folded output layer: Z / unfolded dY [N,Ho,Wo+4,16], filters, filter grads
    size_t rem_ws;      // scratch of the remainder split of fs_wino2 (256 units x 16x16 pixels x 64 channels)
    size_t fin_counter; // 16 unsigned: the "last workgroup" counters of the fused instance-norm finalize (fs_kernels.h FinArgs)
    size_t fwd_floats;
    size_t wTu[16];   // per-unit input-gradient filters (flip+transpose / collapsed), all built by one wt_batch launch
    size_t dweff2;    // the collapsed filter gradient of the SECOND resize-conv unit (both units' reductions are pending at once since round 5)
    size_t g[3], dz[2], wT, dweff, inbwd, slabs;  // backward scratch (dz double-buffered: filter gradients run on a side stream)
    size_t wino_d[10]; // Winograd-transformed input-gradient filters of the residual convs (0: direct kernel)
    int wino_dh[10];   // ... 1: through the half-item kernel, 2: through the 16-tile F(4x4) kernel (fs_wino4t.hip)
    size_t inb_rec;   // instance-norm-backward partial-sum records [N][items][64][2] written by the epilogue of a residual input-gradient launch
                      // (fs_wino4t_kernel.h EPI 5 / 6) for the unit below it; one buffer, consumed by that unit's in_bwd_rec right after
    size_t inb_S[16]; // per unit: the per-sample sums [N][Cout][2] in_bwd_rec leaves for in_bwd_params (dgamma / dbeta of all units in one launch)
    size_t slab_u[16]; // per unit: slabs of its own filter-gradient launch (the six non-residual units' reductions run as ONE launch at the end)
    size_t dzres[10]; // dz of the ten residual convs, kept until their filter gradients run as ONE launch (fs_wgrad2.hip)
    int res_batch;    // 1: that batched launch is planned (shapes eligible)
    size_t total_floats;
};

void tnet_layout(int N, int H, int W, int deconv, TnetLayout* L);
int tnet_wino_mode();  // current FS_TNET_WINO
WgradArgs unit_wgrad_args(const Unit& u, int N);
// reuse_filters: the re-laid-out filters in ws are those of the previous call (FS_FLAG_PARAMS_FROZEN): skip the kernels that build them
// with_bwd_filters: also build the input-gradient filters tnet_backward needs (same launches: the parameters cannot change between the forward
// and the backward of a step) -- tnet_backward(..., filters_ready = true) then skips its own two re-layout launches
int tnet_forward(const TnetLayout& L, const float* params, const float* x, float* y, float* ws, hipStream_t s, bool reuse_filters = false,
                 bool with_bwd_filters = false);
// Optional second stream + events: the filter gradients of unit i only need dz_i, so they run concurrently
// with the input-gradient chain of the units below (both are small launches at batch 4).
struct StreamAux {
    hipStream_t side;
    hipEvent_t* ev;
    int nev;
};
int tnet_backward(const TnetLayout& L, const float* params, const float* x, const float* dy, float* grads, float* ws,
                  hipStream_t s, const StreamAux* aux, bool filters_ready = false);

}  // namespace fs
