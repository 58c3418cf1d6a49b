// Host orchestration of the image-transform net (reference im_transf_net.py:14-75) on top of
// the kernels in fs_conv.hip / fs_wgrad.hip / fs_elem.hip: workspace layout, the forward launch
// sequence (16 convs, each followed by the statistics finalize; normalisation+ReLU folded into
// the consumer's load) and the hand-derived backward sequence.  No allocation, no sync.
#include "fs_tnet.h"

#include <cstdlib>

#include <cstring>

namespace fs {

// ---------------------------------------------------------------- parameter table (sorted keys)
static ParamInfo g_params[48];
static bool g_params_ready = false;

static void add_param(int& idx, int& off, const char* layer, const char* leaf, int nd, int d0, int d1, int d2, int d3) {
    ParamInfo& p = g_params[idx++];
    snprintf(p.name, sizeof(p.name), "%s/%s", layer, leaf);
    p.offset = off;
    p.ndim = nd;
    p.dims[0] = d0;
    p.dims[1] = d1;
    p.dims[2] = d2;
    p.dims[3] = d3;
    int n = 1;
    for (int i = 0; i < nd; ++i) n *= p.dims[i];
    p.count = n;
    off += n;
}

const ParamInfo* param_table() {
    if (g_params_ready) return g_params;
    int idx = 0, off = 0;
    const int ic[3][3] = {{9, 3, 16}, {3, 16, 32}, {3, 32, 64}};
    for (int i = 0; i < 3; ++i) {
        char l[32];
        snprintf(l, sizeof(l), "initconv_%d", i);
        add_param(idx, off, l, "INscale", 1, ic[i][2], 1, 1, 1);
        add_param(idx, off, l, "INshift", 1, ic[i][2], 1, 1, 1);
        add_param(idx, off, l, "W", 4, ic[i][0], ic[i][0], ic[i][1], ic[i][2]);
    }
    for (int i = 0; i < 5; ++i) {
        char l[32];
        snprintf(l, sizeof(l), "resblock_%d", i);
        add_param(idx, off, l, "INscale1", 1, 64, 1, 1, 1);
        add_param(idx, off, l, "INscale2", 1, 64, 1, 1, 1);
        add_param(idx, off, l, "INshift1", 1, 64, 1, 1, 1);
        add_param(idx, off, l, "INshift2", 1, 64, 1, 1, 1);
        add_param(idx, off, l, "W1", 4, 3, 3, 64, 64);
        add_param(idx, off, l, "W2", 4, 3, 3, 64, 64);
    }
    const int uc[3][3] = {{3, 64, 32}, {3, 32, 16}, {9, 16, 3}};
    for (int i = 0; i < 3; ++i) {
        char l[32];
        snprintf(l, sizeof(l), "upsample_%d", i);
        add_param(idx, off, l, "INscale", 1, uc[i][2], 1, 1, 1);
        add_param(idx, off, l, "INshift", 1, uc[i][2], 1, 1, 1);
        add_param(idx, off, l, "W", 4, uc[i][0], uc[i][0], uc[i][1], uc[i][2]);
    }
    g_params_ready = true;
    return g_params;
}

static int find_param(const char* name) {
    const ParamInfo* t = param_table();
    for (int i = 0; i < 48; ++i)
        if (!strcmp(t[i].name, name)) return t[i].offset;
    return -1;
}

static void same_pads(int size, int k, int s, int* out, int* before) {
    *out = cdiv(size, s);
    int tot = (*out - 1) * s + k - size;
    if (tot < 0) tot = 0;
    *before = tot / 2;
}

// ---------------------------------------------------------------- layout
struct Bump {
    size_t off = 0;
    size_t take(size_t floats) {
        const size_t o = off;
        off += (floats + 63) & ~(size_t)63;  // 256-byte granules
        return o;
    }
};

static ConvArgs unit_args(const Unit& u, int N) {
    ConvArgs a{};
    a.N = N;
    a.H = u.Hsrc;
    a.W = u.Wsrc;
    a.Cin = u.Cin;
    a.Ho = u.Hc;
    a.Wo = u.Wc;
    a.Cout = u.Cc;
    a.KH = u.K;
    a.KW = u.KWx;
    a.dil_x = u.dil_x;
    a.stride = u.stride;
    a.pad_t = u.pad_t;
    a.pad_l = u.pad_l;
    a.src_mode = u.src_mode;
    a.refl = u.refl;
    a.shuffle = u.kind == 1;
    a.prof_tag = 1;
    a.tnet_plan = 1;
    a.res_x6 = u.x6;
    return a;
}

constexpr int kRemUnits = 256;   // at most one persistent grid of split units

int tnet_wino_mode() {
    return tune_int("FS_TNET_WINO", 1);
}

void tnet_layout(int N, int H, int W, int deconv, TnetLayout* L) {
    memset(L, 0, sizeof(*L));
    L->deconv = deconv;
    L->wino_mode = tnet_wino_mode();
    L->N = N;
    L->H = H;
    L->W = W;
    Bump b;
    int h = H + 80, w = W + 80;  // after reflect_pad(40)   im_transf_net.py:34
    int ui = 0;
    auto conv_unit = [&](const char* layer, const char* wleaf, const char* gleaf, const char* bleaf, int K, int stride,
                         int Cin, int Cout, int valid, int hin, int win) -> Unit& {
        Unit& u = L->u[ui++];
        char nm[64];
        snprintf(nm, sizeof(nm), "%s/%s", layer, wleaf);
        u.w_off = find_param(nm);
        snprintf(nm, sizeof(nm), "%s/%s", layer, gleaf);
        u.g_off = find_param(nm);
        snprintf(nm, sizeof(nm), "%s/%s", layer, bleaf);
        u.b_off = find_param(nm);
        u.kind = 0;
        u.K = K;
        u.KWx = K;
        u.dil_x = 1;
        u.stride = stride;
        u.Cin = Cin;
        u.Cout = u.Cc = Cout;
        u.Hin = u.Hsrc = hin;
        u.Win = u.Wsrc = win;
        if (valid) {
            u.Hc = hin - K + 1;
            u.Wc = win - K + 1;
            u.pad_t = u.pad_l = 0;
        } else {
            same_pads(hin, K, stride, &u.Hc, &u.pad_t);
            same_pads(win, K, stride, &u.Wc, &u.pad_l);
        }
        u.Hout = u.Hc;
        u.Wout = u.Wc;
        return u;
    };
    {  // initconv_0: reflect pad fused into the load
        Unit& u = conv_unit("initconv_0", "W", "INscale", "INshift", 9, 1, 3, 16, 0, h, w);
        u.src_mode = SRC_REFLECT;
        u.refl = 40;
        u.Hsrc = H;
        u.Wsrc = W;
        same_pads(h, 9, 1, &u.Hc, &u.pad_t);
        same_pads(w, 9, 1, &u.Wc, &u.pad_l);
        u.Hout = u.Hc;
        u.Wout = u.Wc;
    }
    {
        Unit& u = conv_unit("initconv_1", "W", "INscale", "INshift", 3, 2, 16, 32, 0, h, w);
        h = u.Hout;
        w = u.Wout;
    }
    {
        Unit& u = conv_unit("initconv_2", "W", "INscale", "INshift", 3, 2, 32, 64, 0, h, w);
        h = u.Hout;
        w = u.Wout;
    }
    for (int k = 0; k < 5; ++k) {
        char l[32];
        snprintf(l, sizeof(l), "resblock_%d", k);
        conv_unit(l, "W1", "INscale1", "INshift1", 3, 1, 64, 64, 1, h, w);
        conv_unit(l, "W2", "INscale2", "INshift2", 3, 1, 64, 64, 1, h - 2, w - 2);
        h -= 4;
        w -= 4;
    }
    for (int k = 0; k < 2; ++k) {  // phase-collapsed resize-conv: 2x2 taps, 4*Cout virtual channels
        char l[32];
        snprintf(l, sizeof(l), "upsample_%d", k);
        const int Cin = k == 0 ? 64 : 32, Cout = k == 0 ? 32 : 16;
        Unit& u = conv_unit(l, "W", "INscale", "INshift", 3, 1, Cin, Cout, 0, h, w);
        if (deconv) {  // conv2d_transpose 3x3 stride 2: zero-dilated input, flipped filter, pad K-1-pad_f
            u.kind = 3;
            u.dstride = 2;
            int dummy;
            same_pads(2 * h, 3, 2, &dummy, &u.dpad_t);
            same_pads(2 * w, 3, 2, &dummy, &u.dpad_l);
            u.src_mode = SRC_DILATE2;
            u.stride = 1;
            u.pad_t = 2 - u.dpad_t;
            u.pad_l = 2 - u.dpad_l;
            u.Hc = u.Hout = 2 * h;
            u.Wc = u.Wout = 2 * w;
            h = u.Hout;
            w = u.Wout;
            continue;
        }
        u.kind = 1;
        u.K = 2;
        u.KWx = 2;
        u.Cc = 4 * Cout;
        u.Hc = h;
        u.Wc = w;
        u.pad_t = u.pad_l = 0;
        u.Hout = 2 * h;
        u.Wout = 2 * w;
        h = u.Hout;
        w = u.Wout;
    }
    {  // output layer: 9x9, 16 -> 3, computed kw-folded (fs_fold.hip): 9x2 taps, spacing 5, 16 virtual channels
        Unit& u = conv_unit("upsample_2", "W", "INscale", "INshift", 9, 1, 16, 3, 0, h, w);
        if (deconv) {  // conv2d_transpose 9x9 stride 1 == conv with the flipped/transposed filter
            u.kind = 3;
            u.dstride = 1;
            u.dpad_t = u.pad_t;
            u.dpad_l = u.pad_l;
            u.pad_t = 8 - u.dpad_t;
            u.pad_l = 8 - u.dpad_l;
        } else {
        u.kind = 2;
        u.KWx = 2;
        u.dil_x = 5;
        u.Cc = 16;
        u.Wc = u.Wout + 4;
        }
    }
    L->Hy = h;
    L->Wy = w;

    size_t max_act = 0, max_slab = 0, max_inbwd = 0;
    for (int i = 0; i < 16; ++i) {
        Unit& u = L->u[i];
        ConvArgs a = unit_args(u, N);
        // Residual convs (3x3 VALID, 64 -> 64) go through the Winograd kernel when its 16x16-pixel blocks fill the
        // chip (720p / 1080p frames: 240-290 blocks per image); at 256x256 batch 4 they are 100 blocks for 256 CUs and
        // the direct kernel stays.  FS_TNET_WINO=0 disables, =2 forces (tests).
        u.wino = 0;
        u.x6 = 0;
        // (round 6) the residual convs as a direct convolution in six exact bf16-piece products (fs_cstream.hip: conv_r64x_kernel) where the fp32 Winograd
        // kernel's 16 x 16-pixel items leave most of the chip idle -- batch 4 per GPU: 100 items on 256 CUs against 200 tiles of 8 x 16 pixels; measured
        // (same lease, profiles/r06_ab_r64x_residual_forward.txt): batch 4 3.22 -> 3.20 ms per step.  Where the items fill the chip the Winograd kernel is
        // ahead (batch 32: ten forward launches 0.75 against 0.88 ms; 720p 1215 against 1112 fps): per 8 x 16 tile the direct kernel's sweep is 15.7 k
        // cycles at the full bf16 matrix rate, and issue + commit + epilogue add 10 k that one wave per SIMD cannot overlap (tools/r64x_trace.py).
        // FS_TNET_RES_X6: 0 never, 1 (default) below FS_TNET_RES_X6_MAX_ITEMS Winograd items (129: less than half of the chip; a 720p frame has 240), 2 wherever the kernel takes the launch.
        if (u.kind == 0 && i >= 3 && i <= 12 && tune_int("FS_TNET_RES_X6", 1)) {
            const long w_items = (long)N * cdiv(u.Hc, 16) * cdiv(u.Wc, 16);
            if (tune_int("FS_TNET_RES_X6", 1) == 2 || w_items < (long)tune_int("FS_TNET_RES_X6_MAX_ITEMS", 129)) {
                a.res_x6 = 1;
                a.stats = reinterpret_cast<float*>(16);
                u.x6 = cstream_eligible(a) ? 1 : 0;
                a.stats = nullptr;
                a.res_x6 = u.x6;
            }
        }
        if (!u.x6) {
            const int mode = L->wino_mode;
            a.w_wino = reinterpret_cast<const float*>(16);   // eligibility looks at the shapes only
            a.w_wino2 = a.w_wino;
            a.stats = reinterpret_cast<float*>(16);
            const long blocks = (long)N * cdiv(u.Hc, 16) * cdiv(u.Wc, 16) * (u.Cc / 64 > 0 ? u.Cc / 64 : 1);
            if (mode && u.kind == 0 && i >= 3 && i <= 12 && (wino_eligible(a) || wino2_eligible(a)) && (mode == 2 || blocks >= 200)) u.wino = 1;
            // ... and through the half-item kernel (<= 32 tiles per item: twice the items) when 64-tile items leave most of
            // the chip idle -- batch 4 per GPU: 100..144 items of 16x16 pixels, 180..252 half items
            if (!u.wino && mode && mode != 2 && u.kind == 0 && i >= 3 && i <= 12 && wino_gen().f2_second() && tune_int("FS_TNET_WINO_HALF", 1)) {
                ConvArgs h = a;
                h.half_items = 1;
                if (wino2h_eligible(h) && wino2h_items(h) >= tune_int("FS_WINO2H_MIN_ITEMS", 96)) u.wino = 2;
            }
            // ... and through the 16-tile Winograd F(4x4,3x3) kernel (fs_wino4t.hip: 36 products per 4x4 outputs instead of 64, filter
            // operand global -> registers) wherever its items occupy a fair part of the chip.  FS_TNET_WINO4=0 disables, =2 forces.
            const int w4 = tune_int("FS_TNET_WINO4", 1);
            if (mode && w4 && u.kind == 0 && i >= 3 && i <= 12) {
                ConvArgs t = a;
                t.w_wino = t.w_wino2 = nullptr;
                t.w_wino4t = reinterpret_cast<const float*>(16);
                if (wino4t_eligible(t) && (w4 == 2 || wino4t_items(t) >= tune_int("FS_WINO4T_MIN_ITEMS", 64))) u.wino = 3;
            }
            if (!u.wino || u.wino == 3) a.w_wino = a.w_wino2 = nullptr;
            if (u.wino == 3) a.w_wino4t = reinterpret_cast<const float*>(16);
            a.stats = nullptr;
        }
        a.half_items = u.wino == 2;
        a.rem_ws = reinterpret_cast<float*>(16);   // (planning looks at the capacity only)
        a.rem_ws_floats = (size_t)kRemUnits * 16384;
        u.plan = conv_plan(a);
        u.wino_u = u.wino ? b.take((size_t)(u.wino == 3 ? 36 : 16) * u.Cin * u.Cc) : 0;
        u.tiles = u.kind == 2 ? cdiv(u.Hout * u.Wout, 256) : u.plan.tiles_y * u.plan.tiles_x;
        const size_t act = (size_t)N * u.Hout * u.Wout * u.Cout;
        u.z = b.take(act);
        // per-tile records + room for the pre-reduced records of in_finalize
        u.stats = b.take((size_t)N * (u.tiles + kFinalizeSplit) * (u.kind == 2 ? u.Cout : u.Cc) * 3);
        u.mean = b.take((size_t)N * u.Cout);
        u.rstd = b.take((size_t)N * u.Cout);
        u.a = b.take((size_t)N * u.Cout);
        u.b = b.take((size_t)N * u.Cout);
        if (act > max_act) max_act = act;
        const size_t ib = in_bwd_scratch_floats(N, u.Hout * u.Wout, u.Cout);
        if (ib > max_inbwd) max_inbwd = ib;
    }
    for (int k = 0; k < 5; ++k) {
        const Unit& u2 = L->u[3 + 2 * k + 1];
        L->h[k] = b.take((size_t)N * u2.Hout * u2.Wout * 64);
    }
    L->weff[0] = b.take(4 * 64 * 128);
    L->weff[1] = b.take(4 * 32 * 64);
    L->zfold = b.take((size_t)N * L->u[15].Hc * L->u[15].Wc * 16);
    L->wfold = b.take(18 * 16 * 16);
    L->dwfold = b.take(18 * 16 * 16);
    L->fin_counter = b.take(64);
    L->rem_ws = b.take((size_t)kRemUnits * 16384);
    L->fwd_floats = b.off;
    // ---- backward scratch ----
    for (int k = 0; k < 3; ++k) L->g[k] = b.take(max_act);
    L->dz[0] = b.take(max_act);
    L->dz[1] = b.take(max_act);
    L->wT = b.take(4);
    for (int i = 1; i < 16; ++i)
        L->wTu[i] = L->u[i].kind == 3 ? 0
                                      : b.take((size_t)(L->u[i].kind == 1 ? 9 : (L->u[i].stride == 2 ? 16 : L->u[i].K * L->u[i].K)) *
                                               L->u[i].Cin * L->u[i].Cout);
    L->dweff = b.take(4 * 64 * 128);
    L->dweff2 = b.take(4 * 64 * 128);
    for (int i = 3; i <= 12; ++i) {   // residual input gradients (3x3 'full' convs of dz) through the Winograd kernel when its blocks fill the chip
        const Unit& u = L->u[i];
        const long blocks = (long)N * cdiv(u.Hin, 16) * cdiv(u.Win, 16);
        bool on = L->wino_mode && wino_gen().f2_second() && (L->wino_mode == 2 || blocks >= 200);
        L->wino_dh[i - 3] = 0;
        if (!on && L->wino_mode && L->wino_mode != 2 && wino_gen().f2_second() && tune_int("FS_TNET_WINO_HALF", 1)) {
            ConvArgs h{};   // the 3x3 'full' conv of dz that unit_dgrad launches
            h.N = N;
            h.H = u.Hout;
            h.W = u.Wout;
            h.Cin = h.Cout = 64;
            h.Ho = u.Hin;
            h.Wo = u.Win;
            h.KH = h.KW = 3;
            h.stride = 1;
            h.pad_t = h.pad_l = 2;
            h.w_wino2 = reinterpret_cast<const float*>(16);
            h.half_items = 1;
            if (wino2h_eligible(h) && wino2h_items(h) >= tune_int("FS_WINO2H_MIN_ITEMS", 96)) on = L->wino_dh[i - 3] = 1;
        }
        if (const int w4 = L->wino_mode ? tune_int("FS_TNET_WINO4", 1) : 0) {   // the 16-tile F(4x4) kernel (see the forward units)
            ConvArgs h{};   // the 3x3 'full' conv of dz that unit_dgrad launches
            h.N = N;
            h.H = u.Hout;
            h.W = u.Wout;
            h.Cin = h.Cout = 64;
            h.Ho = u.Hin;
            h.Wo = u.Win;
            h.KH = h.KW = 3;
            h.stride = 1;
            h.pad_t = h.pad_l = 2;
            h.prof_tag = 1;
            h.tnet_plan = 1;   // (as unit_dgrad sets it: the planner keeps 16-tile items for the transform net)
            h.w_wino4t = reinterpret_cast<const float*>(16);
            if (wino4t_eligible(h) && (w4 == 2 || wino4t_items(h) >= tune_int("FS_WINO4T_MIN_ITEMS", 64))) {
                on = true;
                L->wino_dh[i - 3] = 2;
            }
        }
        L->wino_d[i - 3] = on ? b.take((size_t)(L->wino_dh[i - 3] == 2 ? 36 : 16) * 64 * 64) : 0;
    }
    L->inbwd = b.take(max_inbwd);
    {
        size_t max_rec = 0;
        for (int i = 2; i <= 11; ++i) {   // units whose output gradient a residual input-gradient launch writes (16 x 16-pixel items)
            const size_t r = (size_t)N * cdiv(L->u[i].Hout, 16) * cdiv(L->u[i].Wout, 16) * L->u[i].Cout * 2;
            if (r > max_rec) max_rec = r;
        }
        L->inb_rec = b.take(max_rec);
        for (int i = 0; i < 16; ++i) L->inb_S[i] = b.take((size_t)N * L->u[i].Cout * 2);
    }
    size_t slab_each[16];
    for (int i = 0; i < 16; ++i) {
        Unit& u = L->u[i];
        WgradArgs wa = unit_wgrad_args(u, N);
        u.wplan = wgrad_plan(wa);
        size_t sl = (size_t)u.wplan.n_slabs * u.wplan.K * wa.Cout;
        Wg2Args w2;
        if (const size_t f2 = wgrad2_plan(&wa, 1, &w2)) sl = f2;   // second-generation kernel where eligible (fs_wgrad2.hip)
        if (sl > max_slab) max_slab = sl;
        slab_each[i] = sl;
    }
    {   // the ten residual filter gradients as one launch: every dz is kept until the last one exists
        WgradArgs probs[10];
        for (int i = 3; i <= 12; ++i) probs[i - 3] = unit_wgrad_args(L->u[i], N);
        Wg2Args w2;
        const size_t f2 = tune_int("FS_TNET_WGRAD_BATCH", 1) ? wgrad2_plan(probs, 10, &w2) : 0;
        L->res_batch = f2 > 0;
        if (f2 > max_slab) max_slab = f2;
        WgwArgs ww;   // the Winograd filter-gradient kernel (fs_wgw.hip) takes the batch when it is eligible: its slabs are larger
        const size_t f3 = L->res_batch ? wgw_plan(probs, 10, &ww) : 0;
        if (f3 > max_slab) max_slab = f3;
        for (int i = 3; i <= 12; ++i)
            L->dzres[i - 3] = L->res_batch ? b.take((size_t)N * L->u[i].Hout * L->u[i].Wout * L->u[i].Cout) : 0;
    }
    L->slabs = b.take(max_slab);
    for (int i = 0; i < 16; ++i)   // (the residual units share L->slabs when their filter gradients run as one launch)
        L->slab_u[i] = (L->res_batch && i >= 3 && i <= 12) ? L->slabs : b.take(slab_each[i]);
    L->total_floats = b.off;
}

WgradArgs unit_wgrad_args(const Unit& u, int N) {
    WgradArgs a{};
    a.N = N;
    if (u.kind == 3) {  // roles swapped: the 'input' is the (larger) output gradient, the 'grad' the unit's input
        a.H = u.Hout;
        a.W = u.Wout;
        a.Cin = u.Cout;
        a.Ho = u.Hin;
        a.Wo = u.Win;
        a.Cout = u.Cin;
        a.KH = a.KW = u.K;
        a.stride = u.dstride;
        a.pad_t = u.dpad_t;
        a.pad_l = u.dpad_l;
        a.dil_x = 1;
        return a;
    }
    a.H = u.Hsrc;
    a.W = u.Wsrc;
    a.Cin = u.Cin;
    a.Ho = u.Hc;
    a.Wo = u.Wc;
    a.Cout = u.Cc;
    a.KH = u.K;
    a.KW = u.KWx;
    a.dil_x = u.dil_x;
    a.stride = u.stride;
    a.pad_t = u.pad_t;
    a.pad_l = u.pad_l;
    a.src_mode = u.src_mode;
    a.refl = u.refl;
    a.dy_unshuffle = u.kind == 1;
    return a;
}

// ---------------------------------------------------------------- forward
#define FS_TRY(x)            \
    do {                     \
        int rc_ = (x);       \
        if (rc_) return rc_; \
    } while (0)

// the input-gradient filters of the backward pass (flip-transposed / collapsed / phase-decomposed; the residual ones also Winograd-transformed)
static void bwd_filter_jobs(const TnetLayout& L, const float* params, float* ws, WtBatch* wb, WinoBatch* nb, WinoBatch* nb4) {
    for (int i = 1; i < 16; ++i) {
        const Unit& u = L.u[i];
        if (u.kind == 3) continue;  // conv2d_transpose units use the stored filter as is
        const int K = u.kind == 1 ? 3 : u.K;  // (a resize-conv unit carries its collapsed 2x2 tap count in K)
        if (u.kind == 0 && u.stride == 2)  // phase-decomposed stride-2 input gradient (KH/KW fields = forward pads)
            wb->add(WT_S2DGRAD, params + u.w_off, ws + L.wTu[i], u.pad_t, u.pad_l, u.Cin, u.Cout);
        else
            wb->add(u.kind == 1 ? WT_UPDGRAD : WT_FLIPT, params + u.w_off, ws + L.wTu[i], K, K, u.Cin, u.Cout);
    }
    for (int i = 3; i <= 12; ++i)   // ... and the Winograd transforms of the residual ones ([3][3][Cout][Cin] as the kernel's HWIO)
        if (L.wino_d[i - 3]) {
            WinoBatch& t = L.wino_dh[i - 3] == 2 ? *nb4 : *nb;
            t.w[t.n] = ws + L.wTu[i];
            t.U[t.n] = ws + L.wino_d[i - 3];
            ++t.n;
        }
}

int tnet_forward(const TnetLayout& L, const float* params, const float* x, float* y, float* ws, hipStream_t s, bool reuse_filters, bool with_bwd_filters) {
    const int N = L.N;
    // collapsed resize-conv filters (weights may have changed since the last call: training)
    WtBatch wb{};
    if (L.deconv) {  // filters are stored [K,K,Cout,Cin]: flip + transpose into the [tap][Cin][Cout] the kernel reads
        wb.add(WT_FLIPT, params + L.u[13].w_off, ws + L.weff[0], 3, 3, 32, 64);
        wb.add(WT_FLIPT, params + L.u[14].w_off, ws + L.weff[1], 3, 3, 16, 32);
        wb.add(WT_FLIPT, params + L.u[15].w_off, ws + L.wfold, 9, 9, 3, 16);
    } else {
        wb.add(WT_UPFWD, params + L.u[13].w_off, ws + L.weff[0], 3, 3, 64, 32);
        wb.add(WT_UPFWD, params + L.u[14].w_off, ws + L.weff[1], 3, 3, 32, 16);
        wb.add(WT_FOLD5FWD, params + L.u[15].w_off, ws + L.wfold, 9, 9, 16, 3);
    }
    if (!reuse_filters) {   // Winograd-transformed filters of the residual convs that use wino_conv_kernel (one launch)
        WinoBatch nb{}, nb4{}, nbd{};
        for (int i = 3; i <= 12; ++i)
            if (L.u[i].wino) {
                WinoBatch& t = L.u[i].wino == 3 ? nb4 : nb;   // (3: the register layout of fs_wino4t.hip)
                t.w[t.n] = params + L.u[i].w_off;
                t.U[t.n] = ws + L.u[i].wino_u;
                ++t.n;
            }
        // a training step: the backward's input-gradient filters ride in the same launches (two launches less per step)
        if (with_bwd_filters) bwd_filter_jobs(L, params, ws, &wb, &nbd, &nb4);
        FS_TRY(wt_batch(wb, s));
        FS_TRY(wino_gen().f2_second() ? wt_wino2_batch(nb, 64, 64, s) : wt_wino_batch(nb, 64, 64, s));
        if (with_bwd_filters) FS_TRY(wt_wino2_batch(nbd, 64, 64, s));
        FS_TRY(wt_wino4t_batch(nb4, 64, 64, s));
    }
    // units whose conv kernel is persistent (fs_wino2 / fs_wino2h / fs_cstream) and whose record count is small merge their
    // own instance-norm statistics (the last workgroup to finish; FinArgs in fs_kernels.h): no in_finalize launch
    bool fused[16];
    bool any_fused = false;
    for (int i = 0; i < 16; ++i) {
        const Unit& u = L.u[i];
        fused[i] = u.kind != 2 && (u.plan.variant == 6 || u.plan.variant == 7 || u.plan.variant == 8) && u.plan.rem_ks == 0 &&
                   fused_finalize_ok(N, u.Cout, u.tiles, u.kind == 1 ? 4 : 1);
        any_fused = any_fused || fused[i];
    }
    if (any_fused) FS_TRY(zero_words(ws + L.fin_counter, 16, s));   // (a kernel: memset nodes misbehave in single-stream graph replays, see fs_perceptual_loss)
    const float* src = x;
    const float* src_a = nullptr;
    const float* src_b = nullptr;
    for (int i = 0; i < 16; ++i) {
        const Unit& u = L.u[i];
        ConvArgs a = unit_args(u, N);
        a.p = u.plan;
        a.x = src;
        a.in_a = src_a;
        a.in_b = src_b;
        a.in_nstride = src_a ? u.Cin : 0;
        a.in_relu = src_a ? 1 : 0;
        a.w = (u.kind == 1 || (u.kind == 3 && i < 15)) ? ws + L.weff[i - 13] : ((u.kind == 2 || u.kind == 3) ? ws + L.wfold : params + u.w_off);
        a.w_wino = (u.wino && u.wino != 3 && !wino_gen().f2_second()) ? ws + u.wino_u : nullptr;
        a.w_wino2 = (u.wino && u.wino != 3 && wino_gen().f2_second()) ? ws + u.wino_u : nullptr;
        a.w_wino4t = u.wino == 3 ? ws + u.wino_u : nullptr;
        a.half_items = u.wino == 2;
        a.y = u.kind == 2 ? ws + L.zfold : ws + u.z;
        a.stats = u.kind == 2 ? nullptr : ws + u.stats;
        a.rem_ws = ws + L.rem_ws;
        a.rem_ws_floats = (size_t)kRemUnits * 16384;
        if (fused[i]) {
            a.fin.counter = reinterpret_cast<unsigned*>(ws + L.fin_counter) + i;
            a.fin.gamma = params + u.g_off;
            a.fin.beta = params + u.b_off;
            a.fin.mean = ws + u.mean;
            a.fin.rstd = ws + u.rstd;
            a.fin.a = ws + u.a;
            a.fin.b = ws + u.b;
            a.fin.T = u.tiles;
            a.fin.C = u.Cout;
            a.fin.groups = u.kind == 1 ? 4 : 1;
            a.fin.eps = 1e-3f;
        }
        FS_TRY(conv_launch(a, s));
        if (u.kind == 2)  // shifted 5-term sum of the virtual channels -> z + statistics partials
            FS_TRY(fold5_fwd(ws + L.zfold, ws + u.z, ws + u.stats, N, u.Hout, u.Wout, s));
        if (!fused[i])
        FS_TRY(in_finalize(ws + u.stats, N, u.tiles, u.Cout, u.kind == 1 ? 4 : 1, params + u.g_off, params + u.b_off, 1e-3f,
                           ws + u.mean, ws + u.rstd, ws + u.a, ws + u.b, s,
                           ws + u.stats + (size_t)N * u.tiles * (u.kind == 2 ? u.Cout : u.Cc) * 3));
        // what the next conv reads
        src = ws + u.z;
        src_a = ws + u.a;
        src_b = ws + u.b;
        if (i >= 4 && i <= 12 && ((i - 3) & 1)) {  // second conv of a residual block -> materialise h_k
            const int k = (i - 4) / 2;
            const float* skip;
            const float *sa = nullptr, *sb = nullptr;
            if (k == 0) {
                skip = ws + L.u[2].z;
                sa = ws + L.u[2].a;
                sb = ws + L.u[2].b;
            } else {
                skip = ws + L.h[k - 1];
            }
            FS_TRY(apply_res(ws + u.z, ws + u.a, ws + u.b, skip, sa, sb, k == 0, ws + L.h[k], N, u.Hout, u.Wout, 64, s));
            src = ws + L.h[k];
            src_a = src_b = nullptr;
        }
    }
    const Unit& u = L.u[15];
    return apply_tanh(ws + u.z, ws + u.a, ws + u.b, y, N, u.Hout * u.Wout, 3, s);
}

// ---------------------------------------------------------------- backward
// dgrad helper: d(input of unit u) from dz (gradient of u's raw conv output)
// below / below_relu / rec_tiles: the unit whose OUTPUT gradient `dst` is (its instance-norm backward runs next): when the launch is a
// 16-tile F(4x4) one its epilogue also leaves that unit's partial sums in ws + L.inb_rec and *rec_tiles = records per sample (else 0)
static int unit_dgrad(const TnetLayout& L, const Unit& u, const float* params, const float* dz, float* dst,
                      const float* add_src, float* ws, hipStream_t s, const Unit* below = nullptr, int below_relu = 0, int* rec_tiles = nullptr) {
    const int N = L.N;
    if (rec_tiles) *rec_tiles = 0;
    ConvArgs a{};
    a.prof_tag = 1;
    a.tnet_plan = 1;
    a.N = N;
    a.x = dz;
    a.y = dst;
    a.w = ws + L.wTu[&u - L.u];  // built at the start of tnet_backward
    {
        const int ui = (int)(&u - L.u);
        if (ui >= 3 && ui <= 12 && L.wino_d[ui - 3]) {
            if (L.wino_dh[ui - 3] == 2) {
                a.w_wino4t = ws + L.wino_d[ui - 3];
            } else {
                a.w_wino2 = ws + L.wino_d[ui - 3];
                a.half_items = L.wino_dh[ui - 3];
            }
        }
    }
    a.add_src = add_src;
    a.add_pad = add_src ? 2 : 0;
    if (u.kind == 3) {  // adjoint of conv2d_transpose = the plain strided conv with the stored filter
        a.w = params + u.w_off;
        a.H = u.Hout;
        a.W = u.Wout;
        a.Cin = u.Cout;
        a.Ho = u.Hin;
        a.Wo = u.Win;
        a.Cout = u.Cin;
        a.KH = a.KW = u.K;
        a.stride = u.dstride;
        a.pad_t = u.dpad_t;
        a.pad_l = u.dpad_l;
        a.src_mode = SRC_PLAIN;
    } else if (u.kind == 1) {  // resize-conv: 3x3 stride-2 conv over dY, pad 1 before
        a.H = u.Hout;
        a.W = u.Wout;
        a.Cin = u.Cout;
        a.Ho = u.Hin;
        a.Wo = u.Win;
        a.Cout = u.Cin;
        a.KH = a.KW = 3;
        a.stride = 2;
        a.pad_t = a.pad_l = 1;
        a.src_mode = SRC_PLAIN;
    } else if (u.stride == 2) {
        // 3x3 stride-2 conv: the four parities of the gradient pixel are four 1-, 2-, 2-, 4-tap convs over dz; run them
        // as ONE 2x2-tap conv with 4*Cin channels and a pixel-shuffle store (9 of the 16 tap/parity slots are
        // non-zero) instead of a 3x3 conv over the zero-dilated dz (where 3/4 of the multiplied inputs are zeros)
        a.H = u.Hout;
        a.W = u.Wout;
        a.Cin = u.Cout;
        a.Ho = u.Hout;
        a.Wo = u.Wout;
        a.Cout = 4 * u.Cin;
        a.KH = a.KW = 2;
        a.stride = 1;
        a.pad_t = u.pad_t == 0 ? 1 : 0;
        a.pad_l = u.pad_l == 0 ? 1 : 0;
        a.src_mode = SRC_PLAIN;
        a.shuffle = 1;
        a.shuf_H = u.Hin;
        a.shuf_W = u.Win;
    } else {
        a.H = u.Hout;
        a.W = u.Wout;
        a.Cin = u.Cout;
        a.Ho = u.Hin;
        a.Wo = u.Win;
        a.Cout = u.Cin;
        a.KH = a.KW = u.K;  // the input gradient uses the original square filter (also for the kw-folded unit)
        a.stride = 1;
        a.pad_t = u.K - 1 - u.pad_t;
        a.pad_l = u.K - 1 - u.pad_l;
        a.src_mode = u.stride == 2 ? SRC_DILATE2 : SRC_PLAIN;
    }
    a.rem_ws = ws + L.rem_ws;
    a.rem_ws_floats = (size_t)kRemUnits * 16384;
    a.p = conv_plan(a);
    if (below && rec_tiles && a.w_wino4t && a.p.variant == 11 && a.p.TW == 16 && a.p.ksplit <= 1 && a.Ho == below->Hout && a.Wo == below->Wout &&
        a.Cout == below->Cout && tune_int("FS_INBWD_FUSED", 1)) {
        ConvArgs f = a;
        f.inb_z = ws + below->z;
        f.inb_mean = ws + below->mean;
        f.inb_rstd = ws + below->rstd;
        f.inb_a = ws + below->a;
        f.inb_b = ws + below->b;
        f.inb_relu = below_relu;
        f.inb_rec = ws + L.inb_rec;
        const ConvPlan pf = conv_plan(f);
        if (pf.variant == 11 && pf.TW == 16 && pf.ksplit <= 1 && pf.tiles_y == a.p.tiles_y && pf.tiles_x == a.p.tiles_x) {
            f.p = pf;
            *rec_tiles = pf.tiles_y * pf.tiles_x;
            return conv_launch(f, s);
        }
    }
    return conv_launch(a, s);
}

// Filter gradient of one unit.  Problem description first (which tensor plays 'x', which 'dy', on-load affines), then the
// kernel: the persistent second-generation kernel where the shape is eligible, else the first-generation one.
static void unit_wgrad_problem(const TnetLayout& L, const Unit& u, const float* xin, const float* xa, const float* xb,
                               const float* dz, float* ws, WgradArgs* out) {
    WgradArgs a = unit_wgrad_args(u, L.N);
    a.p = u.wplan;
    if (u.kind == 3) {  // dW[K,K,Cout,Cin] = conv2d_bwd_filter(input = dz, grad = the unit's input activation)
        a.x = dz;
        a.dy = xin;
        a.dy_a = xa;
        a.dy_b = xb;
        a.dy_nstride = xa ? u.Cin : 0;
        a.dy_relu = xa ? 1 : 0;
    } else {
        a.x = xin;
        a.in_a = xa;
        a.in_b = xb;
        a.in_nstride = xa ? u.Cin : 0;
        a.in_relu = xa ? 1 : 0;
        a.dy = u.kind == 2 ? ws + L.zfold : dz;   // kind 2: dY unfolded to [q][(v,co)] by unfold5 (fs_fold.hip)
    }
    a.slabs = ws + L.slabs;
    *out = a;
}

// the filter re-layout behind a unit's reduced filter gradient (collapsed resize-conv -> 3x3, kw-folded output layer -> 9x9)
static int unit_wgrad_fold(const TnetLayout& L, const Unit& u, float* grads, float* ws, hipStream_t s) {
    if (u.kind == 2) return wt_fold5_back(ws + L.dwfold, grads + u.w_off, u.Cin, s);
    if (u.kind == 1) return wt_upconv_wgrad_fold(ws + ((&u - L.u) == 14 ? L.dweff2 : L.dweff), grads + u.w_off, u.Cin, u.Cout, s);
    return 0;
}

// defer: the slab reduction of a second-generation launch is appended there (one reduction launch for all units at the end of the backward);
// *fold_later = 1 then tells the caller to run the unit's filter re-layout (kind 1 / 2) behind that reduction
static int unit_wgrad(const TnetLayout& L, const Unit& u, const float* xin, const float* xa, const float* xb,
                      const float* dz, float* grads, float* ws, hipStream_t s, Wg2Reduce* defer = nullptr, int* fold_later = nullptr) {
    WgradArgs a;
    unit_wgrad_problem(L, u, xin, xa, xb, dz, ws, &a);
    float* const slabs = ws + L.slab_u[&u - L.u];
    a.slabs = slabs;
    if (fold_later) *fold_later = 0;
    if (u.kind == 2)  // the Z buffer of the forward is free by now
        FS_TRY(unfold5(dz, ws + L.zfold, L.N, u.Hout, u.Wout, s));
    // where the reduced gradient goes: the parameter-gradient buffer, or a scratch the unit's filter re-layout reads
    float* dst = u.kind == 2 ? ws + L.dwfold : (u.kind == 1 ? ws + ((&u - L.u) == 14 ? L.dweff2 : L.dweff) : grads + u.w_off);
    Wg2Args w2;
    if (wgrad2_plan(&a, 1, &w2)) {
        float* out[1] = {dst};
        const int before = defer ? defer->n : 0;
        FS_TRY(wgrad2_run(w2, slabs, out, 1.0f, s, defer));
        if (defer && defer->n > before && fold_later) {
            *fold_later = 1;
            return 0;
        }
    } else {
        FS_TRY(wgrad_launch(a, s));
        FS_TRY(reduce_slabs(slabs, 1, a.p.n_slabs, (size_t)a.p.K * a.Cout, 1.0f, dst, s));
    }
    return unit_wgrad_fold(L, u, grads, ws, s);
}

int tnet_backward(const TnetLayout& L, const float* params, const float* x, const float* dy, float* grads, float* ws,
                  hipStream_t s, const StreamAux* aux, bool filters_ready) {
    const int N = L.N;
    // (no fork while the per-kernel profiler is recording: two streams sharing the chip would charge each kernel with its
    // neighbour's time -- bench.py's per-kernel table wants every launch alone)
    // ... and no fork for small batches: at batch 4 the fork / join events cost more than the overlap returns (same lease, FS_NO_SIDE_STREAM = 0 / 1:
    // 1100 against 1133 images/s; at batch 32 the fork is worth 1.2 %).  FS_SIDE_MIN_PIXELS: smallest N * H * W that forks.
    const bool fork = aux && aux->side && aux->nev >= 34 && !Profiler::current() && (long)N * L.H * L.W >= (long)tune_int("FS_SIDE_MIN_PIXELS", 1000000);
    hipStream_t ws_stream = fork ? aux->side : s;  // stream of the filter-gradient branch
    if (!filters_ready) {  // every input-gradient filter of the step in one launch (a forward with FS_FLAG_SAVE_FOR_BWD has built them already)
        WtBatch wb{};
        WinoBatch nb{}, nb4{};
        bwd_filter_jobs(L, params, ws, &wb, &nb, &nb4);
        FS_TRY(wt_batch(wb, s));
        FS_TRY(wt_wino2_batch(nb, 64, 64, s));
        FS_TRY(wt_wino4t_batch(nb4, 64, 64, s));
    }
    Wg2Reduce red{};               // the slab reductions of the non-residual units' filter gradients: ONE launch behind the last of them
    int fold_units[16], n_fold = 0;
    const bool defer_on = tune_int("FS_WGRAD_DEFER", 1) != 0;
    const float* g = dy;         // gradient wrt the current unit's output activation (or h_k)
    const float* res_g = nullptr;  // d h_k, kept alive until the block's first conv adds it to its dgrad
    int dz_reader[2] = {-1, -1};   // event (side stream) of the last filter gradient that read dz[0] / dz[1]
    int rec_tiles = 0;             // > 0: the launch that wrote g left the unit's instance-norm-backward records in ws + L.inb_rec (per sample)
    InbParams ibp{};               // units whose dgamma / dbeta come from in_bwd_params at the end
    ibp.N = N;
    WgradArgs res_probs[10];       // the residual units' filter-gradient problems, launched together after unit 3
    for (int i = 15; i >= 0; --i) {
        const Unit& u = L.u[i];
        const bool batched = L.res_batch && i >= 3 && i <= 12;
        float* dz = batched ? ws + L.dzres[i - 3] : ws + L.dz[i & 1];
        // dz[i&1] was last read by a filter gradient on the side stream
        if (fork && !batched && dz_reader[i & 1] >= 0 && hipStreamWaitEvent(s, aux->ev[dz_reader[i & 1]], 0) != hipSuccess) return -20;
        const bool res2 = i >= 4 && i <= 12 && ((i - 3) & 1);       // second conv of a block (no activation)
        const bool res1 = i >= 3 && i <= 11 && ((i - 3) & 1) == 0;  // first conv of a block
        if (res2) res_g = g;
        const int mode = i == 15 ? 2 : (res2 ? 0 : 1);
        {
            // per-sample sums from records (the producer's epilogue, or a partial-sum pass here), reduced in the apply kernel's prologue;
            // 1: shape not taken -> the three-launch form
            int rc = in_bwd_rec(g, ws + u.z, ws + u.mean, ws + u.rstd, ws + u.a, ws + u.b, mode, dz, rec_tiles > 0 ? ws + L.inb_rec : nullptr, rec_tiles,
                                ws + L.inb_S[i], ws + L.inbwd, N, u.Hout * u.Wout, u.Cout, s);
            if (rc == 0) {
                InbParams::U& q = ibp.u[ibp.n++];
                q.S = ws + L.inb_S[i];
                q.dgamma = grads + u.g_off;
                q.dbeta = grads + u.b_off;
                q.C = u.Cout;
            } else if (rc == 1) {
                rc = in_bwd(g, ws + u.z, ws + u.mean, ws + u.rstd, ws + u.a, ws + u.b, mode, dz, grads + u.g_off, grads + u.b_off, ws + L.inbwd, N,
                            u.Hout * u.Wout, u.Cout, s);
            }
            FS_TRY(rc);
            rec_tiles = 0;
        }
        // the tensor this unit's conv consumed
        const float* xin;
        const float *xa = nullptr, *xb = nullptr;
        if (i == 0) {
            xin = x;
        } else if (i >= 3 && i <= 13 && ((i - 3) & 1) == 0) {  // first conv of block k, or upsample_0 (k=5)
            const int k = (i - 3) / 2;
            if (k == 0) {
                xin = ws + L.u[2].z;
                xa = ws + L.u[2].a;
                xb = ws + L.u[2].b;
            } else {
                xin = ws + L.h[k - 1];
            }
        } else {
            xin = ws + L.u[i - 1].z;
            xa = ws + L.u[i - 1].a;
            xb = ws + L.u[i - 1].b;
        }
        if (batched) {
            unit_wgrad_problem(L, u, xin, xa, xb, dz, ws, &res_probs[i - 3]);
            if (i == 3) {   // every residual dz exists: ten filter gradients, one launch + one reduction
                if (fork && (hipEventRecord(aux->ev[i], s) != hipSuccess || hipStreamWaitEvent(ws_stream, aux->ev[i], 0) != hipSuccess))
                    return -20;
                float* outs[10];
                for (int k = 0; k < 10; ++k) outs[k] = grads + L.u[3 + k].w_off;
                WgwArgs ww;
                if (wgw_plan(res_probs, 10, &ww)) {   // Winograd F(3x3, 2x2) filter gradients (fs_wgw.hip)
                    FS_TRY(wgw_run(ww, ws + L.slabs, outs, 1.0f, ws_stream));
                } else {
                    Wg2Args w2;
                    if (!wgrad2_plan(res_probs, 10, &w2)) return -21;
                    FS_TRY(wgrad2_run(w2, ws + L.slabs, outs, 1.0f, ws_stream));
                }
                if (fork && hipEventRecord(aux->ev[16 + i], ws_stream) != hipSuccess) return -20;
            }
        } else {
            if (fork) {  // dz_i ready -> side stream computes dW_i while this stream continues with the input gradient
                if (hipEventRecord(aux->ev[i], s) != hipSuccess || hipStreamWaitEvent(ws_stream, aux->ev[i], 0) != hipSuccess) return -20;
            }
            int fold_later = 0;
            FS_TRY(unit_wgrad(L, u, xin, xa, xb, dz, grads, ws, ws_stream, defer_on ? &red : nullptr, &fold_later));
            if (fold_later) fold_units[n_fold++] = i;
            if (fork) {
                if (hipEventRecord(aux->ev[16 + i], ws_stream) != hipSuccess) return -20;
                dz_reader[i & 1] = 16 + i;
            }
        }
        if (i == 0) break;
        float* dst = nullptr;
        for (int k = 0; k < 3; ++k) {
            float* cand = ws + L.g[k];
            if (cand != g && cand != res_g) {
                dst = cand;
                break;
            }
        }
        {
            const int j = i - 1;   // the unit whose instance-norm backward consumes dst next
            const bool j_res2 = j >= 4 && j <= 12 && ((j - 3) & 1);
            FS_TRY(unit_dgrad(L, u, params, dz, dst, res1 ? res_g : nullptr, ws, s, &L.u[j], j_res2 ? 0 : 1, &rec_tiles));
        }
        if (res1) res_g = nullptr;
        g = dst;
    }
    FS_TRY(in_bwd_params(ibp, s));
    FS_TRY(wgrad2_reduce(red, ws_stream));
    for (int k = 0; k < n_fold; ++k) FS_TRY(unit_wgrad_fold(L, L.u[fold_units[k]], grads, ws, ws_stream));
    if (fork) {  // join: every filter gradient (and the reduction above) done before the caller's stream proceeds
        if (hipEventRecord(aux->ev[32], ws_stream) != hipSuccess || hipStreamWaitEvent(s, aux->ev[32], 0) != hipSuccess) return -20;
    }
    return 0;
}

}  // namespace fs
