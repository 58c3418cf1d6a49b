// VGG16 feature extraction (reference libs/vgg16.py:36-220), Gram matrices (utils.py:66-83),
// content/style/TV losses (losses.py:12-97) and the gradient of their sum wrt the VGG input
// (the VGG half of train.py:203's gradient graph; VGG itself is frozen, train.py:198-199).
//
// Launch sequence of one fs_perceptual_loss call:
//   [y ; content] -> conv1_1 .. conv(cmax)        one 2N batch through the shared layers
//   y only        -> .. conv(lmax)
//   per style layer: G = F^T F/(hwc) (fs_wgrad per-sample 1x1), loss + S = coef*(G-Gt)
//   per content layer: loss + tap gradient
//   backward lmax..0: route (pool argmax, ReLU mask, + tap gradients incl. dF = F*S as a 1x1 conv
//   with per-sample filters) -> 3x3 dgrad conv with the pre-flipped filters -> ... -> dy (+ beta*dTV)
#include "fs_vgg.h"
#include "fs_tnet.h"   // StreamAux

#include <cstdlib>

#include <cstring>

namespace fs {

static const int kCin[FS_VGG_NLAYERS] = {3, 64, 64, 128, 128, 256, 256, 256, 512, 512};
static const int kCout[FS_VGG_NLAYERS] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512};
static inline bool pool_after(int l) { return l == 1 || l == 3 || l == 6; }
static inline int pool_index(int l) { return l == 1 ? 0 : (l == 3 ? 1 : 2); }

#define FS_TRY(x)            \
    do {                     \
        int rc_ = (x);       \
        if (rc_) return rc_; \
    } while (0)

// The prepared buffer (fs_vgg_prepare, once per frozen weight set): [flip-transposed filters of every layer (input-gradient convs)] followed by the
// Winograd-transformed filters, forward and flip-transposed, of layers 1.., in the layouts of the kernel GENERATIONS the current knobs select -- round 6:
// only those (rounds 2-5 built every generation's layout, 490 MB, although a run reads one):
//   PREP_F2   F(2x2,3x3): U[16][Cin][Cout] (fs_wino.hip) and its K-contiguous form (fs_wino2.hip)     FS_WINO_V <= 3
//   PREP_W4   F(4x4,3x3), filter through LDS (fs_wino4.hip)                                             FS_WINO_V = 4
//   PREP_W4T  F(4x4,3x3), filter global -> registers (fs_wino4t.hip) + its 128-channel item form        FS_WINO_V >= 5
//   PREP_W6   three bf16 pieces for the split-bf16 pipeline (fs_wino6.hip), eligible layers only        FS_WINO_V >= 6 (default)
// FS_VGG_PREPARE_ALL=1 builds all of them (generation cross-checks inside one process).  vgg_prepare returns the mask it built; fs_api.hip remembers it
// for that buffer and hands it to every consumer, which reads a generation only if the buffer has it AND the knobs of the moment want it -- a buffer
// prepared under other knobs degrades to the direct kernels, never to a read of bytes that were not written.
enum { PREP_F2 = 1, PREP_W4 = 2, PREP_W4T = 4, PREP_W6 = 8 };
unsigned vgg_prep_mask() {
    if (tune_int("FS_VGG_PREPARE_ALL", 0)) return PREP_F2 | PREP_W4 | PREP_W4T | PREP_W6;
    const WinoGen g = wino_gen();
    return g.split_bf16() ? (PREP_W4T | PREP_W6) : g.f4_reg() ? PREP_W4T : g.f4_lds() ? PREP_W4 : PREP_F2;
}
static bool w6_layer_ok(int l, bool dgrad) {
    const int ci = dgrad ? kCout[l] : kCin[l], co = dgrad ? kCin[l] : kCout[l];
    return l >= 1 && ci % 32 == 0 && co % 128 == 0 && (long)ci * co >= (long)tune_int("FS_WINO6_MINCC", 512 * 256);
}
struct PrepLayout {
    unsigned mask;
    size_t flipt[FS_VGG_NLAYERS];
    size_t wino[FS_VGG_NLAYERS][2], wino2[FS_VGG_NLAYERS][2], wino4[FS_VGG_NLAYERS][2], wino4t[FS_VGG_NLAYERS][2], wino4u[FS_VGG_NLAYERS][2], wino6[FS_VGG_NLAYERS][2];   // [layer][dgrad]
    size_t total;
};
static PrepLayout prep_layout(unsigned mask) {
    PrepLayout P{};
    P.mask = mask;
    size_t n = 0;
    auto take = [&](size_t floats) {
        const size_t o = n;
        n += (floats + 63) & ~(size_t)63;
        return o;
    };
    for (int l = 0; l < FS_VGG_NLAYERS; ++l) P.flipt[l] = take((size_t)9 * kCin[l] * kCout[l]);
    for (int l = 1; l < FS_VGG_NLAYERS; ++l)
        for (int d = 0; d < 2; ++d) {
            const size_t cc = (size_t)kCin[l] * kCout[l];
            if (mask & PREP_F2) {
                P.wino[l][d] = take(16 * cc);
                P.wino2[l][d] = take(16 * cc);
            }
            if (mask & PREP_W4) P.wino4[l][d] = take(36 * cc);
            if (mask & PREP_W4T) {
                P.wino4t[l][d] = take(36 * cc);
                if ((d ? kCin[l] : kCout[l]) % 128 == 0) P.wino4u[l][d] = take(36 * cc);
            }
            if ((mask & PREP_W6) && w6_layer_ok(l, d == 1)) P.wino6[l][d] = take(wino6_filter_floats(d ? kCout[l] : kCin[l], d ? kCin[l] : kCout[l]));
        }
    P.total = n;
    return P;
}
size_t vgg_prepared_floats() { return prep_layout(vgg_prep_mask()).total; }
// which generation the knobs of the moment want for the F(4x4) convs (FS_WINO_V >= 5: fs_wino4t.hip; 4: fs_wino4.hip) and whether the deep layers may take fs_wino6.hip
static bool vgg_use_4t() { return wino_gen().f4_reg(); }
static bool vgg_want_w6() { return wino_gen().split_bf16(); }
// debugging aid: FS_VGG_WINO_MASK selects the layers that may take the Winograd kernel (bit l: forward of layer l,
// bit 16+l: its input gradient); default all
static bool wino_layer_on(int bit) {
    return ((unsigned)tune_int("FS_VGG_WINO_MASK", -1) >> bit) & 1u;
}

int vgg_prepare(const float* const w[FS_VGG_NLAYERS], float* prepared, hipStream_t s) {
    const PrepLayout P = prep_layout(vgg_prep_mask());
    for (int l = 0; l < FS_VGG_NLAYERS; ++l) FS_TRY(wt_flip_transpose(w[l], prepared + P.flipt[l], 3, 3, kCin[l], kCout[l], s));
    for (int l = 1; l < FS_VGG_NLAYERS; ++l)
        for (int d = 0; d < 2; ++d) {
            // forward: the stored HWIO filter; input gradient: its flip-transposed copy [3][3][Cout][Cin] with the channel roles swapped
            const float* src = d ? prepared + P.flipt[l] : w[l];
            const int ci = d ? kCout[l] : kCin[l], co = d ? kCin[l] : kCout[l];
            if (P.mask & PREP_F2) {
                FS_TRY(wt_wino(src, prepared + P.wino[l][d], ci, co, s));
                FS_TRY(wt_wino2(src, prepared + P.wino2[l][d], ci, co, s));
            }
            if (P.mask & PREP_W4) FS_TRY(wt_wino4(src, prepared + P.wino4[l][d], ci, co, s));
            if (P.mask & PREP_W4T) {
                FS_TRY(wt_wino4t(src, prepared + P.wino4t[l][d], ci, co, s));
                if (co % 128 == 0) FS_TRY(wt_wino4u(src, prepared + P.wino4u[l][d], ci, co, s));
            }
            if ((P.mask & PREP_W6) && w6_layer_ok(l, d == 1)) FS_TRY(wt_wino6(src, reinterpret_cast<unsigned short*>(prepared + P.wino6[l][d]), ci, co, s));
        }
    return (int)P.mask;   // >= 0: the generations the buffer carries
}

struct Bump2 {
    size_t off = 0;
    size_t take(size_t floats) {
        const size_t o = off;
        off += (floats + 63) & ~(size_t)63;
        return o;
    }
};

static WgradArgs gram_args(int N, int H, int W, int C) {
    WgradArgs a{};
    a.N = N;
    a.H = a.Ho = H;
    a.W = a.Wo = W;
    a.Cin = a.Cout = C;
    a.KH = a.KW = 1;
    a.stride = 1;
    a.per_sample = 1;
    return a;
}

void vgg_layout(int N, int H, int W, const fs_loss_cfg& cfg, bool with_content, VggLayout* L) {
    memset(L, 0, sizeof(*L));
    L->N = N;
    L->H = H;
    L->W = W;
    int lmax = 0, cmax = -1;
    for (int i = 0; i < cfg.n_style; ++i)
        if (cfg.style_layer[i] > lmax) lmax = cfg.style_layer[i];
    for (int i = 0; i < cfg.n_content; ++i) {
        if (cfg.content_layer[i] > lmax) lmax = cfg.content_layer[i];
        if (cfg.content_layer[i] > cmax) cmax = cfg.content_layer[i];
        if (with_content) L->content_mask |= 1u << cfg.content_layer[i];
    }
    if (!with_content) cmax = -1;
    L->lmax = lmax;
    L->cmax = cmax;
    L->NB = cmax >= 0 ? 2 * N : N;
    Bump2 b;
    L->xin = b.take((size_t)L->NB * H * W * 3);
    L->ab = b.take(8);
    int h = H, w = W;
    size_t max_act = 0, max_slab = 0;
    for (int l = 0; l <= lmax; ++l) {
        L->Hl[l] = h;
        L->Wl[l] = w;
        const int nb = l <= cmax ? L->NB : N;
        L->act[l] = b.take((size_t)nb * h * w * kCout[l]);
        const size_t grad_act = (size_t)N * h * w * kCout[l];
        if (grad_act > max_act) max_act = grad_act;
        if (pool_after(l) && l < lmax) {
            h = (h + 1) / 2;
            w = (w + 1) / 2;
            L->pool[pool_index(l)] = b.take((size_t)nb * h * w * kCout[l]);
        }
    }
    for (int i = 0; i < cfg.n_style; ++i) {
        const int l = cfg.style_layer[i], C = kCout[l];
        L->gram[i] = b.take((size_t)N * C * C);
        L->sm[i] = b.take((size_t)N * C * C);
        WgradArgs ga = gram_args(N, L->Hl[l], L->Wl[l], C);
        const WgradPlan p = wgrad_plan(ga);
        const size_t sl = (size_t)N * p.n_slabs * C * C;
        if (sl > max_slab) max_slab = sl;
        const size_t sl2 = gram2_slab_floats(N, L->Hl[l] * L->Wl[l], C);   // the streaming kernel's partial slabs (fs_gram.hip)
        if (sl2 > max_slab) max_slab = sl2;
        L->gslab[i] = b.take(sl2 ? sl2 : 4);
    }
    L->slabs = b.take(max_slab);
    {   // room for every term's partial sums: a style term up to max(its finish launch's blocks, 1024), content terms and the TV term 1024 each
        size_t n = 1024;
        for (int i = 0; i < cfg.n_style; ++i) {
            const size_t g = (size_t)gram2_finish_partials(N, kCout[cfg.style_layer[i]]);
            n += g > 1024 ? g : 1024;
        }
        n += (size_t)1024 * (cfg.n_content > 0 ? cfg.n_content : 0);
        L->lossp_floats = n;
        L->lossp = b.take(n);
    }
    L->d_pre = b.take(max_act);
    L->d_in[0] = b.take(max_act);
    L->d_in[1] = b.take(max_act);
    L->d_tap = b.take(max_act);
    L->d_tap2 = b.take(max_act);
    L->scratch = b.take(2048);
    // split-K scratch: plan every VGG conv (forward at its batch, dgrad at N) with unlimited scratch and keep the max need
    size_t need = 0;
    for (int l = 0; l <= lmax; ++l)
        for (int dir = 0; dir < 2; ++dir) {
            if (dir == 1 && l == 0) continue;
            ConvArgs a{};
            a.N = dir == 0 ? (l <= cmax ? L->NB : N) : N;
            a.H = a.Ho = L->Hl[l];
            a.W = a.Wo = L->Wl[l];
            a.Cin = dir == 0 ? kCin[l] : kCout[l];
            a.Cout = dir == 0 ? kCout[l] : kCin[l];
            a.KH = a.KW = 3;
            a.stride = 1;
            a.pad_t = a.pad_l = 1;
            a.split_ws = reinterpret_cast<float*>(16);
            a.split_ws_floats = ~(size_t)0;
            const ConvPlan p = conv_plan(a);
            const size_t f = p.ksplit > 1 ? (size_t)p.ksplit * a.N * a.Ho * a.Wo * a.Cout : 0;
            if (f > need) need = f;
        }
    L->splitws_floats = need;
    L->splitws = b.take(need ? need : 4);
    // scratch of the split-bf16 pipeline (FS_WINO_V=6): V + M of the largest launch that may take it, one pass (capped: beyond the cap the launch runs in tile chunks)
    size_t need6 = 0;
    if (vgg_want_w6())
        for (int l = 1; l <= lmax; ++l)
            for (int dir = 0; dir < 2; ++dir) {
                if (!w6_layer_ok(l, dir == 1)) continue;
                const size_t f = wino6_ws_floats(dir == 0 ? (l <= cmax ? L->NB : N) : N, L->Hl[l], L->Wl[l], dir == 0 ? kCin[l] : kCout[l], dir == 0 ? kCout[l] : kCin[l]);
                if (f > need6) need6 = f;
            }
    const size_t cap6 = (size_t)tune_int("FS_WINO6_WS_MB", 2048) * (1u << 18);   // floats
    if (need6 > cap6) need6 = cap6;
    L->w6ws_floats = need6;
    L->w6ws = b.take(need6 ? need6 : 4);
    L->total_floats = b.off;
}


// Two half-batch chains on two streams (round 6, the split-bf16 pipeline only).  A run of consecutive layers that all take fs_wino6.hip is launched as chain A
// (samples [0, N/2)) on the caller's stream and chain B (the other half) on the side stream: the layers of a chain depend on each other, the chains do not, so
// the memory-bound input / output transforms of one chain run beside the GEMM of the other (their waves are held to <= 168 registers: they fit on a SIMD
// next to a resident GEMM wave).  Measured on three conv4_2-shaped layers at batch 32: 355 -> 284 us per layer (tools/micro_wino6_overlap.py).  Off while the
// per-kernel profiler runs (its event pairs would time overlapping launches), for odd N, and under FS_WINO6_CHAINS=0.
static bool w6_chains_on(const StreamAux* aux, int N) {
    return aux && aux->side && aux->nev >= 2 && !(N & 1) && !Profiler::current() && tune_int("FS_WINO6_OVERLAP", 1) == 1;
}
// ... or every launch on its own as two tile chunks pipelined over the two streams (fs_wino6.hip; FS_WINO6_OVERLAP = 2; measured SLOWER than no overlap in the step -- two GEMM launches of 4.5 grid rounds each and three more graph edges per launch --, kept as a recorded experiment; 0: neither; 1, the default: the chains above)
static void w6_pipe_args(ConvArgs* a, const StreamAux* aux) {
    if (aux && aux->side && aux->nev >= 3 && !Profiler::current() && tune_int("FS_WINO6_OVERLAP", 1) == 2) {
        a->w6_side = aux->side;
        a->w6_ev = aux->ev;
    }
}
// (a half of at least FS_WINO6_CHAIN_MINTILES tiles, default 1024 = conv4_x at batch 32, where the chains were measured; below it -- batch 4 per GPU: 256 tiles
// per launch -- ONE chain: 3.283 -> 3.239 ms per batch-4 step against the fp32 kernel's split-K launches, profiles/r06_ab_wino6_exact_waits.txt)
static bool w6_chain_launch_ok(const ConvArgs& a) {
    return a.p.variant == 12 && !(a.N & 1) && !a.y_keep_n && (long)(a.N / 2) * cdiv(a.Ho, 4) * cdiv(a.Wo, 4) >= (long)tune_int("FS_WINO6_CHAIN_MINTILES", 1024) &&
           a.w6_ws_floats >= 2 * wino6_ws_floats(a.N / 2, a.Ho, a.Wo, a.Cin, a.Cout);
}
struct W6Chains {   // fork / join bookkeeping of one run
    const StreamAux* aux;
    hipStream_t s;
    bool open = false;
    int begin() {
        if (open) return 0;
        if (hipEventRecord(aux->ev[0], s) != hipSuccess || hipStreamWaitEvent(aux->side, aux->ev[0], 0) != hipSuccess) return -20;
        open = true;
        return 0;
    }
    int end() {
        if (!open) return 0;
        open = false;
        return (hipEventRecord(aux->ev[1], aux->side) != hipSuccess || hipStreamWaitEvent(s, aux->ev[1], 0) != hipSuccess) ? -20 : 0;
    }
    int launch(const ConvArgs& a);
};
static ConvArgs w6_half(const ConvArgs& a, int h) {   // samples [h N/2, (h + 1) N/2) of a launch whose tensors are all batch-major
    ConvArgs b = a;
    const int n2 = a.N / 2;
    b.N = n2;
    const size_t xin = (size_t)n2 * a.H * a.W * a.Cin, yout = (size_t)n2 * a.Ho * a.Wo * a.Cout;
    if (h) {
        b.x = a.x + xin;
        b.y = a.y + yout;
        if (a.mask_src) b.mask_src = a.mask_src + yout;
        if (a.add_src) b.add_src = a.add_src + yout;
        if (a.pool_out) b.pool_out = a.pool_out + (size_t)n2 * (a.Ho >> 1) * (a.Wo >> 1) * a.Cout;
    }
    const size_t half_ws = (a.w6_ws_floats / 2) & ~(size_t)63;
    b.w6_ws = a.w6_ws + (h ? half_ws : 0);
    b.w6_ws_floats = half_ws;
    return b;
}

int W6Chains::launch(const ConvArgs& a) {
    ConvArgs hb[2];
    for (int h = 0; h < 2; ++h) {
        hb[h] = w6_half(a, h);
        hb[h].p = conv_plan(hb[h]);
    }
    if (hb[0].p.variant != 12 || hb[1].p.variant != 12) {   // (a half the pipeline does not take -- knobs that disagree: FS_WINO6_CHAIN_MINTILES below FS_WINO6_MINTILES): the whole launch on the caller's stream
        FS_TRY(end());
        return conv_launch(a, s);
    }
    FS_TRY(begin());
    for (int h = 0; h < 2; ++h) FS_TRY(conv_launch(hb[h], h ? aux->side : s));
    return 0;
}

// pool: optional destination of the 2x2/2 max-pool of the result; *pooled tells whether the conv launch produced it
static int vgg_conv(const float* x, int N, int H, int W, int l, const float* w, const float* w_wino, const float* w_wino2, const float* w_wino4, const float* w_wino4t, const float* w_wino4u, const float* bias, const float* ab,
                    float* y, float* split_ws, size_t split_ws_floats, hipStream_t s, float* pool = nullptr, bool* pooled = nullptr, int y_keep_n = 0,
                    const unsigned short* w_wino6 = nullptr, float* w6_ws = nullptr, size_t w6_ws_floats = 0, W6Chains* chains = nullptr, const StreamAux* aux = nullptr) {
    ConvArgs a{};
    if (w_wino6) w6_pipe_args(&a, aux);
    a.w_wino6 = w_wino6;
    a.w6_ws = w6_ws;
    a.w6_ws_floats = w6_ws_floats;
    a.x = x;
    a.w = w;
    a.w_wino = w_wino;
    a.w_wino2 = w_wino2;
    a.w_wino4 = w_wino4;
    a.w_wino4t = w_wino4t;
    a.w_wino4u = w_wino4u;
    a.y = y;
    a.N = N;
    a.H = a.Ho = H;
    a.W = a.Wo = W;
    a.Cin = kCin[l];
    a.Cout = kCout[l];
    a.KH = a.KW = 3;
    a.stride = 1;
    a.pad_t = a.pad_l = 1;
    a.bias = bias;
    a.out_relu = 1;
    a.split_ws = split_ws;
    a.split_ws_floats = split_ws_floats;
    if (l == 0) {  // images - mean folded into the load (padding stays zero, as in TF)
        a.in_a = ab;
        a.in_b = ab + 4;
    }
    a.p = conv_plan(a);
    if (pooled) *pooled = false;
    if (pool && (a.p.variant == 5 || a.p.variant == 6 || a.p.variant == 10 || a.p.variant == 11 || a.p.variant == 12) && a.p.ksplit <= 1 && !(H & 1) && !(W & 1) && tune_int("FS_VGG_POOL_FUSED", 1)) {
        a.pool_out = pool;   // the Winograd epilogues hold whole 2x2 tiles: the pooled tensor comes for one extra store per tile
        if (pooled) *pooled = true;
        if ((a.p.variant == 11 || a.p.variant == 12) && tune_int("FS_VGG_SKIP_CONTENT_Y", 1)) a.y_keep_n = y_keep_n;
    }
    if (chains) {   // (a run of split-bf16 launches goes out as two half-batch chains; anything else closes the run first)
        if (w6_chain_launch_ok(a)) return chains->launch(a);
        FS_TRY(chains->end());
    }
    return conv_launch(a, s);
}

// forward through layers [0..lmax]; samples [0,N) go all the way, [N,NB) stop after cmax
// prepared: the buffer of fs_vgg_prepare (Winograd-transformed filters) or nullptr (direct convolutions only)
static int vgg_forward(const VggLayout& L, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS],
                       const float* prepared, float* ws, hipStream_t s, unsigned prep_mask = 0, const StreamAux* aux = nullptr) {
    FS_TRY(vgg_consts(ws + L.ab, s));
    const PrepLayout P = prep_layout(prepared ? prep_mask : 0);
    W6Chains chains{aux, s};
    const bool use_chains = w6_chains_on(aux, L.N);
    const float* src = ws + L.xin;
    for (int l = 0; l <= L.lmax; ++l) {
        bool pooled = false;
        const int nb = l <= L.cmax ? L.NB : L.N;
        const bool wl = prepared && l >= 1 && wino_layer_on(l);
        const bool f2 = wl && (P.mask & PREP_F2), w4 = wl && (P.mask & PREP_W4) && !vgg_use_4t(), w4t = wl && (P.mask & PREP_W4T) && vgg_use_4t();
        const bool w6 = wl && (P.mask & PREP_W6) && vgg_want_w6() && L.w6ws_floats && w6_layer_ok(l, false);
        FS_TRY(vgg_conv(src, nb, L.Hl[l], L.Wl[l], l, w[l], f2 ? prepared + P.wino[l][0] : nullptr, f2 ? prepared + P.wino2[l][0] : nullptr,
                        w4 ? prepared + P.wino4[l][0] : nullptr, w4t ? prepared + P.wino4t[l][0] : nullptr,
                        (w4t && kCout[l] % 128 == 0) ? prepared + P.wino4u[l][0] : nullptr, b[l],
                        ws + L.ab, ws + L.act[l], ws + L.splitws, L.splitws_floats, s,
                        (pool_after(l) && l < L.lmax) ? ws + L.pool[pool_index(l)] : nullptr, &pooled,
                        // (the content half [N, NB) only feeds the next layer: of a pooled layer below the LAST content layer it needs the pooled tensor
                        // alone -- unless a content term of its own reads the full-resolution half, --loss_content_layers takes several)
                        (nb > L.N && l < L.cmax && !((L.content_mask >> l) & 1u)) ? L.N : 0,
                        w6 ? reinterpret_cast<const unsigned short*>(prepared + P.wino6[l][0]) : nullptr, ws + L.w6ws, L.w6ws_floats,
                        (use_chains && nb == L.N) ? &chains : nullptr, aux));
        src = ws + L.act[l];
        if (pool_after(l) && l < L.lmax) {
            FS_TRY(chains.end());
            if (!pooled)
                FS_TRY(maxpool(ws + L.act[l], ws + L.pool[pool_index(l)], nb, L.Hl[l], L.Wl[l], kCout[l], s));
            src = ws + L.pool[pool_index(l)];
        }
    }
    return chains.end();
}

static int gram_forward(const VggLayout& L, int l, const float* F, float* G, float* ws, hipStream_t s) {
    const int C = kCout[l], H = L.Hl[l], W = L.Wl[l];
    if (gram2_eligible(L.N, H * W, C))   // streaming kernel (fs_gram.hip)
        return gram2_launch(F, G, ws + L.slabs, L.N, H * W, C, 1.0f / ((float)H * W * C), s);
    WgradArgs ga = gram_args(L.N, H, W, C);
    ga.x = F;
    ga.dy = F;
    ga.slabs = ws + L.slabs;
    ga.p = wgrad_plan(ga);
    FS_TRY(wgrad_launch(ga, s));
    return reduce_slabs(ws + L.slabs, L.N, ga.p.n_slabs, (size_t)C * C, 1.0f / ((float)H * W * C), G, s);
}

int vgg_features(const VggLayout& L, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS], const float* x,
                 int n_layers, const int* layers, float* const* out, float* ws, hipStream_t s) {
    if (hipMemcpyAsync(ws + L.xin, x, (size_t)L.N * L.H * L.W * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return -10;
    FS_TRY(vgg_forward(L, w, b, nullptr, ws, s));
    for (int i = 0; i < n_layers; ++i) {
        const int l = layers[i];
        const size_t bytes = (size_t)L.N * L.Hl[l] * L.Wl[l] * kCout[l] * sizeof(float);
        if (hipMemcpyAsync(out[i], ws + L.act[l], bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return -10;
    }
    return 0;
}

// Adjoint of vgg_features (round 6; the pieces train.py:198-204 / slow_style.py:140-176 differentiate through when a script composes its own objective):
// dfeat[i] = dL/d(post-ReLU activation of layers[i]) [N,Hl,Wl,C] for any subset of layers -> dx = dL/d(images) [N,H,W,3].  The forward is recomputed
// into the workspace (VGG is frozen; with `prepared` on the Winograd kernels of fs_perceptual_loss, without on the direct ones), then the backward of
// perceptual_loss with the given tensors as tap gradients: per layer the 3x3 input-gradient conv with the flip-transposed filter, the tap added and the ReLU
// mask applied in its epilogue, the 2x2 max-pool gradient routed to the first maximum (TF MaxPoolGrad) by vgg_bwd_route.
int vgg_dgrad(const VggLayout& L, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS], const float* prepared, unsigned prep_mask,
              const float* x, int n_layers, const int* layers, const float* const* dfeat, float* dx, float* ws, float* flipt_scratch, hipStream_t s) {
    const int N = L.N;
    if (x != ws + L.xin && hipMemcpyAsync(ws + L.xin, x, (size_t)N * L.H * L.W * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return -10;
    FS_TRY(vgg_forward(L, w, b, prepared, ws, s, prep_mask));
    const PrepLayout P = prep_layout(prepared ? prep_mask : 0);
    auto tap_of = [&](int l) -> const float* {
        for (int i = 0; i < n_layers; ++i)
            if (layers[i] == l) return dfeat[i];
        return nullptr;
    };
    float* pre_cur = ws + L.d_pre;
    float* pre_nxt = ws + L.d_in[0];
    FS_TRY(vgg_bwd_route(ws + L.act[L.lmax], nullptr, tap_of(L.lmax), 0, pre_cur, N, L.Hl[L.lmax], L.Wl[L.lmax], kCout[L.lmax], s));
    for (int l = L.lmax; l >= 0; --l) {
        ConvArgs a{};
        a.x = pre_cur;
        a.N = N;
        a.H = a.Ho = L.Hl[l];
        a.W = a.Wo = L.Wl[l];
        a.Cin = kCout[l];
        a.Cout = kCin[l];
        a.KH = a.KW = 3;
        a.stride = 1;
        a.pad_t = a.pad_l = 1;
        float* flipt = nullptr;
        if (prepared) {
            a.w = prepared + P.flipt[l];
            const bool wl = l >= 1 && wino_layer_on(16 + l);
            a.w_wino = (wl && (P.mask & PREP_F2)) ? prepared + P.wino[l][1] : nullptr;
            a.w_wino2 = (wl && (P.mask & PREP_F2)) ? prepared + P.wino2[l][1] : nullptr;
            a.w_wino4 = (wl && (P.mask & PREP_W4) && !vgg_use_4t()) ? prepared + P.wino4[l][1] : nullptr;
            a.w_wino4t = (wl && (P.mask & PREP_W4T) && vgg_use_4t()) ? prepared + P.wino4t[l][1] : nullptr;
            a.w_wino4u = (a.w_wino4t && kCin[l] % 128 == 0) ? prepared + P.wino4u[l][1] : nullptr;
        } else {   // no prepared buffer: flip-transpose this layer's filter into the caller's scratch (9 * 512 * 512 floats)
            flipt = flipt_scratch;
            FS_TRY(wt_flip_transpose(w[l], flipt, 3, 3, kCin[l], kCout[l], s));
            a.w = flipt;
        }
        a.split_ws = ws + L.splitws;
        a.split_ws_floats = L.splitws_floats;
        if (l == 0) {
            a.y = dx;
            if (conv3x3_to3_eligible(a)) return conv3x3_to3_launch(a, s);
            a.p = conv_plan(a);
            return conv_launch(a, s);
        }
        const float* tap = tap_of(l - 1);
        if (!pool_after(l - 1)) {
            a.y = pre_nxt;
            a.add_src = tap;
            a.add_pad = 0;
            a.mask_src = ws + L.act[l - 1];
            a.p = conv_plan(a);
            FS_TRY(conv_launch(a, s));
        } else {
            a.y = ws + L.d_in[1];
            a.p = conv_plan(a);
            FS_TRY(conv_launch(a, s));
            FS_TRY(vgg_bwd_route(ws + L.act[l - 1], ws + L.d_in[1], tap, 1, pre_nxt, N, L.Hl[l - 1], L.Wl[l - 1], kCout[l - 1], s));
        }
        float* t = pre_cur;
        pre_cur = pre_nxt;
        pre_nxt = t;
    }
    return 0;
}

int style_targets(const VggLayout& L, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS],
                  const fs_loss_cfg& cfg, const float* img, float* const grams[4], float* ws, hipStream_t s) {
    if (hipMemcpyAsync(ws + L.xin, img, (size_t)L.H * L.W * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return -10;
    VggLayout L2 = L;
    int lmax = 0;
    for (int i = 0; i < cfg.n_style; ++i)
        if (cfg.style_layer[i] > lmax) lmax = cfg.style_layer[i];
    L2.lmax = lmax;
    FS_TRY(vgg_forward(L2, w, b, nullptr, ws, s));
    for (int i = 0; i < cfg.n_style; ++i) FS_TRY(gram_forward(L2, cfg.style_layer[i], ws + L2.act[cfg.style_layer[i]], grams[i], ws, s));
    return 0;
}

int perceptual_loss(const VggLayout& L, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS],
                    const float* prepared, const fs_loss_cfg& cfg, const float* y, const float* content, float* losses,
                    float* dy, float* ws, hipStream_t s, unsigned prep_mask, const StreamAux* aux) {
    const int N = L.N;
    const size_t img = (size_t)N * L.H * L.W * 3;
    // [y ; content] as one 2N batch at ws + L.xin.  A caller that keeps the two tensors THERE (fs_perceptual_ws_input: the transform net writes y
    // into the workspace, the input batch lives in it) saves both staging copies.
    if (y != ws + L.xin && hipMemcpyAsync(ws + L.xin, y, img * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return -10;
    if (L.cmax >= 0 && content != ws + L.xin + img &&
        hipMemcpyAsync(ws + L.xin + img, content, img * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return -10;
    // (a kernel, not hipMemsetAsync: replayed from a single-stream hipGraph -- the training step of a small batch -- the 16-byte memset NODE left
    // a stale 8-byte value in losses[2..3] from the second replay on (ROCm 7.2; found when the filter-gradient fork became batch-dependent;
    // with the fork in the graph the same node behaved).  No memset on any capturable path of the library any more.)
    // (round 5: the four scalars are WRITTEN once, by loss_finish at the end, from the partial sums every term leaves in ws + L.lossp -- the per-term
    // sum launches, the clear and the total are gone)
    FS_TRY(vgg_forward(L, w, b, prepared, ws, s, prep_mask, aux));
    const PrepLayout P = prep_layout(prep_mask);

    // ---- losses ----
    LossFinish lf{};
    lf.losses = losses;
    size_t lp_off = 0;
    auto term = [&](int slot, int n_partial, float scale) -> float* {   // reserves the term's partial sums; nullptr: no room (never with <= 4 + 4 + 1 terms)
        if (lf.n >= LossFinish::kMax || lp_off + (size_t)n_partial > L.lossp_floats) return nullptr;
        float* p = ws + L.lossp + lp_off;
        lf.job[lf.n].partial = p;
        lf.job[lf.n].n = n_partial;
        lf.job[lf.n].slot = slot;
        lf.job[lf.n].scale = scale;
        ++lf.n;
        lp_off += (size_t)n_partial;
        return p;
    };
    {
        // Gram matrices: the matrix kernels layer by layer, then ONE launch that reduces the slabs of all layers, mirrors them, and leaves
        // S = 4w/(c^2 hwc) (G - Gt) (dF = F S, G symmetric) and the partial sums of (G - Gt)^2 (loss += w * sum / c^2, losses.py:61-64)
        bool batch = cfg.n_style > 0 && tune_int("FS_GRAM_FINISH_BATCH", 1);
        for (int i = 0; i < cfg.n_style; ++i) {
            const int l = cfg.style_layer[i];
            batch = batch && gram2_eligible(N, L.Hl[l] * L.Wl[l], kCout[l]);
        }
        GramFinishJob jobs[4];
        for (int i = 0; i < cfg.n_style; ++i) {
            const int l = cfg.style_layer[i], C = kCout[l], HW = L.Hl[l] * L.Wl[l];
            const float hwc = (float)HW * C;
            const float wgt = cfg.style_weight[i];
            if (batch) {
                FS_TRY(gram2_stream(ws + L.act[l], ws + L.gslab[i], N, HW, C, s));
                GramFinishJob& j = jobs[i];
                j.slabs = ws + L.gslab[i];
                j.Gt = cfg.target_gram[i];
                j.G = tune_int("FS_GRAM_FINISH_KEEP_G", 0) ? ws + L.gram[i] : nullptr;   // (nothing reads G after the finish: S and the loss's partial sums are its products)
                j.S = ws + L.sm[i];
                j.HW = HW;
                j.C = C;
                j.scale = 1.0f / hwc;
                j.gscale = 4.0f * wgt / ((float)C * C * hwc);
                j.partial = term(2, gram2_finish_partials(N, C), wgt / ((float)C * C));
                if (!j.partial) return -12;
            } else {
                FS_TRY(gram_forward(L, l, ws + L.act[l], ws + L.gram[i], ws, s));
                float* pp = term(2, 1024, wgt / ((float)C * C));
                if (!pp) return -12;
                int np = 0;
                FS_TRY(sqdiff_partials(ws + L.gram[i], cfg.target_gram[i], (size_t)C * C, (size_t)N * C * C, 4.0f * wgt / ((float)C * C * hwc), ws + L.sm[i], pp,
                                       &np, s));
                lf.job[lf.n - 1].n = np;
            }
        }
        if (batch) FS_TRY(gram2_finish_batch(jobs, cfg.n_style, N, s));
    }

    // ---- backward ----
    // tap gradient of layer l (content term first, the style term accumulates through add_src)
    // tap gradient of layer l (content term first, the style term accumulates through add_src).
    // fuse_dst != nullptr: the caller wants d_pre[l] = (tap + max-pool gradient of fuse_above routed through act[l]) * (act[l] > 0)
    // written to fuse_dst.  A layer whose only term is ONE style term gets that from the epilogue of the Gram-gradient conv
    // itself (ConvArgs::route_src / mask_src) -- the tap tensor is never written and vgg_bwd_route has nothing left to do;
    // *fused reports it.
    auto compute_tap = [&](int l, const float** out, const float* fuse_above, float* fuse_dst, bool* fused) -> int {
        const int C = kCout[l], H = L.Hl[l], W = L.Wl[l];
        const size_t act_n = (size_t)N * H * W * C;  // the y half
        const float* tap = nullptr;
        if (fused) *fused = false;
        int n_terms = 0;
        for (int i = 0; i < cfg.n_content; ++i) n_terms += cfg.content_layer[i] == l;
        for (int i = 0; i < cfg.n_style; ++i) n_terms += cfg.style_layer[i] == l;
        int n_style_here = 0, n_content_here = 0;
        for (int k = 0; k < cfg.n_style; ++k) n_style_here += cfg.style_layer[k] == l;
        for (int k = 0; k < cfg.n_content; ++k) n_content_here += cfg.content_layer[k] == l;
        const bool route_in_gram = fuse_dst && fuse_above && n_style_here == 1 && gram_bwd2_route_eligible(N, H, W, C);
        // (round 5) a layer that carries one content and one style term and takes the routed Gram-gradient launch below forms the content term THERE
        // (F is that kernel's operand): no sqdiff pass, no content-gradient tensor
        const bool content_in_gram = route_in_gram && n_content_here == 1 && tune_int("FS_GRAM_CONTENT_FUSED", 1) && gram_bwd2_route_grid(N, H, W, C) <= 1024;
        const float* fused_content = nullptr;
        float fused_cscale = 0.f;
        float* fused_cpartial = nullptr;
        for (int i = 0; i < cfg.n_content; ++i)
            if (cfg.content_layer[i] == l) {
                const float hwc = (float)H * W * C;
                const float wgt = cfg.content_weight[i];
                if (content_in_gram) {
                    fused_content = ws + L.act[l] + act_n;
                    fused_cscale = 2.0f * wgt / hwc;
                    fused_cpartial = term(1, gram_bwd2_route_grid(N, H, W, C), wgt / hwc);
                    if (!fused_cpartial) return -12;
                    continue;
                }
                // reference losses.py:32-37: w * sum_{b,h,w,c} (phi(Y)-phi_t)^2 / (h*w*c)
                float* pp = term(1, 1024, wgt / hwc);
                if (!pp) return -12;
                int np = 0;
                FS_TRY(sqdiff_partials(ws + L.act[l], ws + L.act[l] + act_n, act_n, act_n, 2.0f * wgt / hwc, ws + L.d_tap, pp, &np, s));
                lf.job[lf.n - 1].n = np;
                tap = ws + L.d_tap;
            }
        for (int i = 0; i < cfg.n_style; ++i)
            if (cfg.style_layer[i] == l) {
                ConvArgs a{};
                a.x = ws + L.act[l];
                a.w = ws + L.sm[i];
                a.w_nstride = (long long)C * C;
                a.N = N;
                a.H = a.Ho = H;
                a.W = a.Wo = W;
                a.Cin = a.Cout = C;
                a.KH = a.KW = 1;
                a.stride = 1;
                a.add_src = tap;
                a.add_pad = 0;
                float* dst = tap == ws + L.d_tap ? ws + L.d_tap2 : ws + L.d_tap;
                a.y = dst;
                a.p = conv_plan(a);
                // (measured at batch 32: the mask alone -- last layer, no pool behind it -- is free in the conv's epilogue; the pool
                // routing costs the Gram-gradient conv +0.45 ms for 0.75 ms of vgg_bwd_route saved, one 4-byte load per lane
                // against that kernel's 16-byte streams: 0.4 % of the step, so it stays a knob, FS_VGG_ROUTE_FUSED=1)
                if (fuse_dst && n_terms == 1 && (!fuse_above || tune_int("FS_VGG_ROUTE_FUSED", 0))) {
                    ConvArgs f = a;
                    f.mask_src = ws + L.act[l];
                    f.route_src = fuse_above;
                    f.y = fuse_dst;
                    if (!fuse_above || conv_route_ok(f)) {
                        FS_TRY(conv_launch(f, s));
                        *fused = true;
                        *out = nullptr;
                        return 0;
                    }
                }
                // (round 5) ... and the streaming Gram-gradient kernel does the routing and the mask itself where the map tiles into row pairs (the
                // three pooled style layers of a 256 x 256 step): vgg_bwd_route's 0.63 ms per batch-32 step are gone, the tap tensor is never written
                if (route_in_gram) {
                    FS_TRY(gram_bwd2_launch(a.x, a.w, a.add_src, fuse_dst, N, H * W, C, s, fuse_above, W, fused_content, fused_cscale, fused_cpartial));
                    *fused = true;
                    *out = nullptr;
                    return 0;
                }
                if (gram_bwd2_eligible(N, H * W, C))   // streaming kernel with S[n] in registers (fs_gram.hip)
                    FS_TRY(gram_bwd2_launch(a.x, a.w, a.add_src, dst, N, H * W, C, s));
                else
                    FS_TRY(conv_launch(a, s));
                tap = dst;
            }
        *out = tap;
        return 0;
    };
    // d_pre[l] = (gradient reaching act[l]) * (act[l] > 0).  For the last layer it is the tap gradient alone;
    // below, the dgrad conv of layer l either writes d_pre[l-1] directly (no pool in between: tap add and
    // ReLU mask fused into its epilogue) or writes the pooled gradient that vgg_bwd_route scatters.
    float* pre_cur = ws + L.d_pre;
    float* pre_nxt = ws + L.d_in[0];
    {
        const float* tap = nullptr;
        bool fused = false;
        FS_TRY(compute_tap(L.lmax, &tap, nullptr, pre_cur, &fused));
        if (!fused)
            FS_TRY(vgg_bwd_route(ws + L.act[L.lmax], nullptr, tap, 0, pre_cur, N, L.Hl[L.lmax], L.Wl[L.lmax], kCout[L.lmax], s));
    }
    // (runs of split-bf16 input-gradient launches -- conv4_3, conv4_2, conv4_1 at the training shapes -- go out as two half-batch chains, see W6Chains)
    W6Chains bchains{aux, s};
    const bool use_chains = w6_chains_on(aux, N);
    auto launch_bwd = [&](const ConvArgs& a) -> int {
        if (use_chains && w6_chain_launch_ok(a)) return bchains.launch(a);
        FS_TRY(bchains.end());
        return conv_launch(a, s);
    };
    for (int l = L.lmax; l >= 0; --l) {
        const int C = kCout[l], H = L.Hl[l], W = L.Wl[l];
        ConvArgs a{};
        a.x = pre_cur;
        a.w = prepared + P.flipt[l];
        const bool wl = l >= 1 && wino_layer_on(16 + l);
        a.w_wino = (wl && (P.mask & PREP_F2)) ? prepared + P.wino[l][1] : nullptr;
        a.w_wino2 = (wl && (P.mask & PREP_F2)) ? prepared + P.wino2[l][1] : nullptr;
        a.w_wino4 = (wl && (P.mask & PREP_W4) && !vgg_use_4t()) ? prepared + P.wino4[l][1] : nullptr;
        a.w_wino4t = (wl && (P.mask & PREP_W4T) && vgg_use_4t()) ? prepared + P.wino4t[l][1] : nullptr;
        a.w_wino4u = (a.w_wino4t && kCin[l] % 128 == 0) ? prepared + P.wino4u[l][1] : nullptr;
        if (wl && (P.mask & PREP_W6) && vgg_want_w6() && L.w6ws_floats && w6_layer_ok(l, true)) {
            a.w_wino6 = reinterpret_cast<const unsigned short*>(prepared + P.wino6[l][1]);
            a.w6_ws = ws + L.w6ws;
            a.w6_ws_floats = L.w6ws_floats;
            w6_pipe_args(&a, aux);
        }
        a.N = N;
        a.H = a.Ho = H;
        a.W = a.Wo = W;
        a.Cin = C;
        a.Cout = kCin[l];
        a.KH = a.KW = 3;
        a.stride = 1;
        a.pad_t = a.pad_l = 1;
        a.split_ws = ws + L.splitws;
        a.split_ws_floats = L.splitws_floats;
        if (l == 0) {
            FS_TRY(bchains.end());
            a.y = dy;
            if (conv3x3_to3_eligible(a)) {   // 64 -> 3 channels: vector-ALU kernel (fs_c3.hip), no padded MFMA columns
                FS_TRY(conv3x3_to3_launch(a, s));
                break;
            }
            a.p = conv_plan(a);
            FS_TRY(conv_launch(a, s));
            break;
        }
        const float* tap = nullptr;
        if (!pool_after(l - 1)) {
            bool has_tap = false;   // (a tap gradient is computed on the caller's stream from tensors both chains write: join first)
            for (int i = 0; i < cfg.n_content; ++i) has_tap = has_tap || cfg.content_layer[i] == l - 1;
            for (int i = 0; i < cfg.n_style; ++i) has_tap = has_tap || cfg.style_layer[i] == l - 1;
            if (has_tap) FS_TRY(bchains.end());
            FS_TRY(compute_tap(l - 1, &tap, nullptr, nullptr, nullptr));
            a.y = pre_nxt;
            a.add_src = tap;
            a.add_pad = 0;
            a.mask_src = ws + L.act[l - 1];
            a.p = conv_plan(a);
            FS_TRY(launch_bwd(a));
        } else {
            a.y = ws + L.d_in[1];
            a.p = conv_plan(a);
            FS_TRY(launch_bwd(a));
            FS_TRY(bchains.end());
            bool fused = false;
            FS_TRY(compute_tap(l - 1, &tap, ws + L.d_in[1], pre_nxt, &fused));
            if (!fused)
                FS_TRY(vgg_bwd_route(ws + L.act[l - 1], ws + L.d_in[1], tap, 1, pre_nxt, N, L.Hl[l - 1], L.Wl[l - 1], kCout[l - 1], s));
        }
        float* t = pre_cur;
        pre_cur = pre_nxt;
        pre_nxt = t;
    }
    // TV term on y itself (reference losses.py:70-97, train.py:183-184); beta defaults to 0
    if (cfg.beta != 0.0f) {
        float* pp = term(3, 1024, cfg.beta);
        if (!pp) return -12;
        int np = 0;
        FS_TRY(tv_partials(y, N, L.H, L.W, 3, cfg.beta, dy, pp, &np, s));
        lf.job[lf.n - 1].n = np;
    }
    return loss_finish(lf, s);
}

}  // namespace fs
