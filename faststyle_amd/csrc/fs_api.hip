// extern "C" surface of libfaststyle_hip.so (include/faststyle_hip.h).
#include "../../include/faststyle_hip.h"
#include "../../include/faststyle_io.h"

#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>

#include "fs_bf16.h"
#include "fs_tnet.h"
#include "fs_vgg.h"

struct fs_ctx {
    int device;
    hipStream_t stream;
    // cached layouts (recomputed when the shape changes)
    fs::TnetLayout tnet;
    unsigned tnet_epoch = 0;   // fs::tune_epoch() the layout was planned under
    bool tnet_valid;
    // the last fs_tnet_forward (fp32) that rebuilt the re-laid-out filters inside its workspace: FS_FLAG_PARAMS_FROZEN skips the rebuild on a match
    const float* fwd_params = nullptr;
    const void* fwd_ws = nullptr;
    unsigned fwd_serial = 0;     // tnet_serial at that call
    unsigned tnet_serial = 0;    // bumped whenever the cached layout is re-planned
    // the last few workspaces an fp32 fs_tnet_forward filled, and how: fs_tnet_backward refuses a workspace whose forward had another shape / method
    struct FwdRec {
        const void* ws = nullptr;
        int N = 0, H = 0, W = 0, deconv = 0;
        bool bwd_filters = false;   // that forward (FS_FLAG_SAVE_FOR_BWD) also built the backward's input-gradient filters, for these parameters
        const float* params = nullptr;
    } fwd_recs[8];
    int fwd_rec_next = 0;
    fs::BTnetLayout* btnet;  // bf16 inference layout (allocated on first use)
    hipStream_t side;      // second stream for the filter-gradient branch of fs_tnet_backward
    hipEvent_t ev[34];
    bool have_side;
    // the last buffers fs_vgg_prepare filled and the kernel generations each carries (fs_vgg.hip: PrepLayout); a buffer this ctx did not prepare is taken to have
    // been prepared under the knobs of the moment
    struct PrepRec {
        const float* p = nullptr;
        unsigned mask = 0;
    } prep_recs[4];
    int prep_next = 0;
};

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace fs {
namespace {
struct TuneEntry {
    char name[48];
    bool present;
    int val;
};
TuneEntry g_tune[64];
int g_ntune = 0;
std::mutex g_tune_mu;
}  // namespace
int tune_int(const char* name, int unset) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    for (int i = 0; i < g_ntune; ++i)
        if (!strcmp(g_tune[i].name, name)) return g_tune[i].present ? g_tune[i].val : unset;
    const char* v = getenv(name);
    if (g_ntune < 64) {
        TuneEntry& e = g_tune[g_ntune++];
        strncpy(e.name, name, sizeof(e.name) - 1);
        e.name[sizeof(e.name) - 1] = 0;
        e.present = v != nullptr;
        e.val = v ? (int)strtol(v, nullptr, 0) : 0;
    }
    return v ? (int)strtol(v, nullptr, 0) : unset;
}
static std::atomic<unsigned> g_tune_epoch{1};
void tune_reload() {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_ntune = 0;
    g_tune_epoch.fetch_add(1);
}
unsigned tune_epoch() { return g_tune_epoch.load(); }
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace fs

extern "C" {

int fs_resize_bicubic_u8(fs_ctx* ctx, const unsigned char* src, int H, int W, float* dst, int Ho, int Wo) {
    if (!ctx || !src || !dst) return fail(-1, "fs_resize_bicubic_u8: null argument");
    if (H < 1 || W < 1 || Ho < 1 || Wo < 1) return fail(-1, "fs_resize_bicubic_u8: bad shape %dx%d -> %dx%d", H, W, Ho, Wo);
    const int rc = fs::resize_bicubic_u8(src, H, W, dst, Ho, Wo, ctx->stream);
    return rc ? fail(rc, "fs_resize_bicubic_u8: launch failed (%d)", rc) : 0;
}
int fs_resize_bicubic_u8x(fs_ctx* ctx, const unsigned char* src, int H, int W, int pixel_bytes, float* dst, int Ho, int Wo) {
    if (!ctx || !src || !dst) return fail(-1, "fs_resize_bicubic_u8x: null argument");
    if (H < 1 || W < 1 || Ho < 1 || Wo < 1) return fail(-1, "fs_resize_bicubic_u8x: bad shape %dx%d -> %dx%d", H, W, Ho, Wo);
    if (pixel_bytes != 3 && pixel_bytes != 4) return fail(-2, "fs_resize_bicubic_u8x: pixel_bytes must be 3 (RGB) or 4 (RGBX), got %d", pixel_bytes);
    const int rc = fs::resize_bicubic_u8(src, H, W, dst, Ho, Wo, ctx->stream, pixel_bytes);
    return rc ? fail(rc, "fs_resize_bicubic_u8x: launch failed (%d)", rc) : 0;
}

int fs_u8_to_f32(fs_ctx* ctx, const unsigned char* src, size_t n, float* dst) {
    if (!ctx || !src || !dst) return fail(-1, "fs_u8_to_f32: null argument");
    if (((uintptr_t)src & 3) || ((uintptr_t)dst & 15)) return fail(-1, "fs_u8_to_f32: src must be 4-byte, dst 16-byte aligned");
    return fs::u8_to_f32(src, dst, n, ctx->stream) ? fail(-3, "fs_u8_to_f32: launch failed") : 0;
}
int fs_f32_to_u8(fs_ctx* ctx, const float* src, size_t npix, int swap_rb, unsigned char* dst) {
    if (!ctx || !src || !dst) return fail(-1, "fs_f32_to_u8: null argument");
    return fs::f32_to_u8(src, dst, npix, swap_rb, ctx->stream) ? fail(-3, "fs_f32_to_u8: launch failed") : 0;
}

// tests / tuning scripts: re-read the FS_* knobs after changing the environment (not part of the product ABI)
void fs_debug_reload_env(void) { fs::tune_reload(); }
const char* fs_last_error(void) { return g_err; }
const char* fs_version(void) { return "faststyle_hip 0.1 (gfx950, fp32 + split-bf16 MFMA)"; }

int fs_ctx_create(int device, void* hip_stream, fs_ctx** out) {
    if (!out) return fail(-1, "fs_ctx_create: out is null");
    if (hipSetDevice(device) != hipSuccess) return fail(-2, "fs_ctx_create: hipSetDevice(%d) failed", device);
    fs_ctx* c = new fs_ctx();
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    c->tnet_valid = false;
    c->btnet = nullptr;
    c->have_side = false;
    if (!fs::tune_int("FS_NO_SIDE_STREAM", 0) && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess) {
        c->have_side = true;
        for (int i = 0; i < 34; ++i)
            if (hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming) != hipSuccess) c->have_side = false;
    }
    *out = c;
    return 0;
}
void fs_ctx_destroy(fs_ctx* ctx) {
    if (ctx && ctx->have_side) {
        for (int i = 0; i < 34; ++i) (void)hipEventDestroy(ctx->ev[i]);
        (void)hipStreamDestroy(ctx->side);
    }
    if (ctx) delete ctx->btnet;
    delete ctx;
}
int fs_ctx_set_stream(fs_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(-1, "null ctx");
    ctx->stream = (hipStream_t)hip_stream;
    return 0;
}

// ------------------------------------------------------------------------------ profiling hook
static fs::Profiler g_prof;
int fs_profile_begin(fs_ctx* ctx) {
    if (!ctx) return fail(-1, "null ctx");
    g_prof.reset();
    fs::Profiler::current() = &g_prof;
    return 0;
}
const char* fs_profile_family_name(int family) { return fs::prof_family_name(family); }
int fs_profile_end(fs_ctx* ctx, double out[3 * FS_PROFILE_FAMILIES]) {
    if (!ctx || !out) return fail(-1, "fs_profile_end: null argument");
    fs::Profiler::current() = nullptr;
    static_assert(fs::Profiler::kFamilies == FS_PROFILE_FAMILIES, "header and profiler disagree");
    double tmp[fs::Profiler::kFamilies][3];
    if (g_prof.collect(tmp)) return fail(-4, "fs_profile_end: event query failed");
    memcpy(out, tmp, sizeof(tmp));
    g_prof.reset();
    return 0;
}

// ------------------------------------------------------------------------------ transform net
int fs_tnet_param_info(int idx, const char** name, int* offset, int* ndim, int dims[4]) {
    if (idx < 0 || idx >= 48) return fail(-1, "fs_tnet_param_info: index %d out of range", idx);
    const fs::ParamInfo& p = fs::param_table()[idx];
    if (name) *name = p.name;
    if (offset) *offset = p.offset;
    if (ndim) *ndim = p.ndim;
    if (dims) memcpy(dims, p.dims, sizeof(int) * 4);
    return 0;
}

int fs_tnet_out_shape(int H, int W, int* Ho, int* Wo) {
    if (H < 41 || W < 41) return fail(-1, "fs_tnet_out_shape: REFLECT padding by 40 needs H,W >= 41 (got %dx%d)", H, W);
    auto f = [](int s) { return 4 * (fs::cdiv(fs::cdiv(s + 80, 2), 2) - 20); };
    if (Ho) *Ho = f(H);
    if (Wo) *Wo = f(W);
    return 0;
}

static const fs::TnetLayout* get_layout(fs_ctx* ctx, int N, int H, int W, int flags) {
    const int deconv = (flags & FS_FLAG_UPSAMPLE_DECONV) ? 1 : 0;
    // (a layout records plan decisions that depend on the tuning knobs: re-planned after fs_debug_reload_env)
    if (!ctx->tnet_valid || ctx->tnet.N != N || ctx->tnet.H != H || ctx->tnet.W != W || ctx->tnet.deconv != deconv ||
        ctx->tnet.wino_mode != fs::tnet_wino_mode() || ctx->tnet_epoch != fs::tune_epoch()) {
        fs::tnet_layout(N, H, W, deconv, &ctx->tnet);
        ctx->tnet_valid = true;
        ctx->tnet_epoch = fs::tune_epoch();
        ++ctx->tnet_serial;
        for (auto& q : ctx->fwd_recs) q.bwd_filters = false;   // (filters a forward built under the OLD plan are not what a backward under the new one reads)
    }
    return &ctx->tnet;
}

size_t fs_tnet_workspace_bytes(int N, int H, int W, int flags) {
    if (N < 1 || H < 41 || W < 41) return 0;
    if (flags & FS_FLAG_BF16) {
        if (flags & (FS_FLAG_SAVE_FOR_BWD | FS_FLAG_UPSAMPLE_DECONV)) return 0;  // inference, resize-conv models only
        fs::BTnetLayout* B = new fs::BTnetLayout();
        fs::tnet_layout_bf16(N, H, W, B);
        const size_t bytes = B->total_bytes;
        delete B;
        return bytes;
    }
    fs::TnetLayout L;
    fs::tnet_layout(N, H, W, (flags & FS_FLAG_UPSAMPLE_DECONV) ? 1 : 0, &L);
    return L.total_floats * sizeof(float);
}

int fs_tnet_forward(fs_ctx* ctx, const float* params, const float* x, int N, int H, int W, float* y, void* ws,
                    size_t ws_bytes, int flags) {
    if (!ctx || !params || !x || !y || !ws) return fail(-1, "fs_tnet_forward: null argument");
    if (N < 1 || H < 41 || W < 41) return fail(-2, "fs_tnet_forward: need N>=1 and H,W>=41 (got %d,%d,%d)", N, H, W);
    if (flags & FS_FLAG_BF16) {
        if (flags & (FS_FLAG_SAVE_FOR_BWD | FS_FLAG_UPSAMPLE_DECONV))
            return fail(-2, "fs_tnet_forward: FS_FLAG_BF16 is inference-only and covers the resize-conv models");
        if (!ctx->btnet) ctx->btnet = new fs::BTnetLayout();
        fs::BTnetLayout* B = ctx->btnet;
        if (B->geo.N != N || B->geo.H != H || B->geo.W != W) fs::tnet_layout_bf16(N, H, W, B);
        if (ws_bytes < B->total_bytes)
            return fail(-3, "fs_tnet_forward: workspace too small (%zu < %zu bytes)", ws_bytes, B->total_bytes);
        const int rc = fs::tnet_forward_bf16(*B, params, x, y, ws, ctx->stream);
        return rc ? fail(rc, "fs_tnet_forward(bf16): launch failed (%d)", rc) : 0;
    }
    const fs::TnetLayout* L = get_layout(ctx, N, H, W, flags);
    if (ws_bytes < L->total_floats * sizeof(float))
        return fail(-3, "fs_tnet_forward: workspace too small (%zu < %zu bytes)", ws_bytes, L->total_floats * sizeof(float));
    const bool reuse = (flags & FS_FLAG_PARAMS_FROZEN) && ctx->fwd_params == params && ctx->fwd_ws == ws && ctx->fwd_serial == ctx->tnet_serial;
    const bool with_bwd = (flags & FS_FLAG_SAVE_FOR_BWD) && !reuse && fs::tune_int("FS_TNET_BWD_FILTERS_IN_FWD", 1);
    const int rc = fs::tnet_forward(*L, params, x, y, (float*)ws, ctx->stream, reuse, with_bwd);
    if (rc) {
        ctx->fwd_params = nullptr;
        return fail(rc, "fs_tnet_forward: launch failed (%d)", rc);
    }
    ctx->fwd_params = params;
    ctx->fwd_ws = ws;
    ctx->fwd_serial = ctx->tnet_serial;
    {   // remember how this workspace was filled (fs_tnet_backward checks it)
        fs_ctx::FwdRec* r = nullptr;
        for (auto& q : ctx->fwd_recs)
            if (q.ws == ws) r = &q;
        if (!r) r = &ctx->fwd_recs[ctx->fwd_rec_next++ & 7];
        r->ws = ws;
        r->N = N;
        r->H = H;
        r->W = W;
        r->deconv = (flags & FS_FLAG_UPSAMPLE_DECONV) ? 1 : 0;
        r->bwd_filters = with_bwd;
        r->params = params;
    }
    return 0;
}

int fs_tnet_invalidate(fs_ctx* ctx) {
    if (!ctx) return fail(-1, "fs_tnet_invalidate: null ctx");
    ctx->fwd_params = nullptr;   // the next FS_FLAG_PARAMS_FROZEN call rebuilds the re-laid-out filters whatever its pointers
    ctx->fwd_ws = nullptr;
    for (auto& q : ctx->fwd_recs) q = fs_ctx::FwdRec();
    return 0;
}

int fs_tnet_backward(fs_ctx* ctx, const float* params, const float* x, const float* dy, int N, int H, int W, float* grads,
                     void* ws, size_t ws_bytes, int flags) {
    if (!ctx || !params || !x || !dy || !grads || !ws) return fail(-1, "fs_tnet_backward: null argument");
    if (N < 1 || H < 41 || W < 41) return fail(-2, "fs_tnet_backward: need N>=1 and H,W>=41");
    const fs::TnetLayout* L = get_layout(ctx, N, H, W, flags);
    if (ws_bytes < L->total_floats * sizeof(float))
        return fail(-3, "fs_tnet_backward: workspace too small (%zu < %zu bytes)", ws_bytes, L->total_floats * sizeof(float));
    for (const auto& q : ctx->fwd_recs)   // a workspace this context filled with ANOTHER shape / upsample method holds nothing this backward can read
        if (q.ws == ws && (q.N != N || q.H != H || q.W != W || q.deconv != ((flags & FS_FLAG_UPSAMPLE_DECONV) ? 1 : 0)))
            return fail(-5, "fs_tnet_backward: the workspace was filled by fs_tnet_forward(N=%d, %dx%d, %s) -- this call is (N=%d, %dx%d, %s)", q.N, q.H,
                        q.W, q.deconv ? "deconv" : "resize", N, H, W, (flags & FS_FLAG_UPSAMPLE_DECONV) ? "deconv" : "resize");
    bool filters_ready = false;   // the forward that filled this workspace built the input-gradient filters too (same parameters: same step)
    for (auto& q : ctx->fwd_recs)
        if (q.ws == ws && q.bwd_filters && q.params == params) {
            filters_ready = true;
            q.bwd_filters = false;   // (once: a second backward on the same workspace rebuilds them -- the caller may have updated `params` in place)
        }
    fs::StreamAux aux{ctx->side, ctx->ev, 34};
    const int rc = fs::tnet_backward(*L, params, x, dy, grads, (float*)ws, ctx->stream, ctx->have_side ? &aux : nullptr, filters_ready);
    return rc ? fail(rc, "fs_tnet_backward: launch failed (%d)", rc) : 0;
}

// inspection: where fs_tnet_forward(FS_FLAG_SAVE_FOR_BWD) leaves its saved tensors inside the caller's workspace
int fs_tnet_ws_tensor(int N, int H, int W, int flags, int unit, int what, size_t* offset_floats, int dims[4]) {
    if (N < 1 || H < 41 || W < 41 || (flags & FS_FLAG_BF16)) return fail(-2, "fs_tnet_ws_tensor: fp32 layouts only, H,W >= 41");
    if (!offset_floats || !dims) return fail(-1, "fs_tnet_ws_tensor: null argument");
    fs::TnetLayout* L = new fs::TnetLayout();
    fs::tnet_layout(N, H, W, (flags & FS_FLAG_UPSAMPLE_DECONV) ? 1 : 0, L);
    int rc = 0;
    if (what == FS_TNET_WS_H) {
        if (unit < 0 || unit >= 5) {
            rc = fail(-2, "fs_tnet_ws_tensor: residual block %d out of range", unit);
        } else {
            const fs::Unit& u = L->u[3 + 2 * unit + 1];
            *offset_floats = L->h[unit];
            dims[0] = N, dims[1] = u.Hout, dims[2] = u.Wout, dims[3] = 64;
        }
    } else if (unit < 0 || unit >= 16 || what < 0 || what > FS_TNET_WS_RSTD) {
        rc = fail(-2, "fs_tnet_ws_tensor: unit %d / tensor %d out of range", unit, what);
    } else {
        const fs::Unit& u = L->u[unit];
        if (what == FS_TNET_WS_Z) {
            *offset_floats = u.z;
            dims[0] = N, dims[1] = u.Hout, dims[2] = u.Wout, dims[3] = u.Cout;
        } else {
            *offset_floats = what == FS_TNET_WS_A ? u.a : (what == FS_TNET_WS_B ? u.b : (what == FS_TNET_WS_MEAN ? u.mean : u.rstd));
            dims[0] = N, dims[1] = u.Cout, dims[2] = dims[3] = 1;
        }
    }
    delete L;
    return rc;
}

// ------------------------------------------------------------------------------ VGG / losses
size_t fs_vgg_prepared_floats(void) { return fs::vgg_prepared_floats(); }

int fs_vgg_prepare(fs_ctx* ctx, const float* const w[FS_VGG_NLAYERS], float* prepared) {
    if (!ctx || !w || !prepared) return fail(-1, "fs_vgg_prepare: null argument");
    const int rc = fs::vgg_prepare(w, prepared, ctx->stream);
    if (rc < 0) return fail(rc, "fs_vgg_prepare failed (%d)", rc);
    int slot = -1;
    for (int i = 0; i < 4; ++i)
        if (ctx->prep_recs[i].p == prepared) slot = i;
    if (slot < 0) slot = ctx->prep_next++ & 3;
    ctx->prep_recs[slot].p = prepared;
    ctx->prep_recs[slot].mask = (unsigned)rc;
    return 0;
}

static unsigned prep_mask_of(const fs_ctx* ctx, const float* prepared) {
    for (int i = 0; i < 4; ++i)
        if (ctx->prep_recs[i].p == prepared) return ctx->prep_recs[i].mask;
    return fs::vgg_prep_mask();
}

static int check_cfg(const fs_loss_cfg* cfg) {
    if (!cfg) return fail(-1, "null loss cfg");
    if (cfg->n_content < 0 || cfg->n_content > 4 || cfg->n_style < 0 || cfg->n_style > 4)
        return fail(-2, "loss cfg: at most 4 content and 4 style layers");
    for (int i = 0; i < cfg->n_content; ++i)
        if (cfg->content_layer[i] < 0 || cfg->content_layer[i] >= FS_VGG_NLAYERS) return fail(-2, "bad content layer");
    for (int i = 0; i < cfg->n_style; ++i)
        if (cfg->style_layer[i] < 0 || cfg->style_layer[i] >= FS_VGG_NLAYERS) return fail(-2, "bad style layer");
    return 0;
}

size_t fs_perceptual_workspace_bytes(int N, int H, int W, const fs_loss_cfg* cfg) {
    if (check_cfg(cfg) || N < 1 || H < 1 || W < 1) return 0;
    fs::VggLayout L;
    fs::vgg_layout(N, H, W, *cfg, true, &L);
    return L.total_floats * sizeof(float);
}

int fs_perceptual_loss(fs_ctx* ctx, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS],
                       const float* prepared, const fs_loss_cfg* cfg, const float* y, const float* content, int N, int H,
                       int W, float* losses, float* dy, void* ws, size_t ws_bytes) {
    if (!ctx || !w || !b || !prepared || !y || !content || !losses || !dy || !ws)
        return fail(-1, "fs_perceptual_loss: null argument");
    if (int rc = check_cfg(cfg)) return rc;
    for (int i = 0; i < cfg->n_style; ++i)
        if (!cfg->target_gram[i]) return fail(-2, "fs_perceptual_loss: target_gram[%d] is null", i);
    fs::VggLayout L;
    fs::vgg_layout(N, H, W, *cfg, true, &L);
    if (ws_bytes < L.total_floats * sizeof(float)) return fail(-3, "fs_perceptual_loss: workspace too small");
    fs::StreamAux aux{ctx->side, ctx->ev, 34};
    const int rc = fs::perceptual_loss(L, w, b, prepared, *cfg, y, content, losses, dy, (float*)ws, ctx->stream, prep_mask_of(ctx, prepared), ctx->have_side ? &aux : nullptr);
    return rc ? fail(rc, "fs_perceptual_loss: launch failed (%d)", rc) : 0;
}

// inspection: where fs_perceptual_loss leaves the post-ReLU activations of a VGG layer inside the caller's workspace
int fs_perceptual_ws_tensor(int N, int H, int W, const fs_loss_cfg* cfg, int layer, size_t* offset_floats, int dims[4]) {
    if (int rc = check_cfg(cfg)) return rc;
    if (N < 1 || H < 1 || W < 1 || !offset_floats || !dims) return fail(-1, "fs_perceptual_ws_tensor: bad argument");
    fs::VggLayout L;
    fs::vgg_layout(N, H, W, *cfg, true, &L);
    if (layer < 0 || layer > L.lmax) return fail(-2, "fs_perceptual_ws_tensor: layer %d is not evaluated (last: %d)", layer, L.lmax);
    static const int cout[FS_VGG_NLAYERS] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512};
    *offset_floats = L.act[layer];
    dims[0] = layer <= L.cmax ? L.NB : L.N;
    dims[1] = L.Hl[layer];
    dims[2] = L.Wl[layer];
    dims[3] = cout[layer];
    return 0;
}

// where fs_perceptual_loss stages its two inputs inside the caller's workspace (a caller that produces them there skips the copies)
int fs_perceptual_ws_input(int N, int H, int W, const fs_loss_cfg* cfg, size_t* y_offset_floats, size_t* content_offset_floats) {
    if (int rc = check_cfg(cfg)) return rc;
    if (N < 1 || H < 1 || W < 1 || !y_offset_floats || !content_offset_floats) return fail(-1, "fs_perceptual_ws_input: bad argument");
    fs::VggLayout L;
    fs::vgg_layout(N, H, W, *cfg, true, &L);
    *y_offset_floats = L.xin;
    *content_offset_floats = L.cmax >= 0 ? L.xin + (size_t)N * H * W * 3 : (size_t)-1;   // (size_t)-1: no content layer, `content` is not read
    return 0;
}

size_t fs_style_targets_workspace_bytes(int H, int W) {
    fs_loss_cfg cfg{};
    cfg.n_style = 4;
    cfg.style_layer[0] = 1;
    cfg.style_layer[1] = 3;
    cfg.style_layer[2] = 6;
    cfg.style_layer[3] = 9;
    fs::VggLayout L;
    fs::vgg_layout(1, H, W, cfg, false, &L);
    return L.total_floats * sizeof(float);
}

int fs_style_targets(fs_ctx* ctx, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS],
                     const fs_loss_cfg* cfg, const float* style_img, int H, int W, float* const grams[4], void* ws,
                     size_t ws_bytes) {
    if (!ctx || !w || !b || !style_img || !grams || !ws) return fail(-1, "fs_style_targets: null argument");
    if (int rc = check_cfg(cfg)) return rc;
    if (ws_bytes < fs_style_targets_workspace_bytes(H, W)) return fail(-3, "fs_style_targets: workspace too small");
    fs::VggLayout L;
    fs_loss_cfg c2 = *cfg;
    c2.n_content = 0;
    // worst-case layout (all layers) so the size matches fs_style_targets_workspace_bytes
    fs_loss_cfg full{};
    full.n_style = 4;
    full.style_layer[0] = 1;
    full.style_layer[1] = 3;
    full.style_layer[2] = 6;
    full.style_layer[3] = 9;
    fs::vgg_layout(1, H, W, full, false, &L);
    const int rc = fs::style_targets(L, w, b, c2, style_img, grams, (float*)ws, ctx->stream);
    return rc ? fail(rc, "fs_style_targets: launch failed (%d)", rc) : 0;
}

static void features_layout(int N, int H, int W, int max_layer, fs::VggLayout* L) {
    fs_loss_cfg cfg{};
    cfg.n_style = 1;
    cfg.style_layer[0] = max_layer;
    fs::vgg_layout(N, H, W, cfg, false, L);
}
size_t fs_vgg_features_workspace_bytes(int N, int H, int W, int max_layer) {
    if (N < 1 || H < 1 || W < 1 || max_layer < 0 || max_layer >= FS_VGG_NLAYERS) return 0;
    fs::VggLayout L;
    features_layout(N, H, W, max_layer, &L);
    return L.total_floats * sizeof(float);
}
int fs_vgg_features(fs_ctx* ctx, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS], const float* x,
                    int N, int H, int W, int n_layers, const int* layers, float* const* out, void* ws, size_t ws_bytes) {
    if (!ctx || !w || !b || !x || !layers || !out || !ws) return fail(-1, "fs_vgg_features: null argument");
    if (N < 1 || n_layers < 1 || n_layers > FS_VGG_NLAYERS) return fail(-2, "fs_vgg_features: bad N / layer count");
    int lmax = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (layers[i] < 0 || layers[i] >= FS_VGG_NLAYERS) return fail(-2, "fs_vgg_features: layer %d out of range", layers[i]);
        if (!out[i]) return fail(-1, "fs_vgg_features: null output %d", i);
        if (layers[i] > lmax) lmax = layers[i];
    }
    fs::VggLayout L;
    features_layout(N, H, W, lmax, &L);
    if (ws_bytes < L.total_floats * sizeof(float)) return fail(-3, "fs_vgg_features: workspace too small");
    const int rc = fs::vgg_features(L, w, b, x, n_layers, layers, out, (float*)ws, ctx->stream);
    return rc ? fail(rc, "fs_vgg_features: launch failed (%d)", rc) : 0;
}

int fs_loss_sqdiff(fs_ctx* ctx, const float* x, const float* t, size_t t_period, size_t n, float scale, float* out, void* scratch) {
    if (!ctx || !x || !t || !out || !scratch || !t_period) return fail(-1, "fs_loss_sqdiff: null argument");
    if (n % t_period || n / t_period > 1024) return fail(-2, "fs_loss_sqdiff: n = %zu must be 1 .. 1024 whole periods of %zu", n, t_period);
    const int rc = fs::sqdiff_loss(x, t, t_period, n, scale, 0.f, nullptr, out, 0, (float*)scratch, ctx->stream);
    return rc ? fail(rc, "fs_loss_sqdiff: launch failed (%d)", rc) : 0;
}
int fs_loss_tv(fs_ctx* ctx, const float* x, int N, int H, int W, int C, float* out, void* scratch) {
    if (!ctx || !x || !out || !scratch) return fail(-1, "fs_loss_tv: null argument");
    return fs::tv_loss(x, N, H, W, C, 1.0f, 0.f, nullptr, out, (float*)scratch, ctx->stream);
}

// ---- value + gradient forms of the loss terms, and the adjoint of fs_vgg_features (round 6: the pieces a script differentiates through when it composes
// its own objective as train.py:171-204 / slow_style.py:140-176 do; faststyle_amd/autograd.py wraps them as torch.autograd.Functions) -----------------------
int fs_loss_sqdiff_grad(fs_ctx* ctx, const float* x, const float* t, size_t t_period, size_t n, float scale, float* out, float* grad, void* scratch) {
    if (!ctx || !x || !t || !out || !grad || !scratch || !t_period) return fail(-1, "fs_loss_sqdiff_grad: null argument");
    if (n % t_period || n / t_period > 1024) return fail(-2, "fs_loss_sqdiff_grad: n = %zu must be 1 .. 1024 whole periods of %zu", n, t_period);
    const int rc = fs::sqdiff_loss(x, t, t_period, n, scale, 2.0f * scale, grad, out, 0, (float*)scratch, ctx->stream);
    return rc ? fail(rc, "fs_loss_sqdiff_grad: launch failed (%d)", rc) : 0;
}
int fs_loss_tv_grad(fs_ctx* ctx, const float* x, int N, int H, int W, int C, float scale, float* out, float* grad, int accumulate, void* scratch) {
    if (!ctx || !x || !out || !grad || !scratch) return fail(-1, "fs_loss_tv_grad: null argument");
    if (N < 1 || H < 1 || W < 1 || C < 1) return fail(-2, "fs_loss_tv_grad: empty tensor");
    if (!accumulate)
        if (int rc = fs::zero_fill(grad, (size_t)N * H * W * C, ctx->stream)) return fail(rc, "fs_loss_tv_grad: launch failed (%d)", rc);
    const int rc = fs::tv_loss(x, N, H, W, C, scale, scale, grad, out, (float*)scratch, ctx->stream);
    return rc ? fail(rc, "fs_loss_tv_grad: launch failed (%d)", rc) : 0;
}

static const size_t kFliptScratch = (size_t)9 * 512 * 512;   // the largest flip-transposed VGG16 filter (floats)
size_t fs_vgg_dgrad_workspace_bytes(int N, int H, int W, int max_layer) {
    if (N < 1 || H < 1 || W < 1 || max_layer < 0 || max_layer >= FS_VGG_NLAYERS) return 0;
    fs::VggLayout L;
    features_layout(N, H, W, max_layer, &L);
    return (L.total_floats + kFliptScratch) * sizeof(float);
}
int fs_vgg_dgrad(fs_ctx* ctx, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS], const float* prepared, const float* x, int N,
                 int H, int W, int n_layers, const int* layers, const float* const* dfeat, float* dx, void* ws, size_t ws_bytes) {
    if (!ctx || !w || !b || !x || !layers || !dfeat || !dx || !ws) return fail(-1, "fs_vgg_dgrad: null argument");
    if (N < 1 || n_layers < 1 || n_layers > FS_VGG_NLAYERS) return fail(-2, "fs_vgg_dgrad: bad N / layer count");
    int lmax = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (layers[i] < 0 || layers[i] >= FS_VGG_NLAYERS) return fail(-2, "fs_vgg_dgrad: layer %d out of range", layers[i]);
        if (!dfeat[i]) return fail(-1, "fs_vgg_dgrad: null gradient %d", i);
        for (int j = 0; j < i; ++j)
            if (layers[j] == layers[i]) return fail(-2, "fs_vgg_dgrad: layer %d listed twice (sum its gradients first)", layers[i]);
        if (layers[i] > lmax) lmax = layers[i];
    }
    fs::VggLayout L;
    features_layout(N, H, W, lmax, &L);
    if (ws_bytes < (L.total_floats + kFliptScratch) * sizeof(float)) return fail(-3, "fs_vgg_dgrad: workspace too small");
    const int rc = fs::vgg_dgrad(L, w, b, prepared, prepared ? prep_mask_of(ctx, prepared) : 0, x, n_layers, layers, dfeat, dx, (float*)ws,
                                 (float*)ws + L.total_floats, ctx->stream);
    return rc ? fail(rc, "fs_vgg_dgrad: launch failed (%d)", rc) : 0;
}

int fs_adam_tf_step(fs_ctx* ctx, float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                    float beta2, float eps, long long t) {
    if (!ctx || !p || !g || !m || !v) return fail(-1, "fs_adam_tf_step: null argument");
    if (t < 1) return fail(-2, "fs_adam_tf_step: t is the 1-based step count");
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)t)) / (1.0 - std::pow((double)beta1, (double)t));
    return fs::adam_tf(p, g, m, v, n, (float)lr_t, beta1, beta2, eps, ctx->stream);
}

// ------------------------------------------------------------------------------ single ops
static void resolve_pads(int H, int W, int KH, int KW, int stride, int mode, int refl, int src_mode, int* Ho, int* Wo,
                         int* pt, int* pl) {
    int vh = H, vw = W;
    if (src_mode == FS_SRC_REFLECT) {
        vh += 2 * refl;
        vw += 2 * refl;
    }
    if (mode == FS_PAD_SAME) {
        *Ho = fs::cdiv(vh, stride);
        *Wo = fs::cdiv(vw, stride);
        int th = (*Ho - 1) * stride + KH - vh, tw = (*Wo - 1) * stride + KW - vw;
        *pt = th > 0 ? th / 2 : 0;
        *pl = tw > 0 ? tw / 2 : 0;
    } else if (mode == FS_PAD_VALID) {
        *Ho = (vh - KH) / stride + 1;
        *Wo = (vw - KW) / stride + 1;
        *pt = *pl = 0;
    }
}

static int fill_conv(fs_conv_desc* d, fs::ConvArgs* a) {
    if (!d) return fail(-1, "null conv desc");
    if (d->pad_mode != FS_PAD_EXPLICIT)
        resolve_pads(d->H, d->W, d->KH, d->KW, d->stride, d->pad_mode, d->refl, d->src_mode, &d->Ho, &d->Wo, &d->pad_t,
                     &d->pad_l);
    if (d->Cin != 3 && (d->Cin % 4)) return fail(-2, "fs_conv2d: Cin must be 3 or a multiple of 4 (got %d)", d->Cin);
    if (d->shuffle && (d->Cout % 4)) return fail(-2, "fs_conv2d: pixel-shuffle needs Cout %% 4 == 0");
    if (d->Ho < 1 || d->Wo < 1) return fail(-2, "fs_conv2d: empty output");
    *a = fs::ConvArgs{};
    a->x = d->x;
    a->w = d->w;
    a->y = d->y;
    a->N = d->N;
    a->H = d->H;
    a->W = d->W;
    a->Cin = d->Cin;
    a->Ho = d->Ho;
    a->Wo = d->Wo;
    a->Cout = d->Cout;
    a->KH = d->KH;
    a->KW = d->KW;
    a->stride = d->stride;
    a->pad_t = d->pad_t;
    a->pad_l = d->pad_l;
    a->src_mode = d->src_mode;
    a->refl = d->refl;
    a->in_a = d->in_a;
    a->in_b = d->in_b;
    a->in_nstride = d->in_per_sample ? d->Cin : 0;
    a->in_relu = d->in_relu;
    a->bias = d->bias;
    a->out_relu = d->out_relu;
    a->shuffle = d->shuffle;
    a->stats = d->stats;
    a->add_src = d->add_src;
    a->add_pad = d->add_pad;
    a->w_nstride = d->w_nstride;
    a->mask_src = d->mask_src;
    a->route_src = d->route_src;
    if (a->route_src && (!a->mask_src || a->add_pad)) return fail(-2, "fs_conv_desc: route_src needs mask_src and add_pad = 0");
    if (d->w_wino6 && d->w6_ws) {
        a->w_wino6 = static_cast<const unsigned short*>(d->w_wino6);
        a->w6_ws = static_cast<float*>(d->w6_ws);
        a->w6_ws_floats = d->w6_ws_bytes / sizeof(float);
        a->pool_out = d->pool_out;   // (part of the epilogue form the eligibility test looks at; validated below)
        if (!fs::wino6_eligible(*a)) a->w_wino6 = nullptr;
        a->pool_out = nullptr;
    }
    a->w_wino4 = d->w_wino4;
    if (a->w_wino4 && !fs::wino4_eligible(*a)) a->w_wino4 = nullptr;   // (not a 3x3 stride-1 SAME conv of the supported shapes)
    a->w_wino4t = d->w_wino4t;
    a->inb_z = d->inb_z;
    a->inb_mean = d->inb_mean;
    a->inb_rstd = d->inb_rstd;
    a->inb_a = d->inb_a;
    a->inb_b = d->inb_b;
    a->inb_relu = d->inb_relu;
    a->inb_rec = d->inb_rec;
    a->tnet_plan = d->inb_rec ? 1 : 0;   // (the partial sums exist for 16-tile items: plan them whatever the grid)
    if (a->w_wino4t && (a->w_wino4 || !fs::wino4t_eligible(*a))) a->w_wino4t = nullptr;
    if (fs::wino_gen().f2_second()) {   // the filter layout fs_wino_transform_filter produced (see there)
        a->w_wino2 = d->w_wino;
        if (a->w_wino2 && !fs::wino2_eligible(*a)) a->w_wino2 = nullptr;   // (not a 3x3 stride-1 conv of the supported shapes: direct kernel)
    } else {
        a->w_wino = d->w_wino;
        if (a->w_wino && !fs::wino_eligible(*a)) a->w_wino = nullptr;
    }
    a->p = fs::conv_plan(*a);
    if (d->inb_rec && !(a->w_wino4t && a->p.variant == 11 && a->p.TW == 16 && a->p.ksplit <= 1))
        return fail(-2, "fs_conv2d: inb_rec needs an eligible w_wino4t conv (3x3 stride 1, raw or add_src epilogue, inb_z / inb_mean / inb_rstd set)");
    if (d->pool_out) {   // only the Winograd epilogues hold whole pooling windows
        if (!(a->p.variant == 5 || a->p.variant == 6 || a->p.variant == 10 || a->p.variant == 11 || a->p.variant == 12) || a->p.ksplit > 1 || (a->Ho & 1) || (a->Wo & 1))
            return fail(-2, "fs_conv2d: pool_out needs a Winograd-eligible conv with even Ho, Wo");
        a->pool_out = d->pool_out;
    }
    return 0;
}

int fs_conv2d_plan(fs_conv_desc* d, int* tiles_per_image) {
    fs::ConvArgs a;
    if (int rc = fill_conv(d, &a)) return rc;
    if (tiles_per_image) *tiles_per_image = a.p.tiles_y * a.p.tiles_x;
    return 0;
}

int fs_conv2d_fwd(fs_ctx* ctx, fs_conv_desc* d) {
    if (!ctx) return fail(-1, "null ctx");
    fs::ConvArgs a;
    if (int rc = fill_conv(d, &a)) return rc;
    if (!a.x || !a.w || !a.y) return fail(-1, "fs_conv2d_fwd: null tensor");
    // 1x1 convolution with one C x C filter per sample (+ optional addend): the gradient through a Gram matrix, fs_gram.hip
    if (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.Cin == a.Cout && a.w_nstride == (long long)a.Cin * a.Cout && a.src_mode == fs::SRC_PLAIN &&
        !a.in_a && !a.bias && !a.out_relu && !a.shuffle && !a.stats && !a.mask_src && a.add_pad == 0 && a.H == a.Ho && a.W == a.Wo &&
        fs::gram_bwd2_eligible(a.N, a.H * a.W, a.Cin)) {
        const int rc0 = fs::gram_bwd2_launch(a.x, a.w, a.add_src, a.y, a.N, a.H * a.W, a.Cin, ctx->stream);
        return rc0 ? fail(rc0, "fs_conv2d_fwd: launch failed (%d)", rc0) : 0;
    }
    // ... with the max-pool routing and the ReLU mask of x itself in its epilogue (fs_conv_desc.route_src)
    if (a.route_src && a.mask_src == a.x && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.Cin == a.Cout && a.w_nstride == (long long)a.Cin * a.Cout &&
        a.src_mode == fs::SRC_PLAIN && !a.in_a && !a.bias && !a.out_relu && !a.shuffle && !a.stats && a.H == a.Ho && a.W == a.Wo &&
        fs::gram_bwd2_route_eligible(a.N, a.H, a.W, a.Cin)) {
        const int rc0 = fs::gram_bwd2_launch(a.x, a.w, a.add_src, a.y, a.N, a.H * a.W, a.Cin, ctx->stream, a.route_src, a.W);
        return rc0 ? fail(rc0, "fs_conv2d_fwd: launch failed (%d)", rc0) : 0;
    }
    if (a.route_src && !fs::conv_route_ok(a)) return fail(-2, "fs_conv2d_fwd: this launch cannot take route_src (direct kernel, no add_src / shuffle / split-K)");
    // 3x3 SAME, 64 -> 3 channels (the shape of VGG conv1_1's input gradient): vector-ALU kernel, fs_c3.hip
    const int rc = fs::conv3x3_to3_eligible(a) ? fs::conv3x3_to3_launch(a, ctx->stream) : fs::conv_launch(a, ctx->stream);
    return rc ? fail(rc, "fs_conv2d_fwd: launch failed (%d)", rc) : 0;
}

int fs_wino_transform_filter(fs_ctx* ctx, const float* w, int Cin, int Cout, float* U) {
    if (!ctx || !w || !U) return fail(-1, "fs_wino_transform_filter: null argument");
    if (Cin < 1 || Cout < 1) return fail(-2, "fs_wino_transform_filter: bad shape %dx%d", Cin, Cout);
    if (Cin % 8) return fail(-2, "fs_wino_transform_filter: Cin must be a multiple of 8 (got %d)", Cin);
    const int rc = fs::wino_gen().f2_second() ? fs::wt_wino2(w, U, Cin, Cout, ctx->stream) : fs::wt_wino(w, U, Cin, Cout, ctx->stream);
    return rc ? fail(rc, "fs_wino_transform_filter: launch failed (%d)", rc) : 0;
}

int fs_wino4t_transform_filter(fs_ctx* ctx, const float* w, int Cin, int Cout, float* U) {
    if (!ctx || !w || !U) return fail(-1, "fs_wino4t_transform_filter: null argument");
    if (Cin < 8 || Cout < 64 || (Cin % 8) || (Cout % 64)) return fail(-2, "fs_wino4t_transform_filter: Cin %% 8 == 0 and Cout %% 64 == 0 (got %dx%d)", Cin, Cout);
    const int rc = fs::wt_wino4t(w, U, Cin, Cout, ctx->stream);
    return rc ? fail(rc, "fs_wino4t_transform_filter: launch failed (%d)", rc) : 0;
}

size_t fs_wino6_filter_bytes(int Cin, int Cout) { return (Cin < 32 || Cout < 128 || (Cin % 32) || (Cout % 128)) ? 0 : fs::wino6_filter_floats(Cin, Cout) * sizeof(float); }
size_t fs_wino6_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout) {
    return (N < 1 || Ho < 1 || Wo < 1 || Cin < 1 || Cout < 1) ? 0 : fs::wino6_ws_floats(N, Ho, Wo, Cin, Cout) * sizeof(float);
}
int fs_wino6_transform_filter(fs_ctx* ctx, const float* w, int Cin, int Cout, void* U) {
    if (!ctx || !w || !U) return fail(-1, "fs_wino6_transform_filter: null argument");
    if (Cin < 32 || Cout < 128 || (Cin % 32) || (Cout % 128)) return fail(-2, "fs_wino6_transform_filter: Cin %% 32 == 0 and Cout %% 128 == 0 (got %dx%d)", Cin, Cout);
    const int rc = fs::wt_wino6(w, static_cast<unsigned short*>(U), Cin, Cout, ctx->stream);
    return rc ? fail(rc, "fs_wino6_transform_filter: launch failed (%d)", rc) : 0;
}

int fs_wino4_transform_filter(fs_ctx* ctx, const float* w, int Cin, int Cout, float* U) {
    if (!ctx || !w || !U) return fail(-1, "fs_wino4_transform_filter: null argument");
    if (Cin < 4 || Cout < 64 || (Cin % 4) || (Cout % 64)) return fail(-2, "fs_wino4_transform_filter: Cin %% 4 == 0 and Cout %% 64 == 0 (got %dx%d)", Cin, Cout);
    const int rc = fs::wt_wino4(w, U, Cin, Cout, ctx->stream);
    return rc ? fail(rc, "fs_wino4_transform_filter: launch failed (%d)", rc) : 0;
}

int fs_instnorm_finalize(fs_ctx* ctx, const float* stats, int N, int tiles, int C, int groups, const float* gamma,
                         const float* beta, float eps, float* mean, float* rstd, float* a, float* b) {
    if (!ctx || !stats || !gamma || !beta || !mean || !rstd || !a || !b) return fail(-1, "fs_instnorm_finalize: null argument");
    return fs::in_finalize(stats, N, tiles, C, groups, gamma, beta, eps, mean, rstd, a, b, ctx->stream);
}

size_t fs_instnorm_bwd_workspace_bytes(int N, int HW, int C) { return fs::in_bwd_scratch_floats(N, HW, C) * sizeof(float); }

int fs_instnorm_bwd(fs_ctx* ctx, const float* gin, const float* z, const float* mean, const float* rstd, const float* a,
                    const float* b, int mode, int N, int HW, int C, float* dz, float* dgamma, float* dbeta, void* ws,
                    size_t ws_bytes) {
    if (!ctx || !gin || !z || !mean || !rstd || !a || !b || !dz || !dgamma || !dbeta || !ws)
        return fail(-1, "fs_instnorm_bwd: null argument");
    if (C > 256) return fail(-2, "fs_instnorm_bwd: C <= 256");
    if (N < 1 || HW < 1 || C < 1 || mode < 0 || mode > 2) return fail(-2, "fs_instnorm_bwd: bad shape / mode (N=%d HW=%d C=%d mode=%d)", N, HW, C, mode);
    if (ws_bytes < fs_instnorm_bwd_workspace_bytes(N, HW, C))
        return fail(-3, "fs_instnorm_bwd: workspace too small (%zu < %zu bytes)", ws_bytes, fs_instnorm_bwd_workspace_bytes(N, HW, C));
    // the form the train step runs (round 5): partial sums as records, their reduction in the apply kernel's prologue, dgamma / dbeta from the
    // per-sample sums it leaves -- where the shape is taken (C % 4 == 0); the three-launch form otherwise
    float* S = (float*)ws + (size_t)N * fs::cdiv(HW, 64) * C * 2;   // (the tail of the workspace: [N][C][2])
    int rc = fs::in_bwd_rec(gin, z, mean, rstd, a, b, mode, dz, nullptr, 0, S, (float*)ws, N, HW, C, ctx->stream);
    if (rc == 0) {
        fs::InbParams q{};
        q.n = 1;
        q.N = N;
        q.u[0].S = S;
        q.u[0].dgamma = dgamma;
        q.u[0].dbeta = dbeta;
        q.u[0].C = C;
        rc = fs::in_bwd_params(q, ctx->stream);
    } else if (rc == 1) {
        rc = fs::in_bwd(gin, z, mean, rstd, a, b, mode, dz, dgamma, dbeta, (float*)ws, N, HW, C, ctx->stream);
    }
    return rc ? fail(rc, "fs_instnorm_bwd: launch failed (%d)", rc) : 0;
}

// ---- input gradient of a conv described by its FORWARD descriptor (im_transf_net.py:115 / vgg16.py:47 adjoint) ------------------------------------------
// dx [N,H,W,Cin] = conv2d_backprop_input(dy [N,Ho,Wo,Cout], w): the forward kernels on the flip-transposed filter (built into ws: KH*KW*Cin*Cout floats) --
// stride 1: a conv with padding K - 1 - pad over dy; stride 2: the same over the zero-dilated dy (the library's own backward takes the phase-decomposed
// form for its two stride-2 units; this entry point is the general one).  Square kernels, plain source, no on-load / epilogue options.
size_t fs_conv2d_dgrad_workspace_bytes(const fs_conv_desc* d) {
    return (!d || d->KH < 1 || d->KW < 1 || d->Cin < 1 || d->Cout < 1) ? 0 : (size_t)d->KH * d->KW * d->Cin * d->Cout * sizeof(float);
}
int fs_conv2d_dgrad(fs_ctx* ctx, fs_conv_desc* d, const float* dy, float* dx, void* ws, size_t ws_bytes) {
    if (!ctx || !d || !dy || !dx || !ws || !d->w) return fail(-1, "fs_conv2d_dgrad: null argument");
    if (d->KH != d->KW || (d->stride != 1 && d->stride != 2) || d->src_mode != FS_SRC_PLAIN || d->shuffle || d->w_nstride)
        return fail(-2, "fs_conv2d_dgrad: square kernel, stride 1 or 2, plain source");
    if (d->Cout % 4 || d->Cin % 4) return fail(-2, "fs_conv2d_dgrad: Cin and Cout must be multiples of 4 (got %d -> %d)", d->Cin, d->Cout);
    if (d->pad_mode != FS_PAD_EXPLICIT) resolve_pads(d->H, d->W, d->KH, d->KW, d->stride, d->pad_mode, d->refl, d->src_mode, &d->Ho, &d->Wo, &d->pad_t, &d->pad_l);
    if (d->Ho < 1 || d->Wo < 1) return fail(-2, "fs_conv2d_dgrad: empty output");
    if (ws_bytes < fs_conv2d_dgrad_workspace_bytes(d)) return fail(-3, "fs_conv2d_dgrad: workspace too small");
    const int K = d->KH;
    if (int rc = fs::wt_flip_transpose(d->w, (float*)ws, K, K, d->Cin, d->Cout, ctx->stream)) return fail(rc, "fs_conv2d_dgrad: launch failed (%d)", rc);
    fs::ConvArgs a{};
    a.x = dy;
    a.w = (const float*)ws;
    a.y = dx;
    a.N = d->N;
    a.H = d->Ho;
    a.W = d->Wo;
    a.Cin = d->Cout;
    a.Ho = d->H;
    a.Wo = d->W;
    a.Cout = d->Cin;
    a.KH = a.KW = K;
    a.stride = 1;
    a.pad_t = K - 1 - d->pad_t;
    a.pad_l = K - 1 - d->pad_l;
    a.src_mode = d->stride == 2 ? fs::SRC_DILATE2 : fs::SRC_PLAIN;
    a.p = fs::conv_plan(a);
    const int rc = fs::conv_launch(a, ctx->stream);
    return rc ? fail(rc, "fs_conv2d_dgrad: launch failed (%d)", rc) : 0;
}

// ---- upconv2d (im_transf_net.py:122-155: NEAREST x4, then 3x3 stride-2 SAME conv) as the transform net runs it: phase-collapsed ------------------------------
// forward: a 2x2-tap conv on the low-resolution input with the pre-summed filters of the four output parities + pixel-shuffle store (9 instead of 36 taps per
// output quad, no 16x intermediate); input gradient: a 3x3 stride-2 conv over dy with the collapsed flip-transposed filter; filter gradient: a 2x2-tap filter
// gradient on the pixel-unshuffled dy folded back to 3x3.  x [N,H,W,Cin], w / dw [3,3,Cin,Cout], y / dy [N,2H,2W,Cout]; Cin % 4 == 0, Cout % 4 == 0.
size_t fs_resizeconv_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    if (N < 1 || H < 1 || W < 1 || Cin < 4 || Cout < 4 || (Cin % 4) || (Cout % 4)) return 0;
    fs::WgradArgs a{};
    a.N = N;
    a.H = a.Ho = H;
    a.W = a.Wo = W;
    a.Cin = Cin;
    a.Cout = 4 * Cout;
    a.KH = a.KW = 2;
    a.stride = 1;
    a.dy_unshuffle = 1;
    a.p = fs::wgrad_plan(a);
    const size_t slabs = (size_t)a.p.n_slabs * a.p.K * a.Cout;
    return ((size_t)16 * Cin * Cout * 2 + slabs) * sizeof(float);   // [collapsed filter (16 Cin Cout)] [reduced collapsed gradient (16 Cin Cout)] [partial slabs]
}
static int resizeconv_check(const char* fn, fs_ctx* ctx, const void* p0, const void* p1, const void* p2, const void* ws, size_t ws_bytes, int N, int H, int W, int Cin,
                            int Cout) {
    if (!ctx || !p0 || !p1 || !p2 || !ws) return fail(-1, "%s: null argument", fn);
    const size_t need = fs_resizeconv_workspace_bytes(N, H, W, Cin, Cout);
    if (!need) return fail(-2, "%s: N, H, W >= 1 and Cin, Cout multiples of 4 (got %d x %d x %d, %d -> %d)", fn, N, H, W, Cin, Cout);
    if (ws_bytes < need) return fail(-3, "%s: workspace too small", fn);
    return 0;
}
int fs_resizeconv_fwd(fs_ctx* ctx, const float* x, const float* w, int N, int H, int W, int Cin, int Cout, float* y, void* ws, size_t ws_bytes) {
    if (int rc = resizeconv_check("fs_resizeconv_fwd", ctx, x, w, y, ws, ws_bytes, N, H, W, Cin, Cout)) return rc;
    float* weff = (float*)ws;
    if (int rc = fs::wt_upconv_fwd(w, weff, Cin, Cout, ctx->stream)) return fail(rc, "fs_resizeconv_fwd: launch failed (%d)", rc);
    fs::ConvArgs a{};
    a.x = x;
    a.w = weff;
    a.y = y;
    a.N = N;
    a.H = a.Ho = H;
    a.W = a.Wo = W;
    a.Cin = Cin;
    a.Cout = 4 * Cout;
    a.KH = a.KW = 2;
    a.stride = 1;
    a.shuffle = 1;
    a.p = fs::conv_plan(a);
    const int rc = fs::conv_launch(a, ctx->stream);
    return rc ? fail(rc, "fs_resizeconv_fwd: launch failed (%d)", rc) : 0;
}
int fs_resizeconv_dgrad(fs_ctx* ctx, const float* dy, const float* w, int N, int H, int W, int Cin, int Cout, float* dx, void* ws, size_t ws_bytes) {
    if (int rc = resizeconv_check("fs_resizeconv_dgrad", ctx, dy, w, dx, ws, ws_bytes, N, H, W, Cin, Cout)) return rc;
    float* v = (float*)ws;
    if (int rc = fs::wt_upconv_dgrad(w, v, Cin, Cout, ctx->stream)) return fail(rc, "fs_resizeconv_dgrad: launch failed (%d)", rc);
    fs::ConvArgs a{};
    a.x = dy;
    a.w = v;
    a.y = dx;
    a.N = N;
    a.H = 2 * H;
    a.W = 2 * W;
    a.Cin = Cout;
    a.Ho = H;
    a.Wo = W;
    a.Cout = Cin;
    a.KH = a.KW = 3;
    a.stride = 2;
    a.pad_t = a.pad_l = 1;
    a.p = fs::conv_plan(a);
    const int rc = fs::conv_launch(a, ctx->stream);
    return rc ? fail(rc, "fs_resizeconv_dgrad: launch failed (%d)", rc) : 0;
}
int fs_resizeconv_wgrad(fs_ctx* ctx, const float* x, const float* dy, int N, int H, int W, int Cin, int Cout, float* dw, void* ws, size_t ws_bytes) {
    if (int rc = resizeconv_check("fs_resizeconv_wgrad", ctx, x, dy, dw, ws, ws_bytes, N, H, W, Cin, Cout)) return rc;
    fs::WgradArgs a{};
    a.x = x;
    a.dy = dy;
    a.N = N;
    a.H = a.Ho = H;
    a.W = a.Wo = W;
    a.Cin = Cin;
    a.Cout = 4 * Cout;
    a.KH = a.KW = 2;
    a.stride = 1;
    a.dy_unshuffle = 1;
    a.p = fs::wgrad_plan(a);
    float* dweff = (float*)ws + (size_t)16 * Cin * Cout;
    a.slabs = dweff + (size_t)16 * Cin * Cout;
    if (int rc = fs::wgrad_launch(a, ctx->stream)) return fail(rc, "fs_resizeconv_wgrad: launch failed (%d)", rc);
    if (int rc = fs::reduce_slabs(a.slabs, 1, a.p.n_slabs, (size_t)a.p.K * a.Cout, 1.0f, dweff, ctx->stream)) return fail(rc, "fs_resizeconv_wgrad: launch failed (%d)", rc);
    const int rc = fs::wt_upconv_wgrad_fold(dweff, dw, Cin, Cout, ctx->stream);
    return rc ? fail(rc, "fs_resizeconv_wgrad: launch failed (%d)", rc) : 0;
}

// ---- the instance-norm output MATERIALISED (im_transf_net.py:246 with the activation behind it): out = act(a[n,c] z + b[n,c]) with a, b of
// fs_instnorm_finalize.  mode 0: none, 1: ReLU (:98, :150), 2: scaled tanh (:202-215); skip != NULL (mode 0 only; C % 4 == 0): the residual block's sum
// (:268-274) out = a z + b + T(skip[n, y + 2, x + 2, c]), skip [N,H+4,W+4,C], T = identity or ReLU(skip_a s + skip_b) when skip_a is given.
int fs_instnorm_apply(fs_ctx* ctx, const float* z, const float* a, const float* b, int N, int H, int W, int C, int mode, const float* skip,
                      const float* skip_a, const float* skip_b, float* out) {
    if (!ctx || !z || !a || !b || !out) return fail(-1, "fs_instnorm_apply: null argument");
    if (N < 1 || H < 1 || W < 1 || C < 1 || mode < 0 || mode > 2) return fail(-2, "fs_instnorm_apply: bad shape / mode");
    int rc;
    if (skip) {
        if (mode != 0 || (C % 4) || (!skip_a != !skip_b)) return fail(-2, "fs_instnorm_apply: the residual sum takes mode 0, C %% 4 == 0, skip_a and skip_b together");
        rc = fs::apply_res(z, a, b, skip, skip_a, skip_b, skip_a ? 1 : 0, out, N, H, W, C, ctx->stream);
    } else if (mode == 2)
        rc = fs::apply_tanh(z, a, b, out, N, H * W, C, ctx->stream);
    else
        rc = fs::apply_affine(z, a, b, out, N, H * W, C, mode == 1, ctx->stream);
    return rc ? fail(rc, "fs_instnorm_apply: launch failed (%d)", rc) : 0;
}

static int fill_wgrad(fs_wgrad_desc* d, fs::WgradArgs* a) {
    if (!d) return fail(-1, "null wgrad desc");
    if (d->pad_mode != FS_PAD_EXPLICIT)
        resolve_pads(d->H, d->W, d->KH, d->KW, d->stride, d->pad_mode, d->refl, d->src_mode, &d->Ho, &d->Wo, &d->pad_t,
                     &d->pad_l);
    if (d->Cin > 128 && (d->Cin % 128)) return fail(-2, "fs_conv2d_wgrad: Cin > 128 must be a multiple of 128");
    *a = fs::WgradArgs{};
    a->x = d->x;
    a->dy = d->dy;
    a->N = d->N;
    a->H = d->H;
    a->W = d->W;
    a->Cin = d->Cin;
    a->Ho = d->Ho;
    a->Wo = d->Wo;
    a->Cout = d->Cout;
    a->KH = d->KH;
    a->KW = d->KW;
    a->stride = d->stride;
    a->pad_t = d->pad_t;
    a->pad_l = d->pad_l;
    a->src_mode = d->src_mode;
    a->refl = d->refl;
    a->in_a = d->in_a;
    a->in_b = d->in_b;
    a->in_nstride = d->in_per_sample ? d->Cin : 0;
    a->in_relu = d->in_relu;
    a->per_sample = d->per_sample;
    a->p = fs::wgrad_plan(*a);
    return 0;
}

// the Gram case of fs_conv2d_wgrad (per-sample 1x1 "filter gradient" of a tensor with itself): streaming kernel, fs_gram.hip
static bool is_gram2(const fs::WgradArgs& a) {
    return a.per_sample && a.x == a.dy && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.Cin == a.Cout && !a.in_a && !a.dy_a &&
           a.src_mode == fs::SRC_PLAIN && a.H == a.Ho && a.W == a.Wo && fs::gram2_eligible(a.N, a.H * a.W, a.Cin);
}

size_t fs_conv2d_wgrad_workspace_bytes(fs_wgrad_desc* d) {
    fs::WgradArgs a;
    if (fill_wgrad(d, &a)) return 0;
    if (is_gram2(a)) return fs::gram2_slab_floats(a.N, a.H * a.W, a.Cin) * sizeof(float);
    fs::WgwArgs ww;
    if (const size_t f = fs::wgw_plan(&a, 1, &ww)) return f * sizeof(float);      // Winograd filter gradient (fs_wgw.hip)
    fs::Wg2Args w2;
    if (const size_t f = fs::wgrad2_plan(&a, 1, &w2)) return f * sizeof(float);   // second-generation kernel (fs_wgrad2.hip)
    return (size_t)(a.per_sample ? a.N : 1) * a.p.n_slabs * a.p.K * a.Cout * sizeof(float);
}

int fs_conv2d_wgrad(fs_ctx* ctx, fs_wgrad_desc* d, void* ws, size_t ws_bytes) {
    if (!ctx || !ws) return fail(-1, "fs_conv2d_wgrad: null argument");
    fs::WgradArgs a;
    if (int rc = fill_wgrad(d, &a)) return rc;
    if (!a.x || !a.dy || !d->dw) return fail(-1, "fs_conv2d_wgrad: null tensor");
    if (is_gram2(a)) {
        if (ws_bytes < fs::gram2_slab_floats(a.N, a.H * a.W, a.Cin) * sizeof(float)) return fail(-3, "fs_conv2d_wgrad: workspace too small");
        const int rc2 = fs::gram2_launch(a.x, d->dw, (float*)ws, a.N, a.H * a.W, a.Cin, d->scale, ctx->stream);
        return rc2 ? fail(rc2, "fs_conv2d_wgrad: launch failed (%d)", rc2) : 0;
    }
    fs::WgwArgs ww;
    if (const size_t f = fs::wgw_plan(&a, 1, &ww)) {
        if (ws_bytes < f * sizeof(float)) return fail(-3, "fs_conv2d_wgrad: workspace too small");
        float* out[1] = {d->dw};
        const int rc2 = fs::wgw_run(ww, (float*)ws, out, d->scale, ctx->stream);
        return rc2 ? fail(rc2, "fs_conv2d_wgrad: launch failed (%d)", rc2) : 0;
    }
    fs::Wg2Args w2;
    if (const size_t f = fs::wgrad2_plan(&a, 1, &w2)) {
        if (ws_bytes < f * sizeof(float)) return fail(-3, "fs_conv2d_wgrad: workspace too small");
        float* out[1] = {d->dw};
        const int rc2 = fs::wgrad2_run(w2, (float*)ws, out, d->scale, ctx->stream);
        return rc2 ? fail(rc2, "fs_conv2d_wgrad: launch failed (%d)", rc2) : 0;
    }
    const size_t need = (size_t)(a.per_sample ? a.N : 1) * a.p.n_slabs * a.p.K * a.Cout * sizeof(float);
    if (ws_bytes < need) return fail(-3, "fs_conv2d_wgrad: workspace too small");
    a.slabs = (float*)ws;
    int rc = fs::wgrad_launch(a, ctx->stream);
    if (rc) return fail(rc, "fs_conv2d_wgrad: launch failed (%d)", rc);
    return fs::reduce_slabs(a.slabs, a.per_sample ? a.N : 1, a.p.n_slabs, (size_t)a.p.K * a.Cout, d->scale, d->dw, ctx->stream);
}

// ---- utils.get_grams (utils.py:66-83) and its gradient as named entry points -------------------------------------------
static int gram_desc(const float* F, int N, int HW, int C, float* G, fs_wgrad_desc* d) {
    if (N < 1 || HW < 1 || C < 4 || (C % 4) || (C > 128 && (C % 128))) return fail(-2, "fs_gram: C must be a multiple of 4, of 128 beyond 128 (got %d)", C);
    *d = fs_wgrad_desc{};
    d->x = d->dy = F;
    d->dw = G;
    d->N = N;
    d->H = d->Ho = 1;
    d->W = d->Wo = HW;
    d->Cin = d->Cout = C;
    d->KH = d->KW = d->stride = 1;
    d->pad_mode = FS_PAD_EXPLICIT;
    d->per_sample = 1;
    d->scale = 1.0f / ((float)HW * (float)C);
    return 0;
}

size_t fs_gram_workspace_bytes(int N, int HW, int C) {
    fs_wgrad_desc d;
    if (gram_desc(nullptr, N, HW, C, nullptr, &d)) return 0;
    const size_t fwd = fs_conv2d_wgrad_workspace_bytes(&d);
    const size_t bwd = (size_t)N * C * C * sizeof(float);   // S = (dG + dG^T) / (HW C)
    return fwd > bwd ? fwd : bwd;
}

int fs_gram_fwd(fs_ctx* ctx, const float* F, int N, int HW, int C, float* G, void* ws, size_t ws_bytes) {
    if (!ctx || !F || !G || !ws) return fail(-1, "fs_gram_fwd: null argument");
    fs_wgrad_desc d;
    if (int rc = gram_desc(F, N, HW, C, G, &d)) return rc;
    return fs_conv2d_wgrad(ctx, &d, ws, ws_bytes);
}

int fs_gram_bwd(fs_ctx* ctx, const float* F, const float* dG, int N, int HW, int C, float* dF, void* ws, size_t ws_bytes) {
    if (!ctx || !F || !dG || !dF || !ws) return fail(-1, "fs_gram_bwd: null argument");
    fs_wgrad_desc d;
    if (int rc = gram_desc(F, N, HW, C, nullptr, &d)) return rc;
    if (ws_bytes < (size_t)N * C * C * sizeof(float)) return fail(-3, "fs_gram_bwd: workspace too small");
    float* S = (float*)ws;
    if (int rc = fs::gram_symmetrize(dG, S, N, C, d.scale, ctx->stream)) return fail(rc, "fs_gram_bwd: launch failed (%d)", rc);
    if (fs::gram_bwd2_eligible(N, HW, C)) {
        const int rc = fs::gram_bwd2_launch(F, S, nullptr, dF, N, HW, C, ctx->stream);
        return rc ? fail(rc, "fs_gram_bwd: launch failed (%d)", rc) : 0;
    }
    fs::ConvArgs a{};   // 1x1 convolution with one C x C filter per sample
    a.x = F;
    a.w = S;
    a.w_nstride = (long long)C * C;
    a.y = dF;
    a.N = N;
    a.H = a.Ho = 1;
    a.W = a.Wo = HW;
    a.Cin = a.Cout = C;
    a.KH = a.KW = a.stride = 1;
    a.p = fs::conv_plan(a);
    const int rc = fs::conv_launch(a, ctx->stream);
    return rc ? fail(rc, "fs_gram_bwd: launch failed (%d)", rc) : 0;
}

}  // extern "C"
