/* libfaststyle_hip.so -- C ABI of the MI355X (gfx950) fast-style-transfer hot path.
 *
 * The reference (ghwatson/faststyle) has no FFI of its own: every FLOP of the path is a stock
 * TensorFlow-1 op called from Python.  This header is the boundary a maintainer binds instead
 * (ctypes stub: INTEGRATION.md / faststyle_amd/_lib.py); each entry point names the reference
 * call site it replaces (file:line relative to the reference repo).
 *
 * Conventions
 *   - every call returns 0 on success, a negative code on error; fs_last_error() (thread-local)
 *     holds the message;
 *   - the caller owns ALL tensor memory (device pointers: hipMalloc / torch data_ptr()); the
 *     library never allocates or frees it; scratch comes from fs_*_workspace_bytes() + a
 *     caller-provided workspace;
 *   - every launch is asynchronous on the ctx stream, no hidden synchronisation, no allocation
 *     after fs_ctx_create (hipGraph-capturable);
 *   - a ctx is not thread-safe; distinct ctxs are independent;
 *   - tensors are NHWC fp32, filters HWIO fp32 (TensorFlow layout).
 */
#ifndef FASTSTYLE_HIP_H
#define FASTSTYLE_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fs_ctx fs_ctx;

int fs_ctx_create(int device, void* hip_stream, fs_ctx** out);
void fs_ctx_destroy(fs_ctx* ctx);
int fs_ctx_set_stream(fs_ctx* ctx, void* hip_stream);
const char* fs_last_error(void);
const char* fs_version(void);

/* ---- measurement hook (bench.py): HIP events around every MFMA-kernel launch on the ctx stream.
 * out[f*3+{0,1,2}] = {launches, FLOPs executed, milliseconds} of row f; fs_profile_family_name(f) is the kernel symbol the
 * row belongs to (one row per symbol, so a row can be re-derived from a `rocprofv3 --kernel-trace --stats` summary; the
 * one symbol shared by two workloads, wino2_conv_kernel, has a row per caller), "" for unused rows. */
#define FS_PROFILE_FAMILIES 23
int fs_profile_begin(fs_ctx* ctx);
int fs_profile_end(fs_ctx* ctx, double out[3 * FS_PROFILE_FAMILIES]);
const char* fs_profile_family_name(int family);

/* ---- image-transform net: reference im_transf_net.py:14-75 (create_net) ------------------ */
#define FS_TNET_NPARAMS 424102 /* 48 fp32 tensors, sorted-key (= checkpoint) order */
#define FS_TNET_NTENSORS 48
#define FS_FLAG_SAVE_FOR_BWD 1 /* keep every intermediate the backward pass needs */
#define FS_FLAG_BF16 4            /* mixed-precision inference (BASELINE config 5): bf16 activations/weights in HBM and LDS,
                                    * bf16 MFMA with fp32 accumulation, fp32 instance-norm statistics; image in and out stay
                                    * fp32.  Forward only (not with SAVE_FOR_BWD), resize-conv models.  Error vs the fp32 path
                                    * is ~1e-2 of the pixel range -- NOT the 1e-3 parity bar, which only the fp32 path meets. */
#define FS_FLAG_UPSAMPLE_DECONV 2 /* --upsample_method deconv (im_transf_net.py:57-63): the three upsample_* filters are
                                     [K,K,Cout,Cin] conv2d_transpose filters (same element counts and offsets) */
#define FS_FLAG_PARAMS_FROZEN 8   /* fs_tnet_forward, fp32: the caller promises that `params` holds the values it held at this context's
                                   * previous fs_tnet_forward call and that nobody wrote to `ws` in between (stylize_image.py /
                                   * stylize_webcam.py: one checkpoint, many frames).  The re-laid-out filters (collapsed resize-conv,
                                   * folded output layer, Winograd transforms) live in `ws`; when params pointer, ws pointer and
                                   * plan are those of the previous call the kernels that rebuild them are skipped -- any other
                                   * call rebuilds them as usual, so the first call of a sequence is always complete. */

/* name / offset (floats) / shape of the idx-th parameter tensor in the flat buffer; the order is
 * the key order of the TF bundle (models/<style>_final.ckpt.index), without the "img_t_net/" scope. */
int fs_tnet_param_info(int idx, const char** name, int* offset, int* ndim, int dims[4]);
/* im_transf_net.py:14-75: output size of create_net for an HxW input. */
int fs_tnet_out_shape(int H, int W, int* Ho, int* Wo);
size_t fs_tnet_workspace_bytes(int N, int H, int W, int flags);
/* y[N,Ho,Wo,3] = create_net(x[N,H,W,3], 'resize') with the 48 tensors in params[FS_TNET_NPARAMS].
 * Replaces sess.run(Y, {X: img}) at stylize_image.py:75 and the forward half of train.py:256-275. */
int fs_tnet_forward(fs_ctx* ctx, const float* params, const float* x, int N, int H, int W, float* y, void* ws,
                    size_t ws_bytes, int flags);
/* Forget what FS_FLAG_PARAMS_FROZEN remembers (params pointer, workspace pointer, plan): the next fs_tnet_forward rebuilds the re-laid-out
 * filters whatever its pointers are.  Call it whenever a workspace or a parameter buffer the context has seen is freed, re-used for something
 * else or rewritten behind the library's back (an allocator handing the same address out again would otherwise look like "nothing changed"). */
int fs_tnet_invalidate(fs_ctx* ctx);
/* grads[FS_TNET_NPARAMS] = d loss / d params given dy = d loss / d y; `ws` must be the workspace a
 * fs_tnet_forward(..., FS_FLAG_SAVE_FOR_BWD) call on the same inputs just filled.  Replaces the
 * transform-net half of AdamOptimizer.minimize's gradient graph (train.py:203).  Error -5: this context filled `ws` with a forward of
 * another shape or another upsample method.
 * Contract: `params` must hold, bit for bit, the values that forward read -- the FS_FLAG_SAVE_FOR_BWD forward also builds the
 * re-laid-out input-gradient filters, and the FIRST backward on (`ws`, `params`) uses them as they are (a second one rebuilds them).
 * A caller that rewrites `params` in place between the two (a finite-difference probe, a delayed optimiser update) calls
 * fs_tnet_invalidate first; a re-plan (another shape, fs_debug_reload_env) drops them by itself. */
int fs_tnet_backward(fs_ctx* ctx, const float* params, const float* x, const float* dy, int N, int H, int W, float* grads,
                     void* ws, size_t ws_bytes, int flags);

/* ---- VGG16 + Gram + losses: reference libs/vgg16.py:36-220, utils.py:66-83, losses.py ------ */
#define FS_VGG_NLAYERS 10 /* conv1_1 .. conv4_3 (conv5_x is never fetched: train.py:55-59) */
/* floats needed for the derived forms of the 10 frozen filters: flip-transposed (input-gradient convs) and, for conv1_2 .. conv4_3 in both
 * orientations, the Winograd-transformed filters in the layout(s) of the kernel generation the tuning knobs select AT THE TIME OF THE CALL (round 6; the
 * default, FS_WINO_V=6: the F(4x4,3x3) register layouts of fs_wino4t.hip + the bf16 pieces of fs_wino6.hip for conv4_x: about 110 M floats).  Call it
 * immediately before fs_vgg_prepare.  A buffer prepared under other knobs than a later fs_perceptual_loss runs with stays valid: generations it lacks fall
 * back to the direct kernels. */
size_t fs_vgg_prepared_floats(void);
/* w[i], b[i]: conv1_1,conv1_2,conv2_1,...,conv4_3 in the npz convention of vgg16.load_weights
 * (vgg16.py:257-266: HWIO [3,3,Cin,Cout] kernels, [Cout] biases).  Fills `prepared` with the
 * flipped/transposed filters the input-gradient convs use and the Winograd-transformed filters of the 3x3
 * stride-1 convs (VGG is frozen: train.py:198-199, so this runs once per weight set). */
int fs_vgg_prepare(fs_ctx* ctx, const float* const w[FS_VGG_NLAYERS], float* prepared);

typedef struct {
    int n_content;              /* content layers (default 1: conv3_3, train.py:52-55) */
    int content_layer[4];       /* layer index 0..9 */
    float content_weight[4];
    int n_style;                /* style layers (default conv1_2,conv2_2,conv3_3,conv4_3) */
    int style_layer[4];
    float style_weight[4];
    const float* target_gram[4]; /* [C,C] of the style image, broadcast over the batch (losses.py:63) */
    float beta;                 /* TV weight (train.py:88-93) */
} fs_loss_cfg;

size_t fs_perceptual_workspace_bytes(int N, int H, int W, const fs_loss_cfg* cfg);
/* One evaluation of loss = content + style + beta*tv (train.py:164-184) AND its gradient wrt y:
 *   y[N,H,W,3]        transform-net output fed to VGG directly (train.py:164-165)
 *   content[N,H,W,3]  the raw batch whose VGG features are the content targets (train.py:250-251)
 *   losses[4]         device floats: {loss, content_loss, style_loss, beta*tv_loss}
 *   dy[N,H,W,3]       d loss / d y  (VGG filter gradients are never formed) */
int fs_perceptual_loss(fs_ctx* ctx, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS],
                       const float* prepared, const fs_loss_cfg* cfg, const float* y, const float* content, int N, int H,
                       int W, float* losses, float* dy, void* ws, size_t ws_bytes);
/* utils.get_grams on the style image (train.py:144-151): grams[i] = [C_i,C_i] for cfg->style_layer[i]. */
size_t fs_style_targets_workspace_bytes(int H, int W);
int fs_style_targets(fs_ctx* ctx, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS],
                     const fs_loss_cfg* cfg, const float* style_img, int H, int W, float* const grams[4], void* ws,
                     size_t ws_bytes);

/* ---- builder-level pieces (the reference's Python helpers, for scripts that compose their own loss) ----
 * fs_vgg_features: libs/vgg16.py:36-220 -- the post-ReLU tensors `vgg/convX_Y:0` of N images (RGB 0..255, the
 *   ImageNet mean is subtracted inside); layers[i] in 0..9 = conv1_1..conv4_3; out[i] device [N,H_l,W_l,C_l].
 * fs_gram_fwd: utils.get_grams (utils.py:66-83) on one such tensor F[N,HW,C] (the [b,h*w,c] reshape of utils.py:76-77):
 *   G[n] = F[n]^T F[n] / (HW*C), [N,C,C], exactly symmetric.  C: a multiple of 4, of 128 beyond 128.
 * fs_gram_bwd: the gradient through it, dF[n] = F[n] (dG[n] + dG[n]^T) / (HW*C) for an upstream dG[N,C,C] (what
 *   tf.gradients forms behind losses.style_loss, losses.py:61-64).  ws: fs_gram_workspace_bytes(N,HW,C) for both.
 *   (Inside fs_perceptual_loss the same two kernels run with the symmetric factor folded into the loss kernel.)
 * fs_loss_sqdiff: out[0] = scale * sum_i (x[i] - t[i % t_period])^2 -- losses.content_loss (losses.py:32-37, scale
 *   w/(h*w*c)) and losses.style_loss (losses.py:61-64, scale w/(c*c), t = the [1,c,c] target broadcast over the batch).
 * fs_loss_tv: losses.tv_loss (losses.py:70-97).  scratch: 4096 bytes of device memory; out: device scalar. */
size_t fs_vgg_features_workspace_bytes(int N, int H, int W, int max_layer);
int fs_vgg_features(fs_ctx* ctx, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS], const float* x,
                    int N, int H, int W, int n_layers, const int* layers, float* const* out, void* ws, size_t ws_bytes);
size_t fs_gram_workspace_bytes(int N, int HW, int C);
int fs_gram_fwd(fs_ctx* ctx, const float* F, int N, int HW, int C, float* G, void* ws, size_t ws_bytes);
int fs_gram_bwd(fs_ctx* ctx, const float* F, const float* dG, int N, int HW, int C, float* dF, void* ws, size_t ws_bytes);
int fs_loss_sqdiff(fs_ctx* ctx, const float* x, const float* t, size_t t_period, size_t n, float scale, float* out,
                   void* scratch);
int fs_loss_tv(fs_ctx* ctx, const float* x, int N, int H, int W, int C, float* out, void* scratch);
/* ---- round 6: value + gradient forms and the adjoint of fs_vgg_features -- the pieces a script differentiates through when it composes its own objective the
 * way train.py:171-204 and slow_style.py:140-176 do (faststyle_amd/autograd.py wraps each as a torch.autograd.Function; fs_perceptual_loss remains the fused
 * fixed-form path the training step runs). */
/* out[0] = scale * sum((x - t)^2) as fs_loss_sqdiff, and grad = 2 * scale * (x - t[i % t_period]) WRITTEN (n floats): losses.py:32-37 / :61-64 with their
 * derivative.  scratch: 1024 floats. */
int fs_loss_sqdiff_grad(fs_ctx* ctx, const float* x, const float* t, size_t t_period, size_t n, float scale, float* out, float* grad, void* scratch);
/* out[0] = scale * TV(x) (losses.py:70-97) and grad (+)= scale * dTV/dx: written when accumulate == 0, added to otherwise (train.py:184's beta * tv term on
 * top of a gradient that already sits in `grad`).  scratch: 1024 floats. */
int fs_loss_tv_grad(fs_ctx* ctx, const float* x, int N, int H, int W, int C, float scale, float* out, float* grad, int accumulate, void* scratch);
/* Adjoint of fs_vgg_features (libs/vgg16.py:36-173 behind tf.gradients, train.py:203): dfeat[i] = dL/d(post-ReLU activation of layers[i]), [N,Hl,Wl,Cl], for
 * any subset of conv1_1 .. conv4_3 (each layer at most once) -> dx = dL/d(x) [N,H,W,3].  The forward is recomputed into the workspace (the weights are frozen);
 * `prepared` (fs_vgg_prepare) selects the Winograd kernels fs_perceptual_loss runs, NULL the direct ones.  ReLU gradient by the sign of the activation, max-pool
 * gradient to the first maximum of each window (TF MaxPoolGrad), SAME padding -- the same launches as the backward half of fs_perceptual_loss. */
size_t fs_vgg_dgrad_workspace_bytes(int N, int H, int W, int max_layer);
int fs_vgg_dgrad(fs_ctx* ctx, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS], const float* prepared, const float* x, int N,
                 int H, int W, int n_layers, const int* layers, const float* const* dfeat, float* dx, void* ws, size_t ws_bytes);

/* ---- optimiser: tf.train.AdamOptimizer (train.py:203), TF1 form -- lr_t = lr*sqrt(1-b2^t)/(1-b1^t), epsilon OUTSIDE the
 * bias correction: theta -= lr_t * m / (sqrt(v) + eps) ------------------------------------------------------------------ */
int fs_adam_tf_step(fs_ctx* ctx, float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                    float beta2, float eps, long long t /* 1-based step */);

/* ---- inspection (parity tests, debugging): where the composite calls leave their saved tensors inside the CALLER's
 * workspace.  Offsets are in floats from the start of `ws`; the layout is a pure function of the arguments (and of the
 * library's tuning knobs as cached at the time), so these need no ctx and launch nothing.
 * fs_tnet_ws_tensor: after fs_tnet_forward(..., FS_FLAG_SAVE_FOR_BWD).  unit 0..15 = initconv_0..2, resblock_k conv 1 / 2
 *   (3+2k, 4+2k), upsample_0..2.  what: FS_TNET_WS_Z the raw conv output z [N,Ho,Wo,C] (before the instance norm);
 *   _A / _B / _MEAN / _RSTD the per-sample constants [N,C] (the unit's activation is relu(a*z + b), a = gamma*rstd,
 *   b = beta - mean*a; im_transf_net.py:238-245); FS_TNET_WS_H (unit = k in 0..4) the output of residual block k.
 * fs_perceptual_ws_tensor: after fs_perceptual_loss.  The post-ReLU activations `vgg/convX_Y:0` of layer 0..9, [NB,h,w,c]:
 *   the first N samples belong to y, the next N (layers up to the last content layer only) to the content batch. */
#define FS_TNET_WS_Z 0
#define FS_TNET_WS_A 1
#define FS_TNET_WS_B 2
#define FS_TNET_WS_MEAN 3
#define FS_TNET_WS_RSTD 4
#define FS_TNET_WS_H 5
int fs_tnet_ws_tensor(int N, int H, int W, int flags, int unit, int what, size_t* offset_floats, int dims[4]);
int fs_perceptual_ws_tensor(int N, int H, int W, const fs_loss_cfg* cfg, int layer, size_t* offset_floats, int dims[4]);
/* fs_perceptual_ws_input: where fs_perceptual_loss stages y and content ([y ; content] is ONE 2N batch through the shared VGG layers).  Passing
 *   y == ws + *y_offset_floats and / or content == ws + *content_offset_floats skips the corresponding device copy: the transform net can write
 *   its output, and the input pipeline its batch, straight into the workspace (train.py:250-256 without the sess.run round trip).
 *   *content_offset_floats = (size_t)-1 when cfg has no content layer. */
int fs_perceptual_ws_input(int N, int H, int W, const fs_loss_cfg* cfg, size_t* y_offset_floats, size_t* content_offset_floats);

/* ---- single ops (used by the parity tests; same kernels the composite calls launch) --------- */
#define FS_PAD_SAME 0
#define FS_PAD_VALID 1
#define FS_PAD_EXPLICIT 2
#define FS_SRC_PLAIN 0
#define FS_SRC_REFLECT 1
#define FS_SRC_DILATE2 2
typedef struct {
    const float* x;      /* [N,H,W,Cin] */
    const float* w;      /* [KH,KW,Cin,Cout] */
    float* y;            /* [N,Ho,Wo,Cout] */
    int N, H, W, Cin, Cout, KH, KW, stride;
    int pad_mode, pad_t, pad_l, Ho, Wo; /* Ho/Wo/pad_* are inputs only with FS_PAD_EXPLICIT */
    int src_mode, refl;
    const float* in_a;   /* optional producer instance-norm folded into the load: relu(x*a+b) */
    const float* in_b;
    int in_per_sample;   /* in_a/in_b are [N,Cin] (else [Cin]) */
    int in_relu;
    const float* bias;   /* optional [Cout] */
    int out_relu;
    int shuffle;         /* 2x2 pixel-shuffle store: y is [N,2Ho,2Wo,Cout/4] */
    float* stats;        /* optional per-tile instance-norm partials, see fs_conv2d_stats_floats */
    const float* add_src;
    int add_pad;
    long long w_nstride; /* per-sample filter stride in floats (0: shared) */
    const float* w_wino; /* optional: the same filter as transformed by fs_wino_transform_filter (16*Cin*Cout floats, layout
                          * private to the library); an eligible conv -- 3x3, stride 1, SAME / VALID / 'full' padding,
                          * Cin % 8 == 0, Cin <= 128, Cout % 64 == 0 -- then runs on the second-generation Winograd
                          * F(2x2,3x3) kernel, the one the 64/128-channel VGG16 convs of fs_perceptual_loss and the residual
                          * convs of the transform net take.  Any other shape (Cin > 128 included: the composite calls run
                          * those layers on the first-generation kernel, whose filter order this descriptor does not carry)
                          * silently takes the direct kernel -- same result within the parity tolerance. */
    const float* w_wino4; /* optional: the filter as transformed by fs_wino4_transform_filter (36*Cin*Cout floats): a 3x3
                           * stride-1 SAME conv with Cin % 4 == 0 and Cout % 64 == 0 then runs on the Winograd F(4x4,3x3) kernel
                           * (fs_wino4.hip), the one fs_perceptual_loss runs conv1_2 ... conv4_3 and their input gradients on
                           * (reference libs/vgg16.py:45-173).  Takes precedence over w_wino; other shapes fall through. */
    const float* mask_src; /* optional [N,Ho,Wo,Cout]: y = mask_src > 0 ? y : 0 -- the consumer-ReLU mask of the VGG input-gradient
                            * convs (the gradient of vgg16.py:48's tf.nn.relu folded into the conv in front of it) */
    float* pool_out;       /* optional [N,Ho/2,Wo/2,Cout] (Winograd kernels only, even Ho and Wo): tf.nn.max_pool 2x2/2 of the
                            * stored result (vgg16.py:68,104,154) written by the same launch */
    const float* w_wino4t; /* optional: the filter as transformed by fs_wino4t_transform_filter (36*Cin*Cout floats): a 3x3 stride-1
                            * conv with padding 0, 1 or 2 (VALID / SAME / 'full'), Cin % 8 == 0, Cout % 64 == 0, no bias / activation /
                            * mask / pool then runs on the 16-tile Winograd F(4x4,3x3) kernel (fs_wino4t.hip) -- the one fs_tnet_forward
                            * / fs_tnet_backward run the ten residual convs (im_transf_net.py:250-276) and their input gradients on:
                            * in_a / in_b (+ in_relu) on load with padding 0, stats, add_src as for the other kernels.  w_wino4 wins
                            * when both are given; other shapes fall through. */
    /* optional, with w_wino4t and a plan of 16 x 16-pixel items (fs_conv2d_plan: tiles_per_image = ceil(Ho/16) * ceil(Wo/16)), raw or add_src
     * epilogue only: y is the gradient g wrt the OUTPUT of an instance-norm unit whose raw conv output is inb_z [N,Ho,Wo,Cout]; the launch also
     * leaves that unit's instance-norm-backward partial sums (im_transf_net.py:218-247 adjoint) per item,
     *   inb_rec[n][item][c] = { sum g', sum g' * (z - mean[n][c]) * rstd[n][c] },  g' = g where relu(a z + b) > 0 when inb_relu, else g,
     * [N][tiles_per_image][Cout][2] floats -- what fs_tnet_backward has the residual input-gradient launches produce for the unit below them.
     * Any other plan with inb_rec set is an error (-2). */
    const float* inb_z;
    const float* inb_mean; /* [N][Cout] each */
    const float* inb_rstd;
    const float* inb_a;    /* with inb_relu */
    const float* inb_b;
    int inb_relu;
    float* inb_rec;
    /* optional, with mask_src, no add_pad: [N,ceil(Ho/2),ceil(Wo/2),Cout], the gradient of tf.nn.max_pool 2x2/2 SAME over mask_src
     * (vgg16.py:68,104,154); it is routed to the FIRST maximum of every window (TF MaxPoolGrad) and added before the mask:
     *   y = mask_src > 0 ? y + (this pixel is its window's arg-max ? route_src : 0) : 0
     * -- what fs_perceptual_loss has the Gram-gradient launch of relu1_2 / relu2_2 / relu3_3 do (a 1x1 conv with one C x C filter per sample over
     * x = mask_src): the streaming kernel of fs_gram.hip for C = 64 / 128 / 256, even Ho and Wo a multiple of 128 (C = 64), 64 (C = 128) or 32 (add_src allowed),
     * else the direct kernel (no add_src); a launch that cannot take it is an error (-2). */
    const float* route_src;
    /* optional (round 6), all three together: the filter as split by fs_wino6_transform_filter, and scratch of w6_ws_bytes (fs_wino6_workspace_bytes = one
     * pass; less makes the launch run in chunks of >= 128 tiles of 4 x 4 outputs).  A 3x3 stride-1 conv with padding 0, 1 or 2, Cin % 32 == 0, Cout % 128 == 0,
     * epilogue raw | bias / ReLU / pool_out | mask_src, of at least FS_WINO6_MINTILES tiles and FS_WINO6_MINCC = Cin * Cout then runs as the split-bf16
     * Winograd F(4x4,3x3) pipeline of fs_wino6.hip (input transform, 36 GEMMs on the bf16 matrix cores as six exact products of bf16 pieces with fp32
     * accumulation, output transform) -- what fs_perceptual_loss runs conv4_x (libs/vgg16.py:131-173) and its input gradients on under FS_WINO_V=6.
     * It wins over the other Winograd layouts when given; other shapes fall through. */
    const void* w_wino6;
    void* w6_ws;
    size_t w6_ws_bytes;
} fs_conv_desc;
/* U = G g G^T for every (ci, co) filter g = w[:, :, ci, co] of a 3x3 HWIO filter (Lavin & Gray F(2x2,3x3)), 16 values per
 * filter, in the order the Winograd kernels stage them; the caller owns U (16*Cin*Cout floats; Cin % 8 == 0).
 * fs_vgg_prepare does this once for the frozen VGG16 filters. */
int fs_wino_transform_filter(fs_ctx* ctx, const float* w, int Cin, int Cout, float* U);
/* The same for F(4x4,3x3) (interpolation points 0, +-1, +-2, infinity; computed in float64, rounded once): 36 values per
 * filter, U holds 36*Cin*Cout floats; Cin % 4 == 0, Cout % 64 == 0. */
int fs_wino4_transform_filter(fs_ctx* ctx, const float* w, int Cin, int Cout, float* U);
/* ... in the register layout of the 16-tile kernel (fs_conv_desc.w_wino4t); Cin % 8 == 0, Cout % 64 == 0. */
int fs_wino4t_transform_filter(fs_ctx* ctx, const float* w, int Cin, int Cout, float* U);
/* ... as three bf16 pieces per element (the float64 transform rounded once to fp32, then split exactly: h + m + l == the fp32 value) in the stage order of
 * the GEMM of fs_wino6.hip (fs_conv_desc.w_wino6); U holds fs_wino6_filter_bytes(Cin, Cout) bytes; Cin % 32 == 0, Cout % 128 == 0. */
size_t fs_wino6_filter_bytes(int Cin, int Cout);
int fs_wino6_transform_filter(fs_ctx* ctx, const float* w, int Cin, int Cout, void* U);
/* scratch for one pass of that pipeline over an [N,Ho,Wo,Cout] result: V [36][tiles][Cin] + M [36][tiles][Cout] floats, tiles = N ceil(Ho/4) ceil(Wo/4) padded to 128 */
size_t fs_wino6_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout);
/* tf.nn.conv2d (im_transf_net.py:115, vgg16.py:47) on the matrix cores. */
int fs_conv2d_fwd(fs_ctx* ctx, fs_conv_desc* d);
/* tf.nn.conv2d_backprop_input for the conv a FORWARD descriptor describes (im_transf_net.py:115 / vgg16.py:47 adjoint): dx [N,H,W,Cin] from dy [N,Ho,Wo,Cout]
 * and d->w (HWIO).  Square kernels, stride 1 or 2, plain source, Cin % 4 == 0 and Cout % 4 == 0; x / y / on-load / epilogue fields of the descriptor are
 * ignored.  ws holds the flip-transposed filter (fs_conv2d_dgrad_workspace_bytes). */
size_t fs_conv2d_dgrad_workspace_bytes(const fs_conv_desc* d);
int fs_conv2d_dgrad(fs_ctx* ctx, fs_conv_desc* d, const float* dy, float* dx, void* ws, size_t ws_bytes);
/* upconv2d (im_transf_net.py:122-155: tf.image.resize_images NEAREST x4 then conv2d 3x3 stride 2 SAME), phase-collapsed as fs_tnet_forward / _backward run
 * it: x [N,H,W,Cin], w / dw [3,3,Cin,Cout] (HWIO), y / dy [N,2H,2W,Cout]; Cin % 4 == 0, Cout % 4 == 0.  One workspace size serves the three. */
size_t fs_resizeconv_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int fs_resizeconv_fwd(fs_ctx* ctx, const float* x, const float* w, int N, int H, int W, int Cin, int Cout, float* y, void* ws, size_t ws_bytes);
int fs_resizeconv_dgrad(fs_ctx* ctx, const float* dy, const float* w, int N, int H, int W, int Cin, int Cout, float* dx, void* ws, size_t ws_bytes);
int fs_resizeconv_wgrad(fs_ctx* ctx, const float* x, const float* dy, int N, int H, int W, int Cin, int Cout, float* dw, void* ws, size_t ws_bytes);
/* The instance-norm output materialised: out = act(a[n,c] z + b[n,c]) with a, b from fs_instnorm_finalize (im_transf_net.py:238-246).  mode 0: no activation,
 * 1: ReLU (:98, :150), 2: scaled tanh (:202-215).  skip != NULL (mode 0, C % 4 == 0): the residual block's sum (:268-274),
 * out = a z + b + T(skip[n, y + 2, x + 2, c]) with skip [N,H+4,W+4,C] and T = identity, or ReLU(skip_a s + skip_b) when skip_a / skip_b [N,C] are given.
 * (The fused paths never form these tensors: the consumer conv applies the affine while staging its input.) */
int fs_instnorm_apply(fs_ctx* ctx, const float* z, const float* a, const float* b, int N, int H, int W, int C, int mode, const float* skip,
                      const float* skip_a, const float* skip_b, float* out);
/* resolves Ho/Wo/pads and returns the per-image tile count the launch will use */
int fs_conv2d_plan(fs_conv_desc* d, int* tiles_per_image);
/* tf.nn.moments + normalisation constants (im_transf_net.py:238-245) from the conv epilogue's
 * per-tile partials: mean,rstd,a,b are [N,C];  a = gamma*rstd, b = beta - mean*a. */
int fs_instnorm_finalize(fs_ctx* ctx, const float* stats, int N, int tiles, int C, int groups, const float* gamma,
                         const float* beta, float eps, float* mean, float* rstd, float* a, float* b);
/* Backward of activation(inst_norm(z)) (im_transf_net.py:218-247 adjoint): gin = dL/d(activation
 * output); mode 0: no activation, 1: ReLU, 2: scaled tanh.  dz = dL/dz, dgamma/dbeta = [C]. */
size_t fs_instnorm_bwd_workspace_bytes(int N, int HW, int C);
int fs_instnorm_bwd(fs_ctx* ctx, const float* gin, const float* z, const float* mean, const float* rstd, const float* a,
                    const float* b, int mode, int N, int HW, int C, float* dz, float* dgamma, float* dbeta, void* ws,
                    size_t ws_bytes);
typedef struct {
    const float* x;
    const float* dy;
    float* dw;           /* [KH,KW,Cin,Cout], or [N,Cin,Cout] when per_sample */
    int N, H, W, Cin, Cout, KH, KW, stride, pad_mode, pad_t, pad_l, Ho, Wo, src_mode, refl;
    const float* in_a;
    const float* in_b;
    int in_per_sample, in_relu;
    int per_sample;
    float scale;
} fs_wgrad_desc;
size_t fs_conv2d_wgrad_workspace_bytes(fs_wgrad_desc* d);
/* tf.nn.conv2d_backprop_filter; with per_sample and KH=KW=1, x==dy it is utils.get_grams. */
int fs_conv2d_wgrad(fs_ctx* ctx, fs_wgrad_desc* d, void* ws, size_t ws_bytes);

#ifdef __cplusplus
}
#endif
#endif
