// VGG16 feature extractor + Gram + perceptual losses: host orchestration (see fs_vgg.hip).
#pragma once
#include "../../include/faststyle_hip.h"
#include "fs_kernels.h"

namespace fs {

struct VggLayout {
    int N;        // samples that carry gradient (the transform-net outputs)
    int NB;       // samples pushed through the shared layers (2N when content targets ride along)
    int H, W;
    int lmax;     // last conv layer evaluated
    int cmax;     // last layer the content half is needed for (-1: none)
    unsigned content_mask;   // bit l: a content term reads the content half of act[l] (its full-resolution store must not be skipped)
    int Hl[FS_VGG_NLAYERS], Wl[FS_VGG_NLAYERS];
    size_t xin, ab, act[FS_VGG_NLAYERS], pool[3];
    size_t gram[4], sm[4], slabs;
    size_t gslab[4];   // per style layer: the partial slabs of its streaming Gram kernel (the four layers are finished by ONE launch: all must exist at once)
    size_t lossp, lossp_floats;   // partial sums of every loss term of a step, summed by loss_finish
    size_t d_pre, d_in[2], d_tap, d_tap2, scratch;
    size_t splitws, splitws_floats;  // split-K partial sums of the deep, small-grid convs (conv4_x at batch 4)
    size_t w6ws, w6ws_floats;        // scratch of the split-bf16 F(4x4) pipeline (fs_wino6.hip; FS_WINO_V=6 only)
    size_t total_floats;
};

size_t vgg_prepared_floats();
int vgg_prepare(const float* const w[FS_VGG_NLAYERS], float* prepared, hipStream_t s);   // < 0: error; >= 0: mask of the kernel generations whose filter layouts the buffer carries
unsigned vgg_prep_mask();   // the mask a prepare under the knobs of the moment builds
void vgg_layout(int N, int H, int W, const fs_loss_cfg& cfg, bool with_content, VggLayout* L);
int perceptual_loss(const VggLayout& L, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS],
                    const float* prepared, const fs_loss_cfg& cfg, const float* y, const float* content, float* losses,
                    float* dy, float* ws, hipStream_t s, unsigned prep_mask, const struct StreamAux* aux = nullptr);
int style_targets(const VggLayout& L, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS],
                  const fs_loss_cfg& cfg, const float* img, float* const grams[4], float* ws, hipStream_t s);
// libs/vgg16.py:36-220 for N images (RGB 0..255): post-ReLU activations of the requested layers copied to out[i]
int vgg_features(const VggLayout& L, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS], const float* x,
                 int n_layers, const int* layers, float* const* out, float* ws, hipStream_t s);
int vgg_consts(float* ab, hipStream_t s);
// adjoint of vgg_features: upstream gradients of any subset of layers -> dL/d(images); the forward is recomputed into ws
int vgg_dgrad(const VggLayout& L, const float* const w[FS_VGG_NLAYERS], const float* const b[FS_VGG_NLAYERS], const float* prepared, unsigned prep_mask,
              const float* x, int n_layers, const int* layers, const float* const* dfeat, float* dx, float* ws, float* flipt_scratch, hipStream_t s);

}  // namespace fs
