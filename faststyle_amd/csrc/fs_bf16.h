// bf16 mixed-precision inference path of the transform net (fs_bf16.hip).
#pragma once
#include "fs_tnet.h"

namespace fs {

enum PackKind { PK_CONV = 0, PK_UP = 1, PK_FOLD = 2, PK_C4 = 3 };

struct ConvBPlan {
    int WM;        // 32-pixel MFMA tiles per wave (workgroup tile = 4*WM*32 pixels)
    int BN;        // output channels per workgroup (32 or 64)
    int cout_pad;  // Cout rounded up to BN (packed filters are zero-padded to it)
    int c4;        // image layer (Cin == 3, fp32 input, 4-channel bf16 pixels in LDS)
    int CC, PP;    // channels per staged chunk; LDS pixel pitch in elements (CC + 8)
    int TH, TW, tiles_y, tiles_x, PH, PW;
    int lds_bytes;
    int wst_off;   // byte offset of the per-tile statistics scratch [4][BN][4] in LDS
    int bs;        // > 0: instance of the streaming kernel (fs_bstream.hip) that takes this launch; 0: the kernels of fs_bf16.hip
};

struct ConvBArgs {
    const void* x;            // bf16 [N,H,W,Cin], or fp32 [N,H,W,3] when x_f32
    const unsigned short* w;  // packed bf16 filter (see pack_bf16_kernel)
    void* y;                  // bf16 [N,Ho,Wo,Cout] ([N,2Ho,2Wo,Cout/4] when shuffle), or fp32 when y_f32
    int N, H, W, Cin;
    int Ho, Wo, Cout;
    int KH, KW, stride, pad_t, pad_l, dil_x;
    int src_mode, refl;
    const float* in_a;  // producer instance norm applied on load: v = relu(x*a + b)
    const float* in_b;
    int in_nstride, in_relu;
    int shuffle;
    float* stats;  // per-tile {mean, M2, count} per channel, from the fp32 accumulators
    int x_f32, y_f32;
    ConvBPlan p;
};

int bstream_instance(const ConvBArgs& a);            // fs_bstream.hip: 0 = not eligible
void bstream_plan(const ConvBArgs& a, ConvBPlan* out);
int bstream_launch(const ConvBArgs& a, hipStream_t s);
ConvBPlan conv_bf16_plan(const ConvBArgs& a);
int conv_bf16_launch(const ConvBArgs& a, hipStream_t s);

struct BTnetLayout {
    TnetLayout geo;  // unit geometry (shapes, pads, parameter offsets) shared with the fp32 path
    ConvBPlan plan[16];
    int tiles[16];
    size_t z[16], stats[16], mean[16], rstd[16], a[16], b[16], wpk[16], wpk_elems[16];  // byte offsets
    size_t h[5], zfold;
    size_t total_bytes;
};
int tnet_layout_bf16(int N, int H, int W, BTnetLayout* L);
int tnet_forward_bf16(const BTnetLayout& L, const float* params, const float* x, float* y, void* ws, hipStream_t s);

}  // namespace fs
