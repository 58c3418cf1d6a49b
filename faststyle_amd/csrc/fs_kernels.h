// Internal declarations shared by the HIP translation units of libfaststyle_hip.so.
// (The public C ABI is include/faststyle_hip.h; nothing here is exported.)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <mutex>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int fs_u32x4 __attribute__((ext_vector_type(4)));   // payload type of raw_buffer_store_b128

namespace fs {

// How a conv reads its (virtual) input image.  Padding outside the virtual image is zero.
enum SrcMode {
    SRC_PLAIN = 0,    // virtual image == tensor
    SRC_REFLECT = 1,  // virtual image = tensor mirrored by `refl` px on H and W (tf.pad REFLECT)
    SRC_DILATE2 = 2,  // virtual[2i] = tensor[i], odd positions zero (dgrad of a stride-2 conv)
    SRC_UP4 = 3,      // virtual[i] = tensor[i/4] (as-written NEAREST x4 upsample; wgrad only)
};

struct ConvPlan {
    int variant;  // conv_igemm_kernel<MT,WM,WN>: 0 <32,2,2>  1 <32,2,1>  2 <16,4,1>  3 <32,1,2>  4 <32,1,1>;  5 wino_conv_kernel;  6 wino2_conv_kernel;
                  // 7 conv_stream_kernel;  8 wino2h_conv_kernel;  9 conv_s16_kernel;  10 wino4_conv_kernel (F(4x4,3x3));  11 wino4t_conv_kernel (F(4x4,3x3), 16-tile items)
                  // 12 the split-bf16 F(4x4,3x3) pipeline of fs_wino6.hip (input transform, 36 GEMMs on the bf16 matrix cores, output transform)
    int BN;       // output channels per workgroup
    int flat;     // Cin==3: K runs over (kw,ci) contiguously per kernel row
    int CC;       // input channels staged per chunk
    int LG;       // k-extent of one tap group (CC, or roundup(KW*Cin) when flat)
    int S;        // LDS floats per patch pixel
    int TH, TW, tiles_y, tiles_x;
    int PH, PW;   // staged patch extent
    int lds_bytes;
    int ksplit;   // > 1: blockIdx.z splits the input-channel chunks; raw partials go to ConvArgs::split_ws
    int rem_full;     // wino2, remainder split: items [0, rem_full) are whole items (rem_full = a multiple of the persistent grid) ...
    int rem_ks;       // ... and each item behind them is split into rem_ks units over its input-channel chunks (0: no split)
    int flat_tiles;   // fs_wino4t.hip: 16-tile items over the sample's FLATTENED row-major tile list (tiles_y = 1, tiles_x = items per sample) instead of 16 x 16-pixel blocks
    int xcd_swizzle;  // workgroup -> (tile, channel block) map that keeps sharers of an input patch on one XCD
    int skew;  // > 0: first-round workgroups in odd wave slots start late by skew x 2048 cycles (see conv_igemm_kernel)
};

// Device-coherent accesses for the few values workgroups of ONE launch hand to each other (the statistics records of the
// fused instance-norm finalize): agent-scope relaxed atomics compile to loads / stores that bypass the non-coherent per-XCD
// L2 (sc1), so no cache-wide release/acquire (buffer_wbl2 / buffer_inv: measured ~35 us per launch with __threadfence()) is
// needed -- the producer waits for its own stores (s_waitcnt vmcnt(0)) before it bumps the counter.
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_COHERENT_STORE(ptr, v) __hip_atomic_store((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define FS_COHERENT_LOAD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define FS_COUNTER_BUMP(ptr) __hip_atomic_fetch_add((ptr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define FS_DRAIN_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define FS_COHERENT_STORE(ptr, v) (*(ptr) = (v))
#define FS_COHERENT_LOAD(ptr) (*(ptr))
#define FS_COUNTER_BUMP(ptr) atomicAdd((ptr), 1u)
#define FS_DRAIN_STORES() __threadfence()
#endif

// Instance-norm finalize fused into the kernel that produces the statistics records (fs_wino2 / fs_wino2h / fs_cstream): the
// LAST workgroup of the launch to finish merges the per-tile {mean, M2, count} records of every (sample, channel) and
// writes mean, rstd, a = gamma*rstd, b = beta - mean*a (im_transf_net.py:238-245) -- what in_finalize_kernel does in a
// launch of its own (~4.5 us + two launch gaps; 16 of them are ~10 % of a 720p frame).  counter == nullptr: off.
struct FinArgs {
    unsigned* counter;   // zero before the launch; the kernel leaves it zero
    const float* gamma;  // [C]
    const float* beta;
    float* mean;         // [N][C] each
    float* rstd;
    float* a;
    float* b;
    int T;               // records per sample (tiles of the launch)
    int C;               // real channels
    int groups;          // virtual channel groups per real channel (4 for the phase-collapsed resize-conv), Cv = C * groups
    float eps;
};

struct ConvArgs {
    const float* x;  // [N,H,W,Cin]
    const float* w;  // [KH*KW, Cin, Cout] (HWIO); + n*w_nstride for per-sample weights
    const float* w_wino;  // optional: the same filter Winograd-transformed, [16][Cin][Cout] (fs::wt_wino); enables variant 5
    const float* w_wino2; // optional: ... in the K-contiguous order [16][Cin/8][Cout][8] (fs::wt_wino2); enables variant 6
    const float* w_wino4; // optional: the filter transformed for F(4x4,3x3), [36][Cin/4][Cout/64][2][4][16][2] (fs::wt_wino4); enables variant 10
    const float* w_wino4t; // optional: the filter transformed for F(4x4,3x3) in the register layout of fs_wino4t.hip, [Cin/8][Cout/16][18][64][4] (fs::wt_wino4t); enables variant 11
    const float* w_wino4u; // optional, with w_wino4t: the same filter in the layout of the 128-channel item form of fs_wino4t.hip, [Cin/8][Cout/32][36][64][4] (fs::wt_wino4u)
    const unsigned short* w_wino6; // optional: the F(4x4,3x3) filter as three bf16 pieces, [36][Cin/32][3][Cout][32] (fs::wt_wino6), with w6_ws: enables variant 12 (fs_wino6.hip)
    float* w6_ws;          // scratch of the split-bf16 pipeline: V [36][tiles][Cin] + M [36][tiles][Cout] of one tile chunk (w6_ws_floats capacity; wino6_ws_floats() = one pass)
    size_t w6_ws_floats;
    hipStream_t w6_side;   // optional second stream + three events (fs_wino6.hip, round 6): the launch runs as two tile chunks software-pipelined over the two streams --
    hipEvent_t* w6_ev;     // input transform of chunk b beside the GEMM of chunk a, output transform of chunk a beside the GEMM of chunk b; joined before it returns
    float* y;        // [N,Ho,Wo,Cout], or [N,2Ho,2Wo,Cout/4] when shuffle
    int N, H, W, Cin;
    int Ho, Wo, Cout;
    int KH, KW, stride, pad_t, pad_l;
    int dil_x;  // horizontal tap spacing (0/1: dense; 5 for the kw-folded 9x9x16->3 layer)
    int src_mode, refl;
    const float* in_a;  // optional on-load affine: v = x*in_a[n*in_nstride+c] + in_b[...]
    const float* in_b;
    int in_nstride;     // 0 -> per-channel only
    int in_relu;
    const float* bias;  // optional [Cout]
    int out_relu;
    int shuffle;           // 2x2 pixel-shuffle store (phase-collapsed resize-conv, phase-decomposed stride-2 dgrad)
    int shuf_H, shuf_W;    // extent of the shuffled output when it is not exactly 2Ho x 2Wo (odd sizes are clipped); 0: exact
    float* stats;          // optional per-tile {mean, M2, count} partials [N*tiles][Cout][3]
    const float* add_src;  // optional residual [N,Ho-2*add_pad,Wo-2*add_pad,Cout] added in the interior
    int add_pad;
    const float* mask_src;  // optional [N,Ho,Wo,Cout]: after the add, v = mask_src > 0 ? v : 0 (ReLU gradient of the consumer)
    const float* route_src; // optional [N,ceil(Ho/2),ceil(Wo/2),Cout], with mask_src (direct kernel, no add_src / shuffle / split-K): the gradient of a 2x2/2 SAME max-pool over mask_src, routed to the FIRST maximum of every
                            // window, is added before the mask -- v = mask > 0 ? v + (mask is its window's arg-max ? route_src : 0) : 0
                            // (the backward of vgg16.py's pool + ReLU behind a tapped layer, fused into the Gram-gradient conv)
    float* pool_out;        // optional [N,Ho/2,Wo/2,Cout] (Winograd kernels, even Ho/Wo, no split-K): max over every 2x2 output tile =
                            // tf.nn.max_pool 2x2/2 of the stored result (vgg16.py:68,104,154) straight from the epilogue's registers
    int y_keep_n;           // with pool_out (fs_wino4t.hip): > 0 -- samples n >= y_keep_n get the pooled tensor only, their full-resolution y is NOT stored (the content
                            // half of the perceptual-loss batch at conv1_2 / conv2_2: nothing reads it again)
    long long w_nstride;
    int prof_tag;            // 1: launched by the transform net -- selects the PROFILER ROW only (no effect on the plan or the computation)
    int tnet_plan;           // 1: plan as the transform net's launches are planned -- 16-tile items in fs_wino4t.hip whatever the grid (the layout allocated
                             // statistics records and workspace for that tiling), the remainder split in fs_wino2.hip.  Every launch site of fs_tnet.hip sets it
                             // together with prof_tag; the public fs_conv2d_fwd leaves both 0 (round 4 keyed the plan on prof_tag: a launch site that forgot the
                             // "profiler" tag planned another tile count than the layout had allocated for)
    int res_x6;              // 1: a forward residual conv of the transform net that may take the split-bf16 direct kernel (conv_r64x_kernel, fs_cstream.hip) instead of
                             // the fp32 Winograd one -- set by the layout (Unit::x6) at plan AND launch time
    FinArgs fin;             // fused instance-norm finalize (with stats; persistent kernels only)
    float* rem_ws;           // optional scratch for the remainder split of fs_wino2 (rem_ws_floats capacity): the items of the last,
    size_t rem_ws_floats;    // partial round of a persistent launch are split over the reduction dimension across ALL workgroups
    int half_items;          // 1: with w_wino2, prefer the half-item Winograd kernel (fs_wino2h.hip: grids too small for 64-tile items)
    float* split_ws;         // optional scratch for split-K partials (split_ws_floats capacity); enables ksplit plans
    size_t split_ws_floats;
    // optional (fs_wino4t.hip, raw / add_src epilogues -- the residual input gradients of the transform net): y is the gradient g wrt the OUTPUT
    // of an instance-norm unit whose raw conv output is inb_z [N,Ho,Wo,Cout]; the epilogue also leaves that unit's instance-norm-backward
    // partial sums inb_rec [N][tiles][Cout][2] = {sum g', sum g' * xhat} per item (g' = g where relu(a z + b) > 0 when inb_relu, else g;
    // xhat = (z - mean) * rstd; im_transf_net.py:218-247 adjoint) -- what in_bwd_partial4_kernel computes in a pass of its own over g and z
    const float* inb_z;
    const float* inb_mean;   // [N][Cout] each
    const float* inb_rstd;
    const float* inb_a;
    const float* inb_b;
    int inb_relu;
    float* inb_rec;
    ConvPlan p;
};

struct WgradPlan {
    int TH, TW, tiles_y, tiles_x, PH, PW, S;
    int K;        // KH*KW*Cin
    int KB;       // k-blocks of 32 (ceil)
    int NB;       // cout blocks of 32 (ceil)
    int KWV, NWV; // a wave owns KWV k-blocks x NWV co-blocks of 32x32 MFMA tiles
    int n_wg;     // workgroups per sample-group (each strides over the tile list)
    int waves_k;  // waves tiling the k dimension (1, 2, 4); the other 4/waves_k split the pixel rows
    int n_slabs;  // partial slabs per sample-group = n_wg * 4/waves_k
    int lds_bytes;
};

struct WgradArgs {
    const float* x;   // conv input  [N,H,W,Cin]  (virtual via src_mode)
    const float* dy;  // conv output gradient [N,Ho,Wo,Cout]
    float* slabs;     // partial sums [groups][n_wg][K][Cout]; groups = N when per_sample else 1
    int N, H, W, Cin, Ho, Wo, Cout;
    int KH, KW, stride, pad_t, pad_l;
    int dil_x;
    int src_mode, refl;
    const float* in_a;
    const float* in_b;
    int in_nstride, in_relu;
    const float* dy_a;  // optional affine+ReLU applied to dy on load (deconv filter gradient: dy is an activation)
    const float* dy_b;
    int dy_nstride, dy_relu;
    int per_sample;   // 1: Gram-style, one result per n; 0: summed over the batch
    int dy_unshuffle; // dy is [N,2Ho,2Wo,Cout/4]; read it as the 2x2 pixel-unshuffled [N,Ho,Wo,Cout]
    int same_xy;      // set by wgrad_launch: Gram of <= 128 channels (x == dy, 1x1): the B operand comes from the staged x tile
    WgradPlan p;
};

// ---- second-generation filter gradients (fs_wgrad2.hip): persistent workgroups, 16x16x4 MFMA, batched problems
constexpr int kW2MaxProb = 10;
struct Wg2Plan {
    int TH, TW, PH, PW;   // pixel tile and its input patch
    int S, DP;            // LDS pitches (floats) of a patch pixel / a dY pixel
    int K, KB, NB;        // KH*KW*Cin, its 16-row blocks, 16-channel blocks of Cout
    int KM, KN;           // blocks per wave (kernel instantiation)
    int waves_k;          // waves tiling the k dimension; the other 4/waves_k split the tile's pixel rows
    int patch_floats, stage_floats, lds_bytes;
    int xn, dn;           // 16-byte loads per thread and tile: patch / dY
    int combine;          // the pixel-row groups of a workgroup are summed through LDS before the store: ONE slab per workgroup
};
struct Wg2Prob {
    const float* x;       // [N,H,W,Cin] (virtual via src_mode)
    const float* dy;      // [N,Ho,Wo,Cout]
    const float* in_a;    // optional [N,Cin] on-load affine of x (+ ReLU): the producer's instance norm
    const float* in_b;
    const float* dy_a;    // optional on-load affine of dy (conv2d_transpose units)
    const float* dy_b;
    float* slabs;         // [wg_count * (combine ? 1 : 4/waves_k)][K][Cout]
    size_t slab_off;      // offset of `slabs` inside the launch's scratch (floats)
    int N, H, W, Ho, Wo, tiles_y, tiles_x;
    int wg_begin, wg_count;
};
struct Wg2Args {
    int Cin, Cout, KH, KW, stride, pad_t, pad_l, dil_x, src_mode, refl;
    int in_nstride, in_relu, dy_nstride, dy_relu, dy_unshuffle;
    int nprob, n_wg;
    int debug;   // FS_WGRAD2_DEBUG (timing experiments)
    Wg2Plan p;
    Wg2Prob prob[kW2MaxProb];
};
struct Wg2Reduce {
    struct Job {
        const float* slabs;
        float* out;
        size_t count;
        int n_slabs;
    } job[kW2MaxProb];
    float scale;
    int n;
};
// ---- Winograd filter gradients F(3x3, 2x2) of the 3x3 stride-1 VALID 64 -> 64 convs (fs_wgw.hip): batched problems, one slab per workgroup
struct WgwProb {
    const float* x;       // [N,H,W,64]
    const float* dy;      // [N,H-2,W-2,64]
    const float* in_a;    // optional [N,64] on-load affine of x (+ ReLU): the producer's instance norm
    const float* in_b;
    int N, H, W, Ho, Wo, Ty, Tx;
    int sps, steps;       // 16-tile steps per sample / of the problem
    int wg_begin, wg_count;
    size_t slab_off;      // offset of the problem's slabs inside the launch's scratch (floats)
};
struct WgwArgs {
    int nprob, n_wg, in_nstride, in_relu;
    float* slabs;
    WgwProb prob[kW2MaxProb];
};
struct WgwReduce {
    struct Job {
        const float* slabs;
        float* out;
        int n_slabs;
    } job[kW2MaxProb];
    float scale;
    int n;
};
bool wgw_eligible(const WgradArgs& a);
size_t wgw_plan(const WgradArgs* probs, int n, WgwArgs* out);   // returns the slab scratch in floats (0: not eligible)
int wgw_run(const WgwArgs& planned, float* slabs, float* const* dw, float scale, hipStream_t s);
bool wgrad2_eligible(const WgradArgs& a);
size_t wgrad2_plan(const WgradArgs* probs, int n, Wg2Args* out);   // returns the slab scratch in floats (0: not eligible)
int wgrad2_run(const Wg2Args& planned, float* slabs, float* const* dw, float scale, hipStream_t s, Wg2Reduce* defer = nullptr);
int wgrad2_reduce(const Wg2Reduce& r, hipStream_t s);

ConvPlan conv_plan(const ConvArgs& a);
int conv_launch(const ConvArgs& a, hipStream_t s);
WgradPlan wgrad_plan(const WgradArgs& a);
int wgrad_launch(const WgradArgs& a, hipStream_t s);

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// f(integral_constant<int, I0>) ... f(integral_constant<int, I1 - 1>) as straight-line code.  (#pragma unroll is a request the
// loop unroller declines beyond its size budget -- a 36-step sweep with its staging slices inlined is -- and a rolled loop
// over an accumulator array puts the accumulators in scratch memory.)
template <int I0, int I1, class F>
__device__ __forceinline__ void fs_static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        fs_static_for<I0 + 1, I1>(f);
    }
}

// Address of a kernel's (single, by-value) argument struct in the kernel-argument segment, for kernels that index a
// table inside it at run time.  (Host passes -- the launch stub, the CPU emulator of tests/ -- take the parameter's own
// address.)
// One element of an MFMA accumulator moved to an ordinary vector register at THIS point of the program (the "a" constraint
// keeps the operand in the accumulator file, `volatile` keeps the read where it is written).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float fs_acc_read(float v) {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(v));
    return x;
}
#define FS_ACC_READ(e) fs_acc_read(e)
#else
#define FS_ACC_READ(e) (e)
#endif
// The column stage of the Winograd input transform on packed fp32.  With t01 = {t0, t1}, t23 = {t2, t3} (one row of B^T d):
//   cols01 = {t0 - t2, t1 + t2},  cols23 = {t2 - t1, t1 - t3}
// Each is ONE v_pk_add_f32: op_sel / op_sel_hi pick the low or high dword of a source for the low / high result, neg_lo /
// neg_hi negate it -- no register moves (left to the compiler the shuffles become v_mov + v_xor and nothing is saved).
// -DFS_NO_PK_ASM builds the plain-C form (what the emulator and the host pass see).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fs_wino_cols01(f32x2 t01, f32x2 t23) {   // {t0 - t2, t1 + t2}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FS_NO_PK_ASM)
    f32x2 o;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(o) : "v"(t01), "v"(t23));
    return o;
#else
    f32x2 o;
    o.x = t01.x - t23.x;
    o.y = t01.y + t23.x;
    return o;
#endif
}
__device__ __forceinline__ f32x2 fs_wino_cols23(f32x2 t01, f32x2 t23) {   // {t2 - t1, t1 - t3}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FS_NO_PK_ASM)
    f32x2 o;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(o) : "v"(t01), "v"(t23));
    return o;
#else
    f32x2 o;
    o.x = t23.x - t01.y;
    o.y = t01.y - t23.y;
    return o;
#endif
}
// a + b / a - b on two packed floats as ONE instruction.  (Written as vector arithmetic the backend un-packs v_pk_add_f32
// next to matrix instructions, betting on co-issue; measured on gfx950 -- tools/mfma_overlap.hip -- every vector instruction
// beside the fp32 MFMA stream costs its issue time, so fewer instructions is what counts.)
__device__ __forceinline__ f32x2 fs_pk_add(f32x2 a, f32x2 b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FS_NO_PK_ASM)
    f32x2 o;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
#else
    return a + b;
#endif
}
__device__ __forceinline__ f32x2 fs_pk_sub(f32x2 a, f32x2 b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FS_NO_PK_ASM)
    f32x2 o;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(a), "v"(b));
    return o;
#else
    return a - b;
#endif
}
__device__ __forceinline__ f32x2 fs_pk_fma(f32x2 a, f32x2 b, f32x2 c) {   // a * b + c on two packed floats, one instruction
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FS_NO_PK_ASM)
    f32x2 o;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
#else
    f32x2 o;
    o.x = fmaf(a.x, b.x, c.x);
    o.y = fmaf(a.y, b.y, c.y);
    return o;
#endif
}
// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4): `gsrc` is the lane's own source, `lds_wave`
// the wave-uniform destination -- lane l lands at lds_wave + 16*l bytes.  Completion is covered by vmcnt / the next barrier.
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_GLOBAL_LOAD_LDS_B128(gsrc, lds_wave)                                                                        \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),                            \
                                     (__attribute__((address_space(3))) void*)(lds_wave), 16, 0, 0)
#elif defined(FS_EMULATOR)   /* tests/emu/hip/hip_runtime.h: the CPU emulator of the test suite */
#define FS_GLOBAL_LOAD_LDS_B128(gsrc, lds_wave) fs_emu_global_load_lds_b128((gsrc), (lds_wave))
#else                        /* host pass of hipcc: kernel bodies are parsed, never run */
#define FS_GLOBAL_LOAD_LDS_B128(gsrc, lds_wave) ((void)0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_KERNARG_PTR(T, param) (reinterpret_cast<const T*>(__builtin_amdgcn_kernarg_segment_ptr()))
#else
#define FS_KERNARG_PTR(T, param) (&(param))
#endif

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is a workgroup-scope release/acquire over every address
// space: the compiler puts `s_waitcnt vmcnt(0) lgkmcnt(0)` in front of s_barrier, i.e. every global load still in flight is
// waited for -- fatal for a software pipeline that keeps the next step's global loads travelling ACROSS the barrier.  Where
// the threads of a workgroup exchange data through LDS only, this waits for the LDS operations alone.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FS_LDS_BARRIER_OFF)   /* -DFS_LDS_BARRIER_OFF: A/B build with plain __syncthreads() */
#define FS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define FS_LDS_BARRIER() __syncthreads()
#endif

// LDS accesses beside matrix instructions.  A vector-ALU instruction between two fp32 v_mfma costs ~20 cycles there (same fp32 units), an LDS
// instruction next to nothing -- so address arithmetic must not reach the vector ALU:
//   FS_LDS_LOAD1 / FS_LDS_STORE1: a 4-byte access the compiler may NOT pair with a neighbour (ds_read2_b32 / ds_write2_b32 take 8-bit offsets in
//     units of 4 bytes: a pair further than 1020 bytes from the base register gets a v_add for its own base; the unpaired access has a 16-bit
//     byte offset).  Measured on fs_wino4t.hip: 18 such adds per step removed = -5 % of the step.
//   FS_OPAQUE(x): the value behind an opaque copy -- the compiler cannot fold a second base register back into "first base + constant"
//     (offsets past 65535 bytes would again cost a v_add each).
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_LDS_LOAD1(ptr) (*(const volatile __attribute__((address_space(3))) float*)(ptr))
#define FS_LDS_STORE1(ptr, v) (*(volatile __attribute__((address_space(3))) float*)(ptr) = (v))
#define FS_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define FS_LDS_LOAD1(ptr) (*(ptr))
#define FS_LDS_STORE1(ptr, v) (*(ptr) = (v))
#define FS_OPAQUE(x) ((void)0)
#endif

// Wait for every vector-memory operation of this wave (s_waitcnt vmcnt(0); gfx9 encoding, the other counters left at their
// maxima).  Placed at the END of a pipeline prologue: the compiler inserts waits statically, so a load that is still pending
// on ONE path into a loop header (the prologue's) costs a vmcnt(0) in EVERY iteration if its destination register is
// recycled inside the loop -- draining once before the loop removes that path's pending set.
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)
#else
#define FS_WAIT_VMEM() ((void)0)
#endif
// The same wait, pinned: the builtin alone is no barrier for the instruction scheduler (round 6, fs_wino6.hip: it sank a prologue's loads BELOW the
// wait), and an inline-assembly s_waitcnt is invisible to the wait-count insertion pass, which then keeps its pending set.  The builtin between two
// empty assembly statements with a memory clobber is both: no load or store crosses it, and the pass clears its pending vector-memory set there.
// Use: at a point of a loop body that EVERY iteration passes after its last staged load has been consumed (the end of the body).  Loads issued and
// waited for inside conditional blocks (`if (more) issue(..)`, per-element `if (i >= n) break`) otherwise stay "possibly pending" at the loop header,
// and the pass protects the first re-use of one of their destination registers -- often an address computation in the NEXT tile's issue phase --
// with s_waitcnt vmcnt(0..1): the tile loads just issued are drained before the matrix instructions they were meant to travel beside
// (fs_wgrad2.hip before this macro: the whole global -> register latency of a tile exposed in front of every sweep).
// A narrower tool for the same disease where a full drain would also wait for stores in flight (a persistent kernel whose epilogue stores are meant to
// drain during the next sweep): an unconditional READ of a register that a conditional block loads (`if (has_ab) va = load(..)`) and another
// conditional block consumes.  The pass puts the exact wait for that load in front of the read and knows the register clean from there on; without it
// the register is "possibly pending" at the loop header and the next issue phase's address arithmetic -- the allocator likes to build the next address
// in the load's own destination -- is guarded with vmcnt(0..1).  Place it where the consumer's waits are (after the commit).
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_TOUCH_F4(v) asm volatile("" ::"v"((v).x), "v"((v).y), "v"((v).z), "v"((v).w))
#else
#define FS_TOUCH_F4(v) ((void)0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_WAIT_VMEM_FENCED()            \
    do {                                 \
        asm volatile("" ::: "memory");   \
        FS_WAIT_VMEM();                  \
        asm volatile("" ::: "memory");   \
    } while (0)
#else
#define FS_WAIT_VMEM_FENCED() ((void)0)
#endif

// The tail of a persistent kernel with ConvArgs::fin set, called by EVERY thread of EVERY workgroup after its last
// statistics record is written.  `scratch`: >= 2 KB of LDS nobody else uses any more.  One workgroup -- the last to arrive at
// the counter -- does the merge: thread = (pair = (sample, channel), lane of LP) with LP = 256 / pairs lanes per pair
// (1, 2, 4, ... 64), two passes over the records in float64 (count and mean first, then M2 around that mean: the same value
// as Chan's pairwise update, without its division per record), fixed order -> deterministic.  The counter only decides WHO
// merges; it is reset for the next launch.
__device__ inline void fs_fused_in_finalize(const FinArgs& f, const float* stats, int N, float* scratch) {
    if (!f.counter) return;
    const int tid = threadIdx.x;
    unsigned* flag = reinterpret_cast<unsigned*>(scratch);
    FS_DRAIN_STORES();   // this thread's records (FS_COHERENT_STORE: written through to the coherence point) have landed ...
    __syncthreads();     // ... and so have those of the whole workgroup, before it is counted
    if (tid == 0) *flag = FS_COUNTER_BUMP(f.counter) == gridDim.x * gridDim.y * gridDim.z - 1u ? 1u : 0u;
    __syncthreads();
    if (!*flag) return;
    double* red = reinterpret_cast<double*>(scratch + 16);   // [256] doubles, 16-byte aligned
    const int C = f.C, G = f.groups, Cv = C * G, R = f.T * G;
    const int pairs = N * C;
    int LP = 1;
    while (LP < 64 && LP * 2 * pairs <= 256) LP *= 2;
    const int PPB = 256 / LP;   // pairs per pass
    for (int p0 = 0; p0 < pairs; p0 += PPB) {
        const int pl = tid / LP, ln = tid - pl * LP;
        const int pr = p0 + pl;
        const bool live = pr < pairs;
        const int n = live ? pr / C : 0, c = live ? pr - n * C : 0;
        const float* base = stats + ((size_t)n * f.T * Cv + c) * 3;
        auto rec = [&](int i) {   // record i of the pair: tile i / G, group i % G
            const int t = i / G, q = i - t * G;
            return base + ((size_t)t * Cv + q * C) * 3;
        };
        double cnt = 0, sm = 0;
        if (live)
            for (int i = ln; i < R; i += LP) {
                const float* st = rec(i);
                const double cb = FS_COHERENT_LOAD(st + 2);
                cnt += cb;
                sm += cb * (double)FS_COHERENT_LOAD(st);
            }
        __syncthreads();
        red[tid] = cnt;
        __syncthreads();
        double tc = 0;
        for (int k = 0; k < LP; ++k) tc += red[pl * LP + k];
        __syncthreads();
        red[tid] = sm;
        __syncthreads();
        double ts = 0;
        for (int k = 0; k < LP; ++k) ts += red[pl * LP + k];
        const double mu = tc > 0 ? ts / tc : 0.0;
        double m2 = 0;
        if (live)
            for (int i = ln; i < R; i += LP) {
                const float* st = rec(i);
                const double cb = FS_COHERENT_LOAD(st + 2), d = (double)FS_COHERENT_LOAD(st) - mu;
                m2 += (double)FS_COHERENT_LOAD(st + 1) + cb * d * d;
            }
        __syncthreads();
        red[tid] = m2;
        __syncthreads();
        if (live && ln == 0) {
            double tm = 0;
            for (int k = 0; k < LP; ++k) tm += red[pl * LP + k];
            const float var = (float)(tm / tc);
            const float r = 1.0f / sqrtf(var + f.eps);
            const float fm = (float)mu;
            const float av = f.gamma[c] * r;
            f.mean[pr] = fm;
            f.rstd[pr] = r;
            f.a[pr] = av;
            f.b[pr] = f.beta[c] - fm * av;
        }
    }
    if (tid == 0) FS_COHERENT_STORE(f.counter, 0u);
}
// More than 64 KiB of dynamic LDS (gfx950: 160 KiB per CU) needs hipFuncAttributeMaxDynamicSharedMemorySize, once per
// kernel AND device (one static BigLds per kernel instantiation; thread-safe).
struct BigLds {
    std::once_flag once[16];
    void ensure(const void* fn) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::call_once(once[dev & 15], [fn] { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    }
};

}  // namespace fs

namespace fs {
// ---- fs_elem.hip ----
constexpr int kFinalizeSplit = 64;  // tile ranges of the statistics pre-reduction (scratch: N*64*Cv*3 floats)
int in_finalize(const float* stats, int N, int T, int C, int groups, const float* gamma, const float* beta, float eps,
                float* mean, float* rstd, float* a, float* b, hipStream_t s, float* scratch = nullptr);
int apply_res(const float* z, const float* a, const float* b, const float* skip, const float* sa, const float* sb,
              int skip_relu, float* out, int N, int H, int W, int C, hipStream_t s);
int apply_tanh(const float* z, const float* a, const float* b, float* y, int N, int HW, int C, hipStream_t s);
int apply_affine(const float* z, const float* a, const float* b, float* y, int N, int HW, int C, int relu, hipStream_t s);
int zero_fill(float* p, size_t n, hipStream_t s);
int reduce_slabs(const float* slabs, int groups, int n_wg, size_t count, float scale, float* out, hipStream_t s);
int zero_words(void* p, int n, hipStream_t s);   // n <= 64 32-bit words = 0 -- a kernel instead of hipMemsetAsync (memset nodes misbehave in single-stream graph replays: fs_perceptual_loss)
// output-pixel tile (TH x TW <= max_px) with the best fill / halo trade-off for an Ho x Wo image (fs_conv.hip)
// Winograd F(2x2,3x3) path (fs_wino.hip)
int wt_wino(const float* w, float* U, int Cin, int Cout, hipStream_t s);
struct WinoBatch {  // several filters of one shape in one launch (the 10 residual convs of the transform net; forward + input-gradient filters: 20)
    const float* w[24];
    float* U[24];
    int n;
};
int wt_wino_batch(const WinoBatch& b, int Cin, int Cout, hipStream_t s);
int wt_wino2(const float* w, float* U, int Cin, int Cout, hipStream_t s);                      // fs_wino2.hip
int wt_wino2_batch(const WinoBatch& b, int Cin, int Cout, hipStream_t s);
bool conv_route_ok(const ConvArgs& a);   // a.p filled: can the launch take a.route_src?
bool wino2_eligible(const ConvArgs& a);
void wino2_plan(const ConvArgs& a, ConvPlan* out);
int wino2_launch(const ConvArgs& a, hipStream_t s);
bool wino2h_eligible(const ConvArgs& a);                                                       // fs_wino2h.hip: plan variant 8
long wino2h_items(const ConvArgs& a);
void wino2h_plan(const ConvArgs& a, ConvPlan* out);
int wino2h_launch(const ConvArgs& a, hipStream_t s);
// second-generation Gram matrices (fs_gram.hip): G[n] = scale * F[n]^T F[n], F [N][HW][C], C = 64 or a multiple of 128
bool gram2_eligible(int N, int HW, int C);
size_t gram2_slab_floats(int N, int HW, int C);
int gram2_launch(const float* F, float* G, float* slabs, int N, int HW, int C, float scale, hipStream_t s);
// ... and the gradient through them: dF[n] = F[n] S[n] (+ add[n]), S [N][C][C], C = 64, 128 or 256
bool gram_bwd2_eligible(int N, int HW, int C);
bool gram_bwd2_route_eligible(int N, int H, int W, int C);
int gram_bwd2_route_grid(int N, int H, int W, int C);
int gram_bwd2_launch(const float* F, const float* S, const float* add, float* dF, int N, int HW, int C, hipStream_t s, const float* above = nullptr,
                     int W = 0, const float* content = nullptr, float cscale = 0.f, float* cpartial = nullptr);
int gram_symmetrize(const float* dG, float* S, int N, int C, float scale, hipStream_t s);   // S[n] = scale * (dG[n] + dG[n]^T)
// streaming conv of the narrow full-resolution layers (fs_cstream.hip): plan variant 7
bool cstream_eligible(const ConvArgs& a);
void cstream_plan(const ConvArgs& a, ConvPlan* out);
int cstream_launch(const ConvArgs& a, hipStream_t s);
bool s16_eligible(const ConvArgs& a);                                                          // fs_s16.hip: plan variant 9 (16 output channels: 9x9 image layer, kw-folded output layer)
void s16_plan(const ConvArgs& a, ConvPlan* out);
int s16_launch(const ConvArgs& a, hipStream_t s);
bool conv3x3_to3_eligible(const ConvArgs& a);                                                 // fs_c3.hip
int conv3x3_to3_launch(const ConvArgs& a, hipStream_t s);
int wt_wino4(const float* w, float* U, int Cin, int Cout, hipStream_t s);                      // fs_wino4.hip: Winograd F(4x4,3x3), plan variant 10
bool wino4_eligible(const ConvArgs& a);
void wino4_plan(const ConvArgs& a, ConvPlan* out);
int wino4_launch(const ConvArgs& a, hipStream_t s);
int wt_wino4t(const float* w, float* U, int Cin, int Cout, hipStream_t s);                     // fs_wino4t.hip: F(4x4,3x3) with 16-tile items (transform-net residual convs), plan variant 11
int wt_wino4t_batch(const WinoBatch& b, int Cin, int Cout, hipStream_t s);
int wt_wino4u(const float* w, float* U, int Cin, int Cout, hipStream_t s);                      // ... for its 128-channel item form (ConvArgs::w_wino4u)
bool wino4t_eligible(const ConvArgs& a);
// fs_wino6.hip: F(4x4,3x3) with the Winograd-domain products as six exact bf16-piece products (round 6), plan variant 12
int wt_wino6(const float* w, unsigned short* U, int Cin, int Cout, hipStream_t s);
size_t wino6_filter_floats(int Cin, int Cout);
size_t wino6_ws_floats(int N, int Ho, int Wo, int Cin, int Cout);
bool wino6_eligible(const ConvArgs& a);
void wino6_plan(const ConvArgs& a, ConvPlan* out);
int wino6_launch(const ConvArgs& a, hipStream_t s);
long wino4t_items(const ConvArgs& a);
void wino4t_plan(const ConvArgs& a, ConvPlan* out);
int wino4t_launch(const ConvArgs& a, hipStream_t s);
bool wino_eligible(const ConvArgs& a);
void wino_plan(const ConvArgs& a, ConvPlan* out);
int wino_launch(const ConvArgs& a, hipStream_t s);
void plan_tile(int Ho, int Wo, int KH, int KW, int stride, int max_px, int* TH, int* TW);
// Tuning / debugging knobs come from the environment ONCE: the first lookup of a name reads and caches it,
// so no launch or planning path calls getenv() afterwards.  fs_debug_reload_env() (tests) drops the cache.  `unset` is
// returned when the variable is absent.  (fs_api.hip)
int tune_int(const char* name, int unset);
void tune_reload();
unsigned tune_epoch();   // bumped by tune_reload: cached plans made under older knob values are stale
// The Winograd generations are selected by ONE knob with ONE default, read through this accessor only (FS_WINO_V):
//   1  fs_wino.hip    F(2x2), first kernel          2  + fs_wino2.hip / fs_wino2h.hip  F(2x2), second generation (Cin <= 128, half items)
//   4  + fs_wino4.hip F(4x4), filter through LDS    5  + fs_wino4t.hip F(4x4), filter in registers (the fp32 default of every 3x3 stride-1 launch)
//   6  + fs_wino6.hip F(4x4) with split-bf16 products for the deep VGG16 layers (the default)
// A generation is "on" when the knob is >= its number; which one a launch takes is decided by conv_plan's family table (fs_conv.hip) from the
// filter layouts the caller provides, in the table's order.
struct WinoGen {
    int v;
    bool f2_second() const { return v >= 2; }
    bool f4_lds() const { return v >= 4; }
    bool f4_reg() const { return v >= 5; }
    bool split_bf16() const { return v >= 6; }
};
inline WinoGen wino_gen() { return WinoGen{tune_int("FS_WINO_V", 6)}; }
// One row per specialised conv kernel family: conv_plan takes the first row whose eligible() accepts the launch, conv_launch finds the row of the
// plan's variant again (and re-checks eligibility: a plan made for other arguments is refused with -7).  fs_conv.hip holds the table.
struct ConvFamily {
    const char* name;
    int variant;       // ConvPlan::variant of its plans
    bool winograd;     // off under FS_CONV_WINO=0
    bool (*eligible)(const ConvArgs&);
    void (*plan)(const ConvArgs&, ConvPlan*);
    int (*launch)(const ConvArgs&, hipStream_t);
};
const ConvFamily* conv_families(int* n);
// does a launch of this size qualify?  (one workgroup reads N*C*T*groups records: beyond a few 10^4 a launch of its own,
// spread over the chip, is faster)
// (GPU-UNVERIFIED when on: its cross-XCD hand-off -- relaxed agent-scope atomics + a manual s_waitcnt -- is exercised by the
// emulator tests only, which cannot model cache visibility.)
inline bool fused_finalize_ok(int N, int C, int T, int groups) {
    // OFF by default -- measured on MI355X (round 3): correct, but SLOWER than the launch it replaces.  The merge runs on ONE
    // compute unit while the other 255 idle: with device-scope fences +36 us per unit, with coherent (sc1) record stores /
    // loads and no fence still +27 us (batch 4: transform-net forward 0.60 -> 0.98 ms; 720p 908 -> 708 fps) against the
    // 4.5 us in_finalize launch.  Kept behind the knob as a recorded experiment (tests/test_path_parity.py runs it on the
    // emulator so that it does not rot).
    return tune_int("FS_FUSED_FINALIZE", 0) && (long)N * C * T * groups <= (long)tune_int("FS_FUSED_FINALIZE_MAX", 24576);
}

// thread-local message behind fs_last_error(); returns `code` (fs_api.hip)
int set_error(int code, const char* fmt, ...);
// tf.image.resize_images(method=2) of TF 1.0 on device u8 [H,W,3] -> f32 [Ho,Wo,3] (fs_io.hip)
int resize_bicubic_u8(const unsigned char* src, int H, int W, float* dst, int Ho, int Wo, hipStream_t s, int pixel_bytes = 3);
int u8_to_f32(const unsigned char* src, float* dst, size_t n, hipStream_t s);
int f32_to_u8(const float* src, unsigned char* dst, size_t npix, int swap_rb, hipStream_t s);
int in_bwd(const float* gin, const float* z, const float* mean, const float* rstd, const float* a, const float* b, int mode,
           float* dz, float* dgamma, float* dbeta, float* scratch, int N, int HW, int C, hipStream_t s);
// ... with the per-sample sums taken from records [N][T][C][2] (rec == nullptr: computed here into `scratch`), reduced in the apply kernel's
// prologue; S_out [N][C][2] feeds in_bwd_params (dgamma / dbeta of up to 16 units in one launch).  Returns 1 when the shape is not taken.
int in_bwd_rec(const float* gin, const float* z, const float* mean, const float* rstd, const float* a, const float* b, int mode,
               float* dz, const float* rec, int T, float* S_out, float* scratch, int N, int HW, int C, hipStream_t s);
struct InbParams {
    struct U {
        const float* S;   // [N][C][2]
        float* dgamma;
        float* dbeta;
        int C;
    } u[16];
    int n, N;
};
int in_bwd_params(const InbParams& p, hipStream_t s);
size_t in_bwd_scratch_floats(int N, int HW, int C);
int maxpool(const float* x, float* y, int N, int H, int W, int C, hipStream_t s);
int vgg_bwd_route(const float* out, const float* d_above, const float* d_tap, int pooled, float* d_pre, int N, int H, int W,
                  int C, hipStream_t s);
int sqdiff_loss(const float* x, const float* t, size_t t_period, size_t total, float lscale, float gscale, float* grad,
                float* loss_out, int accumulate, float* scratch, hipStream_t s);
int tv_loss(const float* x, int N, int H, int W, int C, float lscale, float gscale, float* grad, float* loss_out,
            float* scratch, hipStream_t s);
// the element passes of the two losses without their final sums (partial: 1024 floats each), and the ONE launch that turns the partial sums
// of all terms of a step into losses[4] = {total, content, style, beta * tv} (fs_perceptual_loss)
int sqdiff_partials(const float* x, const float* t, size_t t_period, size_t total, float gscale, float* grad, float* partial, int* n_partial,
                    hipStream_t s);
int tv_partials(const float* x, int N, int H, int W, int C, float gscale, float* grad, float* partial, int* n_partial, hipStream_t s);
struct LossFinish {
    static const int kMax = 10;
    struct Job {
        const float* partial;
        int n, slot;   // slot 1 content, 2 style, 3 tv
        float scale;
    } job[kMax];
    int n;
    float* losses;
};
int loss_finish(const LossFinish& f, hipStream_t s);
// Gram matrices of several layers finished by ONE launch (fs_gram.hip): per job the slab reduction of gram2_launch, the mirror image, and the
// style-loss terms of losses.py:61-64 on the way -- S = gscale * (G - Gt) (the filter of the Gram gradient) and the partial sums of (G - Gt)^2
int gram2_stream(const float* F, float* slabs, int N, int HW, int C, hipStream_t s);   // the matrix part of gram2_launch only
struct GramFinishJob {
    const float* slabs;
    const float* Gt;      // [C][C] target
    float* G;             // [N][C][C], optional (nullptr: not written)
    float* S;             // [N][C][C]
    float* partial;       // gram2_finish_partials(N, C) floats
    int HW, C;
    float scale, gscale;
};
int gram2_finish_partials(int N, int C);
int gram2_finish_batch(const GramFinishJob* jobs, int n, int N, hipStream_t s);
int axpby(const float* x, const float* y, float a, float b, float* out, size_t n, hipStream_t s);
int adam_tf(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps, hipStream_t s);
// Several filter re-layouts in ONE launch (blockIdx.y = job): the ~18 per-step re-layouts of the transform net
// are a few microseconds of work each, so as separate launches they cost more in launch gaps than in compute.
enum WtKind { WT_FLIPT = 0, WT_UPFWD = 1, WT_UPDGRAD = 2, WT_FOLD5FWD = 3, WT_S2DGRAD = 4 };
struct WtJob {
    int kind, KH, KW, Ci, Co, total;
    const float* src;
    float* dst;
};
struct WtBatch {
    int n;
    WtJob j[20];
    void add(int kind, const float* src, float* dst, int KH, int KW, int Ci, int Co) {
        WtJob& q = j[n++];
        q.kind = kind;
        q.src = src;
        q.dst = dst;
        q.KH = KH;
        q.KW = KW;
        q.Ci = Ci;
        q.Co = Co;
        q.total = (kind == WT_UPFWD || kind == WT_S2DGRAD) ? 16 * Ci * Co : (kind == WT_FOLD5FWD ? 18 * Ci * 16 : KH * KW * Ci * Co);
    }
};
int wt_batch(const WtBatch& b, hipStream_t s);
int wt_flip_transpose(const float* w, float* out, int KH, int KW, int Ci, int Co, hipStream_t s);
int wt_upconv_fwd(const float* w, float* weff, int Ci, int Co, hipStream_t s);
int wt_upconv_dgrad(const float* w, float* v, int Ci, int Co, hipStream_t s);
int wt_upconv_wgrad_fold(const float* dweff, float* dw, int Ci, int Co, hipStream_t s);
// ---- fs_fold.hip: kw-folded 9x9 -> 3-channel output layer ----
int fold5_fwd_bf16(const unsigned short* Z, float* z, float* stats, int N, int Ho, int Wo, hipStream_t s);  // bf16 Z (fs_bf16.hip)
int wt_fold5_fwd(const float* w, float* wf, int Ci, hipStream_t s);
int wt_fold5_back(const float* dwf, float* dw, int Ci, hipStream_t s);
int fold5_fwd(const float* Z, float* z, float* stats, int N, int Ho, int Wo, hipStream_t s);
int unfold5(const float* dz, float* dys, int N, int Ho, int Wo, hipStream_t s);
}  // namespace fs

namespace fs {
// Optional HIP-event profiler around the MFMA kernels (bench.py's roofline leg): per kernel SYMBOL (so that a row can be
// re-derived from a rocprofv3 --kernel-trace --stats summary: average duration x launches) it accumulates launches,
// FLOPs executed and the event-measured duration.  wino2_conv_kernel is split by caller: the VGG16 convs (bias + ReLU /
// consumer mask, up to 128 channels) and the transform net's 64-channel residual convs (instance-norm on load, statistics
// epilogue, 'full' padding + residual add in the input gradients) are different workloads on one symbol.
enum ProfFam {
    PF_IGEMM_32_2_2 = 0, PF_IGEMM_32_2_1 = 1, PF_IGEMM_16_4_1 = 2, PF_IGEMM_32_1_2 = 3, PF_IGEMM_32_1_1 = 4,
    PF_WINO = 5, PF_WINO2_VGG = 6, PF_WINO2_TNET = 7, PF_CSTREAM = 8, PF_C3 = 9, PF_WGRAD2 = 10, PF_WGRAD = 11,
    PF_GRAM_STREAM = 12, PF_GRAM_WGRAD = 13, PF_GRAM_BWD = 14, PF_GRAM_BWD_IGEMM = 15, PF_WINO2H_TNET = 16, PF_S16 = 17,
    PF_WINO4 = 18, PF_WGW = 19, PF_WINO4T_TNET = 20, PF_WINO4T_VGG = 21, PF_WINO6 = 22
};
const char* prof_family_name(int f);
struct Profiler {
    static const int kFamilies = 23;   // one per kernel symbol (ProfFam); fs_profile_family_name() names them
    struct Rec { hipEvent_t a, b; int fam; double flops; };
    Rec* recs = nullptr;
    int n = 0, cap = 0;
    static Profiler*& current();
    void begin(int fam, double flops, hipStream_t s);
    void end(hipStream_t s);
    int collect(double out[kFamilies][3]);  // {launches, flops, ms}; synchronises on the events
    void reset();
    ~Profiler();
};
}  // namespace fs
