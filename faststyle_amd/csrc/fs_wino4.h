// Shared pieces of the two Winograd F(4x4,3x3) kernels (fs_wino4.hip: the VGG16 convs, 32-tile items; fs_wino4t.hip: the
// transform net's residual convs, 16-tile items): matrix-instruction / lane-exchange / LDS-address macros, the one-dimensional
// transforms, the float64 filter transform.
#pragma once
#include "fs_kernels.h"

// 36 positions x 2 channel blocks x 4 registers = 288 accumulator registers, but the accumulator file holds 256 and the
// compiler's matrix-instruction form takes its C/D operand from that file only (asked for more, it funnels EVERY accumulator
// through one quad with v_accvgpr copies).  So positions 0..31 use the builtin (256 AGPRs), positions 32..35 an
// inline-assembly v_mfma with C/D in ordinary vector registers (legal on gfx90a+).  Inside a sweep no software wait states are needed (an
// accumulator is next touched 72 matrix instructions later); at the sweep / epilogue boundaries see FS_W4_MFMA_SETTLE below.
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_W4_MFMA_V(accq, av, bv) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accq) : "v"(av), "v"(bv))
#define FS_W4_MFMA_A(accq, av, bv) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(accq) : "v"(av), "v"(bv))
#else
#define FS_W4_MFMA_V(accq, av, bv) (accq) = __builtin_amdgcn_mfma_f32_16x16x4f32((av), (bv), (accq), 0, 0, 0)
#define FS_W4_MFMA_A(accq, av, bv) (accq) = __builtin_amdgcn_mfma_f32_16x16x4f32((av), (bv), (accq), 0, 0, 0)
#endif
// The hazard recogniser of the compiler does not look inside inline assembly: it inserts no wait states between an inline-assembly matrix
// instruction and a following v_accvgpr_read / vector-ALU read of its result (an 8-pass v_mfma_f32_16x16x4_f32 needs up to 11), nor between a
// vector-ALU write of an accumulator (the zeroing) and a following inline-assembly MFMA that takes it as SrcC.  In both kernels a workgroup
// barrier and tens of instructions separate the two -- but a barrier returns at once for the wave that arrives last and is no architectural
// guarantee, and the unexplained wrong lanes of the -fslp-vectorize build (faststyle_amd/build.py, tools/w4_slp_repro.py) are what such a
// hazard would look like.  FS_W4_MFMA_SETTLE(): 16 explicit wait states, placed ONCE per item in front of the epilogue's first accumulator
// read and once behind the zeroing -- nothing measurable against an item of >= 30k cycles.
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_W4_MFMA_SETTLE() asm volatile("s_nop 7\n\ts_nop 7" ::: "memory")
#else
#define FS_W4_MFMA_SETTLE() ((void)0)
#endif
// (FS_W4_MFMA_A: the same with C/D pinned to ONE accumulator-file quad.  fs_wino4t.hip's 144 accumulators leave the register
// allocator room, and with the builtin it keeps a fifth of them in vector registers between their two visits per sweep: 112
// v_accvgpr_write copies per sweep in front of the matrix instructions that need them.)
// v_permlane32_swap_b32 (gfx950): lanes 32..63 of the first register trade places with lanes 0..31 of the second -- the two halves of a wave exchange a
// register pair in ONE instruction, no LDS round trip
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_W4_SWAP(va_, vb_)                                                                                               \
    do {                                                                                                                   \
        const auto r_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(va_), __float_as_uint(vb_), false, false);        \
        (va_) = __uint_as_float(r_[0]);                                                                                    \
        (vb_) = __uint_as_float(r_[1]);                                                                                    \
    } while (0)
#elif defined(FS_EMULATOR)
#define FS_W4_SWAP(va_, vb_)                                                     \
    do {                                                                         \
        const float as_ = __shfl_xor((va_), 32), bs_ = __shfl_xor((vb_), 32);    \
        if ((threadIdx.x & 63) < 32) (vb_) = as_;                                \
        else (va_) = bs_;                                                        \
    } while (0)
#else
#define FS_W4_SWAP(va_, vb_) ((void)0)
#endif
// LDS accesses through COMPLETE byte addresses held in pinned vector registers (the lesson of fs_wgrad2.hip): with pointer
// arithmetic on the shared array the backend re-derives "array base + stage + lane part + row" in front of the accesses -- six
// vector adds per sweep for the patch rows alone, each ~12 cycles beside the matrix instructions.  One base register per
// stream (computed before the sweep's first matrix instruction), everything else an immediate offset.
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_W4_ADDR(p) ((int)(size_t)(const __attribute__((address_space(3))) char*)(p))
#define FS_W4_LDS(T, addr) (*(__attribute__((address_space(3))) T*)(size_t)(unsigned)(addr))
#define FS_W4_PIN(x) asm volatile("" : "+v"(x))
/* a 4-byte LDS store the compiler may NOT pair with a neighbour (ds_write2_b32 has 8-bit offsets: pairs at large strides need a vector add for their base) */
#define FS_W4_LDS_STORE1(addr, v) (*(volatile __attribute__((address_space(3))) float*)(size_t)(unsigned)(addr) = (v))
#else   /* emulator / host pass: addresses are byte offsets from the workgroup's LDS array */
#define FS_W4_ADDR(p) ((int)(reinterpret_cast<const char*>(p) - reinterpret_cast<const char*>(smem)))
#define FS_W4_LDS(T, addr) (*reinterpret_cast<T*>(reinterpret_cast<char*>(smem) + (addr)))
#define FS_W4_PIN(x) ((void)0)
#define FS_W4_LDS_STORE1(addr, v) (FS_W4_LDS(float, addr) = (v))
#endif
// B^T x for one 6-vector (input transform, one dimension): 12 instructions
#define FS_W4_BT(d0, d1, d2, d3, d4, d5, t0, t1, t2, t3, t4, t5) \
    do {                                                         \
        const float a_ = fmaf(-4.f, d2, d4);                     \
        const float b_ = fmaf(-4.f, d1, d3);                     \
        const float c_ = d4 - d2;                                \
        const float e_ = d3 - d1;                                \
        t0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));                  \
        t1 = a_ + b_;                                            \
        t2 = a_ - b_;                                            \
        t3 = fmaf(2.f, e_, c_);                                  \
        t4 = fmaf(-2.f, e_, c_);                                 \
        t5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));                  \
    } while (0)
// A^T m for one 6-vector (output transform, one dimension): 10 instructions
#define FS_W4_AT(m0, m1, m2, m3, m4, m5, y0, y1, y2, y3) \
    do {                                                 \
        const float p_ = m1 + m2, q_ = m1 - m2;          \
        const float r_ = m3 + m4, s_ = m3 - m4;          \
        y0 = m0 + p_ + r_;                               \
        y1 = fmaf(2.f, s_, q_);                          \
        y2 = fmaf(4.f, r_, p_);                          \
        y3 = fmaf(8.f, s_, q_) + m5;                     \
    } while (0)


namespace fs {
// (G g G^T) of one (input channel, output channel) pair in float64: out[r * 6 + q], g = w[kh][kw] at w[(kh * 3 + kw) * cc + i] (w HWIO, cc = Cin * Cout)
__device__ __forceinline__ void wino4_filter_transform(const float* __restrict__ w, size_t cc, size_t i, double (&out)[36]) {
    double g[3][3], t[6][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) g[kh][kw] = (double)w[(size_t)(kh * 3 + kw) * cc + i];
    auto row6 = [](double a, double b, double c, double (&o)[6]) {   // G [a b c]^T
        o[0] = a / 4.0;
        o[1] = -(a + b + c) / 6.0;
        o[2] = -(a - b + c) / 6.0;
        o[3] = a / 24.0 + b / 12.0 + c / 6.0;
        o[4] = a / 24.0 - b / 12.0 + c / 6.0;
        o[5] = c;
    };
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        double o[6];
        row6(g[0][kw], g[1][kw], g[2][kw], o);
#pragma unroll
        for (int r = 0; r < 6; ++r) t[r][kw] = o[r];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double o[6];
        row6(t[r][0], t[r][1], t[r][2], o);
#pragma unroll
        for (int q = 0; q < 6; ++q) out[r * 6 + q] = o[q];
    }
}
}  // namespace fs
