// EXPERIMENT (round-4 review item 5), GPU part: the issue time of the Winograd-domain products as exact-fp32 matrix instructions against
// split-bf16 products, WITH the vector-ALU work of the input transform (and of the splits) beside them -- one wave per SIMD, as in fs_wino4t.hip.
//
// One "unit" = a 16 x 16 block of products over K = 32 input channels:
//   fp32  : 8 x v_mfma_f32_16x16x4_f32                                    (what the shipped kernel issues)
//   bf16x3: 3 x v_mfma_f32_16x16x32_bf16  (Uh Vh + Uh Vl + Ul Vh)         (error 1e-4 of the output: tools/bf16x3_error.py -- fails the bar)
//   bf16x6: 6 x v_mfma_f32_16x16x32_bf16  (three pieces per operand)      (error 6e-6: below the fp32 kernel's 1.2e-5)
// and F vector-ALU fillers (dependent v_fma_f32 chains, the transform's instruction kind) per unit, spread evenly behind the matrix
// instructions.  The M = 3 item form of fs_wino4t.hip has 144 fp32 matrix instructions and ~162 transform instructions per 8 channels, i.e.
// 18 units and ~36 fillers per unit at K = 32; the three-piece split of V adds ~7 instructions per value (~14 per unit).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16x3.hip -o exp/mfma_bf16x3 && gpurun -- ./exp/mfma_bf16x3
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NF>   // MODE 0 fp32 (8 instr / unit), 3 / 6: that many bf16 instructions per unit;  NF fillers per unit
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f, v0 = a, v1 = b, v2 = 0.5f;
    bf16x8 pa, pb;
    for (int i = 0; i < 8; ++i) {
        pa[i] = (__bf16)(a + i);
        pb[i] = (__bf16)(b - i);
    }
    constexpr int NM = MODE == 0 ? 8 : MODE;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // four units per trip
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (MODE == 0) acc[(u * NM + m) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[(u * NM + m) & 7], 0, 0, 0);   // (eight independent accumulators: no dependent issue)
                else acc[(u * NM + m) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[(u * NM + m) & 7], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // fillers of this slot: NF spread over the NM slots of the unit
#pragma unroll
                for (int f = (m * NF) / NM; f < ((m + 1) * NF) / NM; ++f) {
                    if (f & 1) v0 = fmaf(v0, 1.0001f, v1);
                    else v1 = fmaf(v1, 0.9999f, v2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = v0 + v1 + v2;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NF>
double run(float* out, long long* cyc) {
    const int iters = 400;
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL((k<MODE, NF>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
    }
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; ++i) m += h[i];
    return m / 256 / (iters * 4.0);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    printf("cycles per unit (16 x 16 products over K = 32), one wave per SIMD, 256 workgroups; F = vector-ALU fillers per unit\n");
    printf("%-8s %10s %10s %10s %10s %10s\n", "", "F = 0", "F = 16", "F = 36", "F = 50", "F = 72");
    printf("%-8s %10.1f %10.1f %10.1f %10.1f %10.1f\n", "fp32", run<0, 0>(out, cyc), run<0, 16>(out, cyc), run<0, 36>(out, cyc), run<0, 50>(out, cyc), run<0, 72>(out, cyc));
    printf("%-8s %10.1f %10.1f %10.1f %10.1f %10.1f\n", "bf16x3", run<3, 0>(out, cyc), run<3, 16>(out, cyc), run<3, 36>(out, cyc), run<3, 50>(out, cyc), run<3, 72>(out, cyc));
    printf("%-8s %10.1f %10.1f %10.1f %10.1f %10.1f\n", "bf16x6", run<6, 0>(out, cyc), run<6, 16>(out, cyc), run<6, 36>(out, cyc), run<6, 50>(out, cyc), run<6, 72>(out, cyc));
    return 0;
}
