// Microbenchmark: how much of other instructions' issue time hides behind v_mfma_f32_32x32x2_f32 (one wave per SIMD)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int NF>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, const float4* gsrc) {
    __shared__ float lds[4096];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f, v0 = a, v1 = b, v2 = 0.5f, v3 = 0.25f;
    int s0 = blockIdx.x, s1 = 3;
    lds[threadIdx.x] = a;
    __syncthreads();
    const float* lp = lds + (threadIdx.x & 63);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (MODE == 1) { v0 = v0 * 1.0001f + v1; v1 = v1 * 0.9999f + v2; }          // 2 dependent-chain VALU per filler
                if (MODE == 2) { v2 += lp[(f * 64 + u * 4) & 1023]; }                         // ds_read + add
                if (MODE == 3) { s0 = s0 * 3 + s1; s1 ^= s0; asm volatile("" : "+s"(s0), "+s"(s1)); }   // 2 SALU
                if (MODE == 5) { lds[(threadIdx.x + f * 256 + u * 64) & 4095] = v3; }                    // ds_write_b32
                if (MODE == 6) { *reinterpret_cast<float4*>(lds + ((threadIdx.x * 4 + f * 1024 + u * 256) & 4092)) = make_float4(v0, v1, v2, v3); }   // ds_write_b128
                if (MODE == 7) { float4 q = *reinterpret_cast<const float4*>(lds + (((threadIdx.x & 63) * 4 + f * 256 + u * 64) & 4092)); v2 += q.x; }   // ds_read_b128 contiguous + add
                if (MODE == 8) { float4 q = *reinterpret_cast<const float4*>(lds + (((threadIdx.x & 63) * 8 + f * 256 + u * 64) & 4092)); v2 += q.x; }   // ds_read_b128, 32-byte lane stride (2-way conflict) + add
                if (MODE == 9) { float4 q = gsrc[(threadIdx.x + f * 256 + u * 1024) & 65535]; v2 += q.x; }                                            // global_load_dwordx4 + add (L2-resident)
                if (MODE == 4) { asm volatile("v_mov_b32 %0, %0" : "+v"(v3)); asm volatile("v_mov_b32 %0, %0" : "+v"(v2)); }  // 2 trivial VALU
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = v0 + v1 + v2 + v3 + (float)s0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
static float* gbuf;
template <int MODE, int NF>
void run(const char* name, float* out, long long* cyc) {
    const int iters = 200;
    hipLaunchKernelGGL((k<MODE, NF>), dim3(256), dim3(256), 0, 0, out, cyc, iters, (const float4*)gbuf);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE, NF>), dim3(256), dim3(256), 0, 0, out, cyc, iters, (const float4*)gbuf);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; ++i) m += h[i];
    m /= 256;
    printf("%-34s fillers/MFMA %d: %.1f cycles per MFMA\n", name, NF, m / (iters * 16.0));
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    hipMalloc(&gbuf, 65536 * 16); hipMemset(gbuf, 0, 65536 * 16);
    run<0, 0>("MFMA only", out, cyc);
    run<1, 1>("+2 dependent v_fma", out, cyc); run<1, 2>("+4 dependent v_fma", out, cyc); run<1, 4>("+8 dependent v_fma", out, cyc);
    run<4, 1>("+2 v_mov", out, cyc); run<4, 2>("+4 v_mov", out, cyc); run<4, 4>("+8 v_mov", out, cyc); run<4, 8>("+16 v_mov", out, cyc);
    run<2, 1>("+1 ds_read+add", out, cyc); run<2, 2>("+2 ds_read+add", out, cyc); run<2, 4>("+4 ds_read+add", out, cyc);
    run<3, 1>("+2 SALU", out, cyc); run<3, 4>("+8 SALU", out, cyc); run<3, 8>("+16 SALU", out, cyc);
    run<5, 1>("+1 ds_write_b32", out, cyc); run<5, 4>("+4 ds_write_b32", out, cyc);
    run<6, 1>("+1 ds_write_b128", out, cyc); run<6, 2>("+2 ds_write_b128", out, cyc);
    run<7, 1>("+1 ds_read_b128+add", out, cyc); run<7, 4>("+4 ds_read_b128+add", out, cyc);
    run<8, 1>("+1 ds_read_b128(2-way)+add", out, cyc); run<8, 4>("+4 ds_read_b128(2-way)+add", out, cyc);
    run<9, 1>("+1 global_load_x4+add", out, cyc); run<9, 2>("+2 global_load_x4+add", out, cyc);
    return 0;
}
