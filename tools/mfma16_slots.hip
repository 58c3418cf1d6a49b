// Microbenchmark behind the wgrad2 sweep layout: cycles per v_mfma_f32_16x16x4_f32 slot (one wave per SIMD, 18 accumulators)
// when every slot also carries the operand traffic of the NEXT step -- one address add + one ds_read_b32 (what the filter-
// gradient sweep of the 9x9 layers needs: M = 18 k-blocks, N = 1 channel block), wider reads shared by several slots, or
// nothing.   hipcc --offload-arch=gfx950 -O3 tools/mfma16_slots.hip -o /tmp/mfma16_slots && /tmp/mfma16_slots
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KM = 18;
// MODE 0: MFMA only | 1: + v_add + ds_read_b32 per slot | 2: + ds_read_b32 per slot, immediate offsets (no add)
//      3: + one ds_read_b128 per 4 slots (+ add) | 4: + v_add only | 5: + ds_read2_b32 per 2 slots (+ add)
//      6: as 1 with 2-way bank conflicts | 7: as 1, lanes of a 16-group share an address (broadcast)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, int soff_step) {
    __shared__ float lds[16384];
    f32x4 acc[KM];
    for (int i = 0; i < KM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int ab[KM];
    for (int q = 0; q < KM; ++q) {
        int l = lane;
        if (MODE == 6) l = lane * 2;          // 2-way conflict
        if (MODE == 7) l = lane >> 4;         // broadcast groups
        ab[q] = ((q * 67 + l) & 4095) * 4;
    }
    const char* lds0 = reinterpret_cast<const char*>(lds);
    float a0[KM], a1[KM], b = 1.0f;
    for (int q = 0; q < KM; ++q) a0[q] = lds[(ab[q] >> 2)], a1[q] = 0.f;
    int soff = __builtin_amdgcn_readfirstlane(0);
    auto step = [&](const float (&ca)[KM], float (&na)[KM]) __attribute__((always_inline)) {
        soff = (soff + soff_step) & 8191;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < KM; ++q) {
            if (MODE == 1 || MODE == 6 || MODE == 7) na[q] = *reinterpret_cast<const float*>(lds0 + (ab[q] + soff));
            if (MODE == 2) na[q] = *reinterpret_cast<const float*>(lds0 + ab[0] + q * 260);
            if (MODE == 3 && (q & 3) == 0) {
                const float4 t = *reinterpret_cast<const float4*>(lds0 + (((ab[q] + soff) & ~15)));
                na[q] = t.x;
                if (q + 1 < KM) na[q + 1] = t.y;
                if (q + 2 < KM) na[q + 2] = t.z;
                if (q + 3 < KM) na[q + 3] = t.w;
            }
            if (MODE == 4) { int t = ab[q] + soff; asm volatile("" : "+v"(t)); na[q] = ca[q]; }
            if (MODE == 5 && (q & 1) == 0) {
                const float* p = reinterpret_cast<const float*>(lds0 + (ab[q] + soff));
                na[q] = p[0];
                if (q + 1 < KM) na[q + 1] = p[33];
            }
            if (MODE == 0) na[q] = ca[q];
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[q], b, acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 2) {
        step(a0, a1);
        step(a1, a0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < KM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + a0[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, float* out, long long* cyc) {
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 64);
        hipDeviceSynchronize();
    }
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; ++i) m += h[i];
    m /= 256;
    printf("%-58s %.1f cycles per MFMA slot\n", name, m / (iters * (double)KM));
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("MFMA 16x16x4 only", out, cyc);
    run<4>("+ v_add per slot", out, cyc);
    run<2>("+ ds_read_b32 per slot (immediate offset)", out, cyc);
    run<1>("+ v_add + ds_read_b32 per slot", out, cyc);
    run<7>("+ v_add + ds_read_b32 per slot (broadcast groups)", out, cyc);
    run<6>("+ v_add + ds_read_b32 per slot (2-way conflict)", out, cyc);
    run<5>("+ v_add + ds_read2_b32 per 2 slots", out, cyc);
    run<3>("+ v_add + ds_read_b128 per 4 slots", out, cyc);
    return 0;
}
