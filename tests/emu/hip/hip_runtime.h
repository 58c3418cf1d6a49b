// TEST INFRASTRUCTURE ONLY -- a fiber-based CPU emulator of the small slice of the HIP
// programming model the faststyle kernels use, so that the *unmodified* kernel sources in
// faststyle_amd/csrc can be compiled with a host compiler (`clang++ -I tests/emu`) and
// their index/tile/fragment logic exercised by `pytest -m "not gpu"` in a container that
// has no GPU.  It is never part of the product: the shipped library is built by hipcc for
// gfx950 against the real <hip/hip_runtime.h>, and faststyle_amd/_lib.py refuses to load
// anything else.
//
// Model: one workgroup = blockDim.x cooperative fibers (ucontext) on one OS thread; a
// fiber runs until it reaches __syncthreads() or a wave-level operation (MFMA, shuffle),
// then yields.  Fibers are resumed in lane order, so a missing barrier shows up
// deterministically as a stale read.  Workgroups of a launch are spread over OS threads.
// MFMA lane layouts follow /opt/skills/guides/cdna_hip_programming.md §3:
//   32x32x2 f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31,
//                row=(r&3)+8*(r>>2)+4*(l>>5);
//   16x16x4 f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15, row=4*(l>>4)+r.
// Results are k-ordered fmaf chains, bit-identical to the hardware instruction.
#pragma once
#define FS_EMULATOR 1
#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local   /* one workgroup at a time per OS thread */

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emulator"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
enum { hipMemcpyHostToDevice = 1 };
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }

typedef void* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipErrorInvalidValue; /* emulator: single in-order stream */ }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

namespace fsemu {

struct Idx { unsigned x, y, z; };
enum State { RUN = 0, WAIT_WAVE, WAIT_BLOCK, DONE };

struct Wave {
    float a[2][64], b[2][64];
    float a8[2][64][8], b8[2][64][8];   // bf16 fragments of the 32x32x16 MFMA, widened to float
    float sh[2][64];
    unsigned arrived = 0, gen = 0;
};

struct Fiber {
    ucontext_t ctx;
    Idx tid;
    int lane, wave;
    State st;
    unsigned wait_gen;
    unsigned mfma_seq, shfl_seq;
};

struct Block {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    std::vector<char*> stacks;
    char* smem = nullptr;
    size_t smem_cap = 0;
    unsigned arrived = 0, gen = 0, nthreads = 0;
    Idx bid, bdim, gdim;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
    ~Block() { for (char* s : stacks) free(s); free(smem); }
};

static const size_t kStack = 256 * 1024;
inline Block*& tls_block() { static thread_local Block* b = nullptr; return b; }
inline Block& blk() { return *tls_block(); }

inline void yield_to_sched() { Block& b = blk(); swapcontext(&b.cur->ctx, &b.sched); }

inline void trampoline() {
    Block& b = blk();
    (*b.body)();
    b.cur->st = DONE;
    swapcontext(&b.cur->ctx, &b.sched);
}

inline void block_sync() {
    Block& b = blk();
    Fiber* f = b.cur;
    unsigned g = b.gen;
    if (++b.arrived == b.nthreads) { b.arrived = 0; b.gen++; return; }
    f->st = WAIT_BLOCK; f->wait_gen = g;
    yield_to_sched();
}

inline void wave_sync() {
    Block& b = blk();
    Fiber* f = b.cur;
    Wave& w = b.waves[f->wave];
    unsigned g = w.gen;
    if (++w.arrived == 64) { w.arrived = 0; w.gen++; return; }
    f->st = WAIT_WAVE; f->wait_gen = g;
    yield_to_sched();
}

inline void run_block(Block& b, const std::function<void()>& body, Idx bid, Idx bdim, Idx gdim, size_t smem) {
    tls_block() = &b;
    unsigned nt = bdim.x * bdim.y * bdim.z;
    if (nt % 64) { fprintf(stderr, "fsemu: blockDim must be a multiple of 64\n"); abort(); }
    b.nthreads = nt; b.bid = bid; b.bdim = bdim; b.gdim = gdim; b.body = &body;
    b.arrived = 0;
    if (b.fibers.size() < nt) {
        b.fibers.resize(nt);
        while (b.stacks.size() < nt) b.stacks.push_back((char*)aligned_alloc(64, kStack));
    }
    b.waves.assign(nt / 64, Wave());
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define FS_EMU_ASAN 1
#endif
#endif
#ifdef FS_EMU_ASAN   /* sanitized build (tests/emu_lib.build_emu_sanitized): the LDS block is EXACTLY the launch's size, so an out-of-range LDS index lands in a redzone */
    if (smem != b.smem_cap || !b.smem) { free(b.smem); b.smem_cap = smem; b.smem = (char*)aligned_alloc(64, (smem + 63) / 64 * 64 ? (smem + 63) / 64 * 64 : 64); }
#else
    if (smem + 64 > b.smem_cap) { free(b.smem); b.smem_cap = smem + 64; b.smem = (char*)aligned_alloc(64, (b.smem_cap + 63) / 64 * 64); }
#endif
    memset(b.smem, 0xCD, smem);   // poison: reading unwritten LDS shows up as garbage
    for (unsigned t = 0; t < nt; ++t) {
        Fiber& f = b.fibers[t];
        f.tid = Idx{t % bdim.x, (t / bdim.x) % bdim.y, t / (bdim.x * bdim.y)};
        f.lane = t % 64; f.wave = t / 64; f.st = RUN; f.mfma_seq = f.shfl_seq = 0;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = b.stacks[t];
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    unsigned done = 0;
    while (done < nt) {
        bool progressed = false;
        for (unsigned wv = 0; wv < nt / 64; ++wv) {
            bool wave_prog = true;
            while (wave_prog) {
                wave_prog = false;
                for (unsigned l = 0; l < 64; ++l) {
                    Fiber& f = b.fibers[wv * 64 + l];
                    if (f.st == DONE) continue;
                    if (f.st == WAIT_BLOCK) { if (b.gen == f.wait_gen) continue; f.st = RUN; }
                    if (f.st == WAIT_WAVE) { if (b.waves[wv].gen == f.wait_gen) continue; f.st = RUN; }
                    b.cur = &f;
                    swapcontext(&b.sched, &f.ctx);
                    wave_prog = progressed = true;
                    if (f.st == DONE) ++done;
                }
            }
        }
        if (!progressed && done < nt) { fprintf(stderr, "fsemu: deadlock (divergent barrier?)\n"); abort(); }
    }
}

template <class F>
inline void launch(F&& body, dim3 grid, dim3 block, size_t smem) {
    std::function<void()> fn = body;
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    unsigned nthr = std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), nblocks);
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        Block* b = new Block();
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            Idx bid{(unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((size_t)grid.x * grid.y))};
            run_block(*b, fn, bid, Idx{block.x, block.y, block.z}, Idx{grid.x, grid.y, grid.z}, smem);
        }
        delete b;
    };
    if (nthr <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthr; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

inline Idx& tid() { return blk().cur->tid; }
inline Idx& bid() { return blk().bid; }
inline Idx& bdim() { return blk().bdim; }
inline Idx& gdim() { return blk().gdim; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    Block& bk = blk(); Fiber* f = bk.cur; Wave& w = bk.waves[f->wave];
    int s = f->mfma_seq++ & 1, l = f->lane;
    w.a[s][l] = a; w.b[s][l] = b;
    wave_sync();
    int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(w.a[s][i + 32 * k], w.b[s][j + 32 * k], acc);
        c[r] = acc;
    }
    return c;
}

// v_mfma_f32_32x32x16_bf16: lane l holds A[i = l&31][k = 8*(l>>5) .. +7] and B[k = 8*(l>>5) .. +7][j = l&31];
// products of bf16 values are exact in fp32, the sum order (and the width of the adder tree) inside the instruction is not
// architecturally specified -- the emulator sums the 16 exact products and the accumulator in double and rounds ONCE per
// instruction (parity tests of the bf16 paths carry a tolerance, not bit equality; the GPU run is what pins the real sum).
typedef __bf16 bf16x8_emu __attribute__((ext_vector_type(8)));
inline f32x16 mfma_32x32x16_bf16(bf16x8_emu a, bf16x8_emu b, f32x16 c) {
    Block& bk = blk(); Fiber* f = bk.cur; Wave& w = bk.waves[f->wave];
    int s = f->mfma_seq++ & 1, l = f->lane;
    unsigned short ra[8], rb[8];
    __builtin_memcpy(ra, &a, 16);
    __builtin_memcpy(rb, &b, 16);
    for (int k = 0; k < 8; ++k) {
        unsigned ua = (unsigned)ra[k] << 16, ub = (unsigned)rb[k] << 16;
        __builtin_memcpy(&w.a8[s][l][k], &ua, 4);
        __builtin_memcpy(&w.b8[s][l][k], &ub, 4);
    }
    wave_sync();
    int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double acc = c[r];
        for (int k = 0; k < 16; ++k) acc += (double)w.a8[s][i + 32 * (k >> 3)][k & 7] * (double)w.b8[s][j + 32 * (k >> 3)][k & 7];
        c[r] = (float)acc;
    }
    return c;
}

// v_mfma_f32_16x16x32_bf16: lane l holds A[i = l&15][k = 8*(l>>4) .. +7] and B[k = 8*(l>>4) .. +7][j = l&15]; result register r of lane l = D[4*(l>>4) + r][l&15]
// (summed in double, rounded once per instruction: see mfma_32x32x16_bf16)
inline f32x4 mfma_16x16x32_bf16(bf16x8_emu a, bf16x8_emu b, f32x4 c) {
    Block& bk = blk(); Fiber* f = bk.cur; Wave& w = bk.waves[f->wave];
    int s = f->mfma_seq++ & 1, l = f->lane;
    unsigned short ra[8], rb[8];
    __builtin_memcpy(ra, &a, 16);
    __builtin_memcpy(rb, &b, 16);
    for (int k = 0; k < 8; ++k) {
        unsigned ua = (unsigned)ra[k] << 16, ub = (unsigned)rb[k] << 16;
        __builtin_memcpy(&w.a8[s][l][k], &ua, 4);
        __builtin_memcpy(&w.b8[s][l][k], &ub, 4);
    }
    wave_sync();
    int j = l & 15;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * (l >> 4) + r;
        double acc = c[r];
        for (int k = 0; k < 32; ++k) acc += (double)w.a8[s][i + 16 * (k >> 3)][k & 7] * (double)w.b8[s][j + 16 * (k >> 3)][k & 7];
        c[r] = (float)acc;
    }
    return c;
}

inline f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    Block& bk = blk(); Fiber* f = bk.cur; Wave& w = bk.waves[f->wave];
    int s = f->mfma_seq++ & 1, l = f->lane;
    w.a[s][l] = a; w.b[s][l] = b;
    wave_sync();
    int j = l & 15;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.a[s][i + 16 * k], w.b[s][j + 16 * k], acc);
        c[r] = acc;
    }
    return c;
}

inline float shfl_idx(float v, int src) {
    Block& bk = blk(); Fiber* f = bk.cur; Wave& w = bk.waves[f->wave];
    int s = f->shfl_seq++ & 1;
    w.sh[s][f->lane] = v;
    wave_sync();
    return w.sh[s][src & 63];
}

}  // namespace fsemu

#define threadIdx (fsemu::tid())
#define blockIdx (fsemu::bid())
#define blockDim (fsemu::bdim())
#define gridDim (fsemu::gdim())
#define __syncthreads() fsemu::block_sync()
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(fsemu::blk().smem);

#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) fsemu::mfma_32x32x2((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) fsemu::mfma_16x16x4((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) fsemu::mfma_32x32x16_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) fsemu::mfma_16x16x32_bf16((a), (b), (c))
#define __builtin_amdgcn_s_getreg(imm) 0u   /* hardware id register: slot 0 everywhere */

/* buffer resources: a (base, size) pair; loads beyond the size return zeros, as the hardware range check does.  The kernels mark a
 * DELIBERATELY dropped access with the offset 0x80000000 or with an EMPTY buffer (size 0); any other out-of-range access is a bug
 * and aborts. */
namespace fsemu {
struct buffer_rsrc { const unsigned char* base; unsigned nbytes; };
static inline uint4 raw_buffer_load_b128(buffer_rsrc r, unsigned voff, unsigned soff) {
    uint4 v{0u, 0u, 0u, 0u};
    const unsigned long long end = (unsigned long long)voff + soff + 16ull;
    if (voff < 0x80000000u && end <= r.nbytes) memcpy(&v, r.base + voff + soff, 16);
    else if (voff < 0x80000000u && r.nbytes) { fprintf(stderr, "emu: buffer load out of range (%u+%u of %u)\n", voff, soff, r.nbytes); abort(); }
    return v;
}
static inline unsigned raw_buffer_load_b32(buffer_rsrc r, unsigned voff, unsigned soff) {
    unsigned v = 0u;
    const unsigned long long end = (unsigned long long)voff + soff + 4ull;
    if (voff < 0x80000000u && end <= r.nbytes) memcpy(&v, r.base + voff + soff, 4);
    else if (voff < 0x80000000u && r.nbytes) { fprintf(stderr, "emu: buffer load out of range (%u+%u of %u)\n", voff, soff, r.nbytes); abort(); }
    return v;
}
struct uint3_emu { unsigned x, y, z; };
static inline uint3_emu raw_buffer_load_b96(buffer_rsrc r, unsigned voff, unsigned soff) {
    uint3_emu v{0u, 0u, 0u};
    const unsigned long long end = (unsigned long long)voff + soff + 12ull;
    if (voff < 0x80000000u && end <= r.nbytes) memcpy(&v, r.base + voff + soff, 12);
    else if (voff < 0x80000000u && r.nbytes) { fprintf(stderr, "emu: buffer load out of range (%u+%u of %u)\n", voff, soff, r.nbytes); abort(); }
    return v;
}
static inline void raw_buffer_store_b32(unsigned v, buffer_rsrc r, unsigned voff, unsigned soff) {
    const unsigned long long end = (unsigned long long)voff + soff + 4ull;
    if (voff < 0x80000000u && end <= r.nbytes) memcpy(const_cast<unsigned char*>(r.base) + voff + soff, &v, 4);
    else if (voff < 0x80000000u) { fprintf(stderr, "emu: buffer store out of range (%u+%u of %u)\n", voff, soff, r.nbytes); abort(); }
}
typedef unsigned int u32x4_emu __attribute__((ext_vector_type(4)));
static inline void raw_buffer_store_b128(u32x4_emu v, buffer_rsrc r, unsigned voff, unsigned soff) {
    const unsigned long long end = (unsigned long long)voff + soff + 16ull;
    if (voff < 0x80000000u && end <= r.nbytes) memcpy(const_cast<unsigned char*>(r.base) + voff + soff, &v, 16);
    else if (voff < 0x80000000u) { fprintf(stderr, "emu: buffer store out of range (%u+%u of %u)\n", voff, soff, r.nbytes); abort(); }
}
static inline float fmed3f(float a, float b, float c) {
    const float lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}
}  // namespace fsemu
#define __amdgpu_buffer_rsrc_t fsemu::buffer_rsrc
#define __builtin_amdgcn_make_buffer_rsrc(ptr, stride, nbytes, flags) \
    fsemu::buffer_rsrc{reinterpret_cast<const unsigned char*>(ptr), (unsigned)(nbytes)}
#define __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, aux) fsemu::raw_buffer_load_b128((r), (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, aux) fsemu::raw_buffer_load_b96((r), (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, aux) fsemu::raw_buffer_load_b32((r), (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, soff, aux) fsemu::raw_buffer_store_b32((v), (r), (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, aux) fsemu::raw_buffer_store_b128((v), (r), (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_fmed3f(a, b, c) fsemu::fmed3f((a), (b), (c))
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
/* global_load_lds_dwordx4: lane l copies 16 bytes from its own source to (wave-uniform LDS base) + 16*l */
static inline void fs_emu_global_load_lds_b128(const void* gsrc, void* lds_wave) {
    memcpy(static_cast<char*>(lds_wave) + 16 * fsemu::blk().cur->lane, gsrc, 16);
}
#define __builtin_amdgcn_s_sleep(imm) ((void)0)
/* a wave executes in lockstep on the GPU; the emulator's fibers do not -- where lanes of ONE wave exchange data through LDS
 * without a workgroup barrier (fs_wino4.hip's transform), the kernel marks the hand-off with a wave barrier */
#define __builtin_amdgcn_wave_barrier() fsemu::wave_sync()
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)   /* only ever applied to wave-uniform values */

static inline float __shfl_xor(float v, int mask, int width = 64) { (void)width; return fsemu::shfl_idx(v, fsemu::blk().cur->lane ^ mask); }
static inline float __shfl_down(float v, int d, int width = 64) { (void)width; int l = fsemu::blk().cur->lane; return fsemu::shfl_idx(v, l + d > 63 ? l : l + d); }
static inline float __shfl(float v, int src, int width = 64) { (void)width; return fsemu::shfl_idx(v, src); }
static inline float atomicAdd(float* p, float v) {
    static std::atomic_flag lk = ATOMIC_FLAG_INIT;
    while (lk.test_and_set(std::memory_order_acquire)) {}
    float o = *p; *p = o + v;
    lk.clear(std::memory_order_release);
    return o;
}
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }   /* blocks run on several host threads */
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
// individually rounded float ops (the emulator build uses -ffp-contract=off semantics for these)
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
using std::max;
using std::min;
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <class T>
static inline hipError_t hipFuncSetAttribute(T, hipFuncAttribute, int) { return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...)                        \
    do {                                                                                   \
        (void)(stream);                                                                    \
        fsemu::launch([=]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (smem));    \
    } while (0)
