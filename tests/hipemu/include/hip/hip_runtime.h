// TEST INFRASTRUCTURE - not part of the product.  A host-side execution model for the HIP kernels of
// pytorch-gan_amd/csrc, so that kernel LOGIC (index arithmetic, LDS staging, barriers, wave collectives, MFMA fragment
// layouts, LDS-DMA addressing, ticket protocols) can be exercised by the `-m "not gpu"` tests in a container without a
// GPU.  tests/hipemu/build_emu.py compiles the unchanged kernel sources as plain C++ against THIS header (it shadows
// <hip/hip_runtime.h>) into tests/hipemu/_build/libmigan_emu.so; only tests load that library.  Nothing under
// pytorch-gan_amd/ knows it exists, and the product path still fails loudly without the real gfx950 library.
//
// Model: one fiber per HIP thread, the fibers of a workgroup scheduled round-robin on one OS thread and switched only at
// synchronisation points (__syncthreads, wave collectives, s_sleep); workgroups are distributed over a few OS threads (all
// of them at once in "co-resident" mode, for kernels with grid-wide waits).  Waves are 64 consecutive threads.  What is NOT
// modelled: timing, asynchrony of ordinary global loads (LDS-DMA is asynchronous: see Ctx::dma), divergent wave collectives,
// memory-ordering bugs.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <functional>
#include <type_traits>

namespace hipemu {
struct U3 {
    unsigned x, y, z;
};
struct DmaOp {  // one lane's share of an LDS-DMA instruction that has been issued and not yet waited for
    char* dst;
    const char* src;   // nullptr: zeros (out of the descriptor's range)
    int size;
};
struct Ctx {  // per HIP thread
    U3 tid, bid, bdim, gdim;
    int lane, wave, linear;
    int phase;  // double-buffer index of the wave exchange slots
    // LDS-DMA is asynchronous: data lands only when the issuing wave waits for it (s_waitcnt vmcnt(N): all but the N youngest
    // loads; __syncthreads(): all of them - the compiler puts vmcnt(0) in front of a barrier when an LDS-DMA may be in flight).
    // A kernel that reads a stage before waiting for it sees the stage's OLD contents here, as it could on the hardware.
    DmaOp dma[64];
    int dma_head, dma_n;
};
extern thread_local Ctx* cur;
void syncthreads();
void barrier_raw();          // s_barrier without the wait for outstanding LDS-DMA
void waitcnt_vm(int n);      // s_waitcnt vmcnt(n) for the LDS-DMA queue of this lane
void yield();
void sleep_hint();   // s_sleep: let the other fibers run - and, in a co-resident launch, the other workgroups' OS threads
void* dyn_lds();
void poison_lds(void* p, size_t n, unsigned long* seen);
// wave rendezvous: publish `words` 32-bit words, wait for every live lane of the wave, return the wave's slot array
// ([64][8] words) of this exchange; valid until this lane's next exchange.
const uint32_t* wave_exchange(const uint32_t* mine, int words);
int launch_impl(const char* kernel, U3 grid, U3 block, size_t lds, const std::function<void()>& body);
int last_error();
void note_error(const char* what);

struct BufRsrc {
    const char* base;
    unsigned bytes;
};
}  // namespace hipemu

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipGraph_t* hipGraph_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719, hipErrorNotSupported = 801 };
static inline hipError_t hipGetLastError() { return hipemu::last_error(); }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
// dynamic LDS beyond 64 KB needs this opt-in on the hardware; the model's dynamic LDS is sized per launch
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }   // the model is one device
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }   // streams are synchronous here

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::cur->bid)
#define blockDim (hipemu::cur->bdim)
#define gridDim (hipemu::cur->gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...)                                            \
    do {                                                                                                   \
        const dim3 g__ = (grid), b__ = (block);                                                            \
        (void)(stream);                                                                                    \
        hipemu::launch_impl(#kern, hipemu::U3{g__.x, g__.y, g__.z}, hipemu::U3{b__.x, b__.y, b__.z}, (size_t)(lds), \
                            [=]() { kern(__VA_ARGS__); });                                                 \
    } while (0)

static inline void __syncthreads() { hipemu::syncthreads(); }

// ---- wave collectives -------------------------------------------------------------------------------------------------
template <class T>
static inline T hipemu_shfl_idx(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shuffle operand");
    uint32_t w[2] = {0, 0};
    memcpy(w, &v, sizeof(T));
    const uint32_t* all = hipemu::wave_exchange(w, 2);
    T r;
    memcpy(&r, all + (src_lane & 63) * 8, sizeof(T));
    return r;
}
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    return hipemu_shfl_idx(v, hipemu::cur->lane ^ mask);
}
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
    (void)width;
    return hipemu_shfl_idx(v, src);
}
template <class T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
    (void)width;
    const int s = hipemu::cur->lane + (int)d;
    return hipemu_shfl_idx(v, s < 64 ? s : hipemu::cur->lane);
}
template <class T>
static inline T __shfl_up(T v, unsigned d, int width = 64) {
    (void)width;
    const int s = hipemu::cur->lane - (int)d;
    return hipemu_shfl_idx(v, s >= 0 ? s : hipemu::cur->lane);
}

typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; accumulator register r of
// lane l is D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31].  Exact f32: an fmaf chain over k (cdna_hip_programming.md).
static inline hipemu_f32x16 hipemu_mfma_32x32x2(float a, float b, hipemu_f32x16 c) {
    uint32_t w[2];
    memcpy(&w[0], &a, 4);
    memcpy(&w[1], &b, 4);
    const uint32_t* all = hipemu::wave_exchange(w, 2);
    const int l = hipemu::cur->lane, j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, all + (k * 32 + i) * 8, 4);
            memcpy(&bv, all + (k * 32 + j) * 8 + 1, 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; register r of lane l is D[4 * (l >> 4) + r][l & 15]
static inline hipemu_f32x4 hipemu_mfma_16x16x4(float a, float b, hipemu_f32x4 c) {
    uint32_t w[2];
    memcpy(&w[0], &a, 4);
    memcpy(&w[1], &b, 4);
    const uint32_t* all = hipemu::wave_exchange(w, 2);
    const int l = hipemu::cur->lane, j = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, all + (k * 16 + i) * 8, 4);
            memcpy(&bv, all + (k * 16 + j) * 8 + 1, 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu_mfma_32x32x2((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_16x16x4((a), (b), (c))

// ---- LDS-DMA ------------------------------------------------------------------------------------------------------------
// buffer_load_dword(x4) ... lds: each lane copies `size` bytes from base + voffset + soffset + imm to (wave-uniform LDS base)
// + lane * size; a lane whose voffset + imm lies outside the descriptor's range writes zeros (raw buffer range check).
typedef hipemu::BufRsrc __amdgpu_buffer_rsrc_t;
static inline hipemu::BufRsrc hipemu_make_rsrc(void* p, short stride, int num_records, int flags) {
    (void)stride;
    (void)flags;
    return hipemu::BufRsrc{(const char*)p, (unsigned)num_records};
}
#define __builtin_amdgcn_make_buffer_rsrc(p, s, n, f) hipemu_make_rsrc((p), (s), (n), (f))
template <class LdsPtr>
static inline void hipemu_buffer_load_lds(hipemu::BufRsrc r, LdsPtr lds, int size, int voffset, int soffset, int imm, int aux) {
    (void)aux;
    const uintptr_t basev = (uintptr_t)lds;  // M0 on the hardware: wave-uniform by construction in these kernels (not checked)
    hipemu::Ctx* c = hipemu::cur;
    char* dst = (char*)basev + (size_t)c->lane * size;
    const uint64_t off = (uint64_t)(uint32_t)voffset + (uint64_t)(uint32_t)imm;
    const char* src = nullptr;
    if (off + (uint64_t)size <= (uint64_t)r.bytes) {
        // the SGPR offset is outside the hardware's range check: a lane that passes it and still leaves the tensor is a bug
        if (off + (uint32_t)soffset + (uint64_t)size > (uint64_t)r.bytes)
            hipemu::note_error("LDS-DMA: voffset in range but voffset + soffset reads past the tensor");
        else
            src = r.base + off + (uint32_t)soffset;
    }
    if (c->dma_n == 64) hipemu::waitcnt_vm(63);   // queue depth of the model; the hardware's counter saturates as well
    c->dma[(c->dma_head + c->dma_n) & 63] = hipemu::DmaOp{dst, src, size};
    c->dma_n++;
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, size, vo, so, imm, aux) \
    hipemu_buffer_load_lds((r), (lds), (size), (vo), (so), (imm), (aux))

// ---- scalar helpers, scheduling hints, atomics ------------------------------------------------------------------------
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
// a wave executes in lockstep: what one lane does behind this point comes after what EVERY lane did in front of it (the fibers of the
// model run one lane at a time between synchronisation points, so the point has to exist here; on the hardware it emits no code)
#define __builtin_amdgcn_wave_barrier() ((void)__shfl(0, 0))
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) hipemu::sleep_hint()
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
// (generic forms: the kernels also hand float tiles over with relaxed agent-scope - sc1 - stores and loads)
template <class T>
static inline T hipemu_atomic_load(const T* p) {
    T v;
    __atomic_load(p, &v, __ATOMIC_SEQ_CST);
    return v;
}
template <class T, class V>
static inline void hipemu_atomic_store(T* p, V v) {
    T t = (T)v;
    __atomic_store(p, &t, __ATOMIC_SEQ_CST);
}
#define __hip_atomic_load(p, order, scope) hipemu_atomic_load((p))
#define __hip_atomic_store(p, v, order, scope) hipemu_atomic_store((p), (v))
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
}
static inline float atomicAdd(float* p, float v) {
    uint32_t* q = (uint32_t*)p;
    uint32_t old = __atomic_load_n(q, __ATOMIC_SEQ_CST), want;
    float f;
    do {
        memcpy(&f, &old, 4);
        f += v;
        memcpy(&want, &f, 4);
    } while (!__atomic_compare_exchange_n(q, &old, want, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
    memcpy(&f, &old, 4);
    return f;
}
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline int __mul24(int a, int b) { return a * b; }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
// IEEE round-to-nearest single operations the compiler must not contract or reassociate
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __ffs(int x) { return __builtin_ffs(x); }
template <class A, class B>
static inline typename std::common_type<A, B>::type min(A a, B b) { return a < b ? a : b; }
template <class A, class B>
static inline typename std::common_type<A, B>::type max(A a, B b) { return a > b ? a : b; }
