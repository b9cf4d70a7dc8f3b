// Shared device/host helpers for the migan HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <atomic>

#define MIGAN_API extern "C" __attribute__((visibility("default")))

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Activation codes shared by every launcher (see include/migan.h).
enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4 };
// Gather modes of the implicit-GEMM loaders and of gather2d.
enum { GATHER_ZERO = 0, GATHER_REFLECT = 1, GATHER_UP2 = 2 };

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case ACT_LRELU: return v > 0.f ? v : v * slope;
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_TANH: return tanhf(v);
        case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// d(act)/d(pre) expressed with the activation OUTPUT y (valid for all codes above;
// LeakyReLU/ReLU: sign(y) == sign(pre) because slope >= 0).
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
    switch (act) {
        case ACT_LRELU: return y > 0.f ? 1.f : slope;
        case ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case ACT_TANH: return 1.f - y * y;
        case ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// Logical -> physical coordinate of the gather. v is a coordinate in the logical
// (padded-free) extent L. Returns false when the tap reads a zero.
__device__ __forceinline__ bool map_coord(int v, int L, int mode, int& src) {
    if (mode == GATHER_REFLECT) {
        if (v < 0) v = -v;
        if (v >= L) v = 2 * L - 2 - v;
        src = v;
        return true;
    }
    if ((unsigned)v >= (unsigned)L) return false;
    src = (mode == GATHER_UP2) ? (v >> 1) : v;
    return true;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Integer A/B knob from the environment, read ONCE per call site - unless MIGAN_TEST_KNOBS is set when the library is first used: then every
// call re-reads it, so that one test process can run a kernel family under several settings (tests/conftest.py sets it).
#define MIGAN_KNOB(name, dflt)                                                                       \
    ([]() -> int {                                                                                   \
        static const bool dyn__ = getenv("MIGAN_TEST_KNOBS") != nullptr;                            \
        static const int cached__ = getenv(name) ? atoi(getenv(name)) : (dflt);                      \
        return dyn__ ? (getenv(name) ? atoi(getenv(name)) : (dflt)) : cached__;                      \
    }())

// q = n / d for 0 <= n < 2^31 without a divide: q = (umulhi(n, m) + n) >> s  (Granlund-Montgomery, s = ceil(log2 d))
static inline void fastdiv_magic(unsigned d, unsigned& m, int& s) {
    s = 0;
    while ((1ull << s) < d) ++s;
    m = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
}
__device__ __forceinline__ int fastdiv(int n, unsigned m, int s) {
    return (int)((__umulhi((unsigned)n, m) + (unsigned)n) >> s);
}

// Debug launch counters: every kernel launch of the library goes through MIGAN_LAUNCH, which counts it per launch SITE (a static
// record per site, linked into one list; a relaxed increment per launch).  migan_debug_launch_count(substr) sums the sites whose
// kernel expression contains substr - the parity tests use it to assert WHICH kernel family served a geometry on the hardware
// (the host execution model of tests/hipemu keeps the same count by kernel expression).
namespace migan_dbg {
struct Site {
    const char* name;
    std::atomic<long> n{0};
    Site* next = nullptr;
    explicit Site(const char* nm);
};
// one list per library: an inline function's static local is shared by the translation units of the shared object
inline std::atomic<Site*>& sites() {
    static std::atomic<Site*> head{nullptr};
    return head;
}
inline Site::Site(const char* nm) : name(nm) {
    Site* h = sites().load(std::memory_order_acquire);
    do next = h;
    while (!sites().compare_exchange_weak(h, this, std::memory_order_release, std::memory_order_acquire));
}
}  // namespace migan_dbg
#define MIGAN_LAUNCH(kern, ...)                                   \
    do {                                                          \
        static migan_dbg::Site site__(#kern);                     \
        site__.n.fetch_add(1, std::memory_order_relaxed);         \
        hipLaunchKernelGGL(kern, __VA_ARGS__);                    \
    } while (0)

#define HIP_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)
