// Torch-free hardware check of the kernels written without GPU time, through the C ABI of include/migan.h:
// each staged kernel against the kernel it replaces (the same entry point with its migan_staged() bit cleared), results
// compared on the host, both timed with hipEvents (min of 10 launches).  It links libmigan.so and the HIP runtime only, so a GPU
// call costs seconds instead of the minute or two a first `import torch` takes on a fresh box:
//     make -C tools abi_check && ./tools/abi_check.bin            (or: python tools/build_abi_check.py)
// Exit status 0 = every comparison within its bound.  Test tooling, not a product path.
#ifndef ABI_CHECK_HOST
#include <hip/hip_runtime.h>
#else
// -DABI_CHECK_HOST: this same program against tests/hipemu's host execution model of the kernels (libmigan_emu.so takes host
// pointers) - checks the harness itself (arguments, geometry, comparisons) without a GPU; the times it prints mean nothing.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int hipError_t;
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
typedef std::chrono::steady_clock::time_point* hipEvent_t;
struct hipDeviceProp_t { char name[64]; int multiProcessorCount; };
static inline const char* hipGetErrorString(hipError_t) { return "host"; }
static inline hipError_t hipMalloc(float** p, size_t n) { *p = (float*)malloc(n); return 0; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { snprintf(p->name, 64, "host execution model"); p->multiProcessorCount = 0; return 0; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new std::chrono::steady_clock::time_point; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, int) { *e = std::chrono::steady_clock::now(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(*b - *a).count(); return 0; }
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../include/migan.h"

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
            exit(3);                                                                            \
        }                                                                                       \
    } while (0)
#define RC(x)                                                                     \
    do {                                                                          \
        int r_ = (x);                                                             \
        if (r_ != 0) {                                                            \
            printf("migan rc %d (%s) at %s:%d\n", r_, migan_error_string(r_), __FILE__, __LINE__); \
            exit(4);                                                              \
        }                                                                         \
    } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float frand() {  // uniform in (-1, 1)
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return (float)((rng_state * 0x2545F4914F6CDD1Dull) >> 40) / (float)(1 << 23) - 1.0f;
}
struct Buf {
    float* d = nullptr;
    size_t n = 0;
    explicit Buf(size_t n_, float scale = 1.f, bool fill = true) : n(n_) {
        CK(hipMalloc(&d, std::max<size_t>(n, 4) * sizeof(float)));
        CK(hipMemset(d, 0xFF, std::max<size_t>(n, 4) * sizeof(float)));   // NaN until written
        std::vector<float> h(n);
        for (auto& v : h) v = fill ? frand() * scale : NAN;
        if (n && fill) CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    ~Buf() { (void)hipFree(d); }
    std::vector<float> host() const {
        std::vector<float> h(n);
        CK(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
        return h;
    }
};
static double rel(const std::vector<float>& a, const std::vector<float>& b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        if (!(a[i] == a[i])) return INFINITY;  // a NaN: an element the kernel did not write
        num += ((double)a[i] - b[i]) * ((double)a[i] - b[i]);
        den += (double)b[i] * b[i];
    }
    return std::sqrt(num / std::max(den, 1e-60));
}
#ifdef ABI_CHECK_HOST
static const int REPS = 1;
#else
static const int REPS = 10;
#endif
static float time_us(const std::function<void()>& f, int reps = REPS) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    f();
    CK(hipDeviceSynchronize());
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a, 0));
        f();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms * 1e3f);
    }
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return best;
}
static int failures = 0;
static void report(const char* what, const char* kernel, double r, double tol, float us_new, float us_old, double mbytes = 0) {
    const bool ok = r <= tol;
    if (!ok) ++failures;
    printf("%-34s %-18s rel %.2e (<= %.0e) %-4s  staged %8.1f us   replaced %8.1f us", what, kernel, r, tol, ok ? "ok" : "FAIL", us_new, us_old);
    if (mbytes > 0) printf("   %.0f GB/s of %.1f MB", mbytes / us_new * 1e3, mbytes);
    printf("\n");
    fflush(stdout);
}

enum { B_THIN = 1, B_RTR = 2, B_MIDK = 4, B_NORM = 8, B_PB16 = 16, B_PTR = 32, B_FEW = 64, B_ALL = 127 };

struct Conv {
    int N, Ci, H, W, Co, k, stride, pt, pl, pb, pr, act;
    bool bias;
    int Ho() const { return (H + pt + pb - k) / stride + 1; }
    int Wo() const { return (W + pl + pr - k) / stride + 1; }
};

// forward / dgrad / wgrad of one conv geometry with `bit` set and cleared
static void conv_case(const char* name, const Conv& c, unsigned bit, bool fwd, bool dgrad, bool wgrad) {
    const int Ho = c.Ho(), Wo = c.Wo();
    const size_t nx = (size_t)c.N * c.H * c.W * c.Ci, ny = (size_t)c.N * Ho * Wo * c.Co, nw = (size_t)c.Co * c.Ci * c.k * c.k;
    Buf x(nx), w_oihw(nw, 0.05f), w_ohwi(nw, 0, false), w_ihwo(nw, 0, false), b(c.Co), dy(ny);
    Buf sk(migan_conv_splitk_workspace() / 4, 0.f);
    CK(hipMemset(sk.d, 0, sk.n * 4));
    migan_staged(B_ALL, 0);
    RC(migan_permute4d(w_oihw.d, w_ohwi.d, c.Co, c.Ci, c.k, c.k, 0, 2, 3, 1, nullptr));
    RC(migan_permute4d(w_oihw.d, w_ihwo.d, c.Co, c.Ci, c.k, c.k, 1, 2, 3, 0, nullptr));
    char kn[64];
    if (fwd) {
        std::vector<float> out[2];
        float us[2];
        for (int s = 0; s < 2; ++s) {
            migan_staged(B_ALL, s == 0 ? bit : 0);
            Buf y(ny, 0, false);
            auto run = [&] {
                RC(migan_conv2d_fwd_ws(x.d, w_ohwi.d, c.bias ? b.d : nullptr, nullptr, y.d, c.N, c.H, c.W, c.Ci, Ho, Wo, c.Co, c.k, c.k, c.stride,
                                       c.pt, c.pl, 0, c.act, 0.2f, sk.d, sk.n * 4, nullptr));
            };
            us[s] = time_us(run);
            out[s] = y.host();
        }
        snprintf(kn, sizeof kn, "fwd");
        report(name, kn, rel(out[0], out[1]), 1e-4, us[0], us[1]);
    }
    if (dgrad && c.pt == c.pb && c.pl == c.pr) {
        std::vector<float> out[2];
        float us[2];
        for (int s = 0; s < 2; ++s) {
            migan_staged(B_ALL, s == 0 ? bit : 0);
            Buf dx(nx, 0, false);
            auto run = [&] {
                RC(migan_conv2d_dgrad_ws(dy.d, w_ihwo.d, nullptr, dx.d, c.N, c.H, c.W, c.Ci, Ho, Wo, c.Co, c.k, c.k, c.stride, c.pt, c.pl, 0, 0.f,
                                         sk.d, sk.n * 4, nullptr));
            };
            us[s] = time_us(run);
            out[s] = dx.host();
        }
        report(name, "dgrad", rel(out[0], out[1]), 1e-4, us[0], us[1]);
    }
    if (wgrad) {
        std::vector<float> out[2];
        float us[2];
        const size_t wsb = migan_conv2d_wgrad_workspace(c.N, Ho, Wo, c.Co, c.k, c.k, c.Ci);
        for (int s = 0; s < 2; ++s) {
            migan_staged(B_ALL, s == 0 ? bit : 0);
            Buf dw(nw, 0, false), ws(std::max<size_t>(wsb / 4, 4), 0, false);
            auto run = [&] {
                RC(migan_conv2d_wgrad(x.d, dy.d, dw.d, ws.d, wsb, c.N, c.H, c.W, c.Ci, Ho, Wo, c.Co, c.k, c.k, c.stride, c.pt, c.pl, 0, 0, nullptr, 0,
                                      nullptr, 0, nullptr));
            };
            us[s] = time_us(run);
            out[s] = dw.host();
        }
        report(name, "wgrad(+reduce)", rel(out[0], out[1]), 1e-4, us[0], us[1]);
    }
}

static void pack_case(const char* name, int d0, int d1, int R) {
    const size_t n = (size_t)d0 * d1 * R;
    Buf src(n);
    for (int perm = 0; perm < 2; ++perm) {
        std::vector<float> out[2];
        float us[2];
        for (int s = 0; s < 2; ++s) {
            migan_staged(B_ALL, s == 0 ? B_PTR : 0);
            Buf dst(n, 0, false);
            auto run = [&] {
                if (perm == 0) RC(migan_permute4d(src.d, dst.d, d0, d1, R, 1, 0, 2, 3, 1, nullptr));
                else RC(migan_permute4d(src.d, dst.d, d0, d1, R, 1, 1, 2, 3, 0, nullptr));
            };
            us[s] = time_us(run);
            out[s] = dst.host();
        }
        report(name, perm == 0 ? "pack ohwi" : "pack ihwo", rel(out[0], out[1]), 0.0, us[0], us[1], 2.0 * n * 4e-6);
    }
}

// the few-pixel conv path against the tiled kernels: Conv2d(Ci, Co, 4, 2, 1) on H x W
static void fewpix_case(const char* name, int N, int Ci, int H, int W, int Co) {
    const int Ho = H / 2, Wo = W / 2, M = N * Ho * Wo, K = Ci * 16;
    const size_t nx = (size_t)N * H * W * Ci, ny = (size_t)M * Co, nw = (size_t)Co * K;
    Buf x(nx), w(nw, 0.05f), w_ohwi(nw, 0, false), w_ihwo(nw, 0, false), b(Co), dy(ny), sk(migan_conv_splitk_workspace() / 4, 0.f);
    CK(hipMemset(sk.d, 0, sk.n * 4));
    migan_staged(B_ALL, B_FEW);
    if (migan_fewpix_ok(M, Co, K) != 1) {
        printf("%-34s not a few-pixel shape\n", name);
        return;
    }
    const double wmb = nw * 4e-6;
    // forward
    Buf col(( size_t)M * K, 0, false), y1(ny, 0, false), y0(ny, 0, false);
    const size_t nb = migan_fewpix_nt_workspace(M, Co, K);
    Buf ws(std::max<size_t>(nb / 4, 4), 0, false);
    float t_new = time_us([&] {
        RC(migan_im2col_small(x.d, col.d, N, H, W, Ci, Ho, Wo, 4, 4, 2, 1, 1, nullptr));
        RC(migan_fewpix_nt(col.d, w.d, b.d, y1.d, ws.d, nb, M, Co, K, 1, 0.2f, nullptr));
    });
    migan_staged(B_ALL, 0);
    float t_old = time_us([&] {
        RC(migan_permute4d(w.d, w_ohwi.d, Co, Ci, 4, 4, 0, 2, 3, 1, nullptr));
        RC(migan_conv2d_fwd_ws(x.d, w_ohwi.d, b.d, nullptr, y0.d, N, H, W, Ci, Ho, Wo, Co, 4, 4, 2, 1, 1, 0, 1, 0.2f, sk.d, sk.n * 4, nullptr));
    });
    report(name, "fwd (+pack)", rel(y1.host(), y0.host()), 1e-4, t_new, t_old, wmb);
    // input gradient
    Buf ycol((size_t)M * K, 0, false), dx1(nx, 0, false), dx0(nx, 0, false);
    t_new = time_us([&] {
        RC(migan_skinny_nn(dy.d, w.d, ycol.d, M, Co, K, nullptr));
        RC(migan_col2im_small(ycol.d, nullptr, dx1.d, N, H, W, Ci, Ho, Wo, 4, 4, 2, 1, 1, 0, 0.f, nullptr));
    });
    t_old = time_us([&] {
        RC(migan_permute4d(w.d, w_ihwo.d, Co, Ci, 4, 4, 1, 2, 3, 0, nullptr));
        RC(migan_conv2d_dgrad_ws(dy.d, w_ihwo.d, nullptr, dx0.d, N, H, W, Ci, Ho, Wo, Co, 4, 4, 2, 1, 1, 0, 0.f, sk.d, sk.n * 4, nullptr));
    });
    report(name, "dgrad (+pack)", rel(dx1.host(), dx0.host()), 1e-4, t_new, t_old, wmb);
    // weight gradient
    Buf dw1(nw, 0, false), dw0(nw, 0, false);
    const size_t wsb = migan_conv2d_wgrad_workspace(N, Ho, Wo, Co, 4, 4, Ci);
    Buf wsl(std::max<size_t>(wsb / 4, 4), 0, false);
    t_new = time_us([&] { RC(migan_skinny_tn(dy.d, col.d, dw1.d, nullptr, M, Co, K, 0, 0, nullptr)); });
    t_old = time_us([&] {
        RC(migan_conv2d_wgrad(x.d, dy.d, dw0.d, wsl.d, wsb, N, H, W, Ci, Ho, Wo, Co, 4, 4, 2, 1, 1, 0, 0, nullptr, 0, nullptr, 0, nullptr));
    });
    report(name, "wgrad", rel(dw1.host(), dw0.host()), 1e-4, t_new, t_old, wmb);
}

int main(int argc, char** argv) {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, dev));
    printf("%s | %s | %d CUs | staged word from the environment: %u\n", migan_version(), p.name, p.multiProcessorCount, migan_staged(0, 0));
    const std::string only = argc > 1 ? argv[1] : "";
    auto want = [&](const char* s) { return only.empty() || only == s; };
    if (want("thin")) {
        conv_case("patchgan head 512->1 @16x16 b1", {1, 512, 16, 16, 1, 4, 1, 2, 2, 1, 1, 0, true}, B_THIN | B_PB16, true, false, false);
        conv_case("patchgan head 512->1 @16x16 b8", {8, 512, 16, 16, 1, 4, 1, 2, 2, 1, 1, 0, true}, B_THIN | B_PB16, true, false, false);
        conv_case("patchgan head sym-pad dgrad b1", {1, 512, 17, 17, 1, 4, 1, 1, 1, 1, 1, 0, false}, B_THIN | B_PB16, true, true, false);
    }
    if (want("midk")) {
        conv_case("pix2pix 6->64 4x4 s2 @256", {1, 6, 256, 256, 64, 4, 2, 1, 1, 1, 1, 1, false}, B_MIDK, true, false, false);
        conv_case("cyclegan D 3->64 4x4 s2 @256 b8", {8, 3, 256, 256, 64, 4, 2, 1, 1, 1, 1, 1, true}, B_MIDK, true, false, false);
        conv_case("srgan D 3->64 3x3 @384 b16", {16, 3, 384, 384, 64, 3, 1, 1, 1, 1, 1, 1, true}, B_MIDK, true, false, false);
    }
    if (want("reduce")) {
        conv_case("unet 256->512 4x4 s2 @32 (2 M w)", {1, 256, 32, 32, 512, 4, 2, 1, 1, 1, 1, 0, false}, B_RTR, false, false, true);
        conv_case("unet 512->512 4x4 s2 @16 (4 M w)", {1, 512, 16, 16, 512, 4, 2, 1, 1, 1, 1, 0, false}, B_RTR, false, false, true);
    }
    if (want("pack")) {
        pack_case("512x512x16 (4 M)", 512, 512, 16);
        pack_case("1024x512x16 (8 M)", 1024, 512, 16);
        pack_case("256x128x16 (512 k)", 256, 128, 16);
    }
    if (want("fewpix")) {
        fewpix_case("fewpix 512->512 @2x2 (d8)", 1, 512, 2, 2, 512);
        fewpix_case("fewpix 512->512 @4x4 (d7)", 1, 512, 4, 4, 512);
        fewpix_case("fewpix 512->512 @8x8 (d6)", 1, 512, 8, 8, 512);
        fewpix_case("fewpix 512->512 @16x16 (d5)", 1, 512, 16, 16, 512);
        fewpix_case("fewpix 1024->512 @8x8", 1, 1024, 8, 8, 512);
    }
    migan_staged(B_ALL, 0);
    printf(failures ? "FAILED: %d comparison(s) out of bound\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
