// Torch-free hardware check of latency-regime kernels through the C ABI of include/migan.h: the one-launch InstanceNorm and the
// fused WGAN-GP kernels against a host fp64 evaluation, timed with hipEvents (min of 10 launches).  It links libmigan.so and the HIP
// runtime only, so a GPU call costs seconds instead of the minute or two a first `import torch` takes on a fresh box:
//     tools/build_abi_check.sh && ./tools/abi_check.bin [floor|norm|critic|mlp]      (host: tools/build_abi_check.sh host)
// (Round 3 used it to compare each kernel written without GPU time against the kernel it replaced: profiles/r03_abi_check.txt;
// those A/B sections went away with the kernels' run-time switches - their parity is tests/test_ops_gpu.py::test_geometry_selects_kernel.)
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
// Every "device" buffer ends (to 16 bytes) at an inaccessible page, starts behind one and is filled with NaN: a kernel that reads or writes
// past one of its operands - which a GPU run never notices, the neighbouring allocation is mapped - dies with SIGSEGV here, and a
// workspace word read before it was written poisons the compared result.  (Buffers are not unmapped: the program is short-lived.)
#include <sys/mman.h>
static inline hipError_t hipMalloc(float** p, size_t n) {
    const size_t page = 4096, body = (n + page - 1) / page * page;
    char* m = (char*)mmap(nullptr, body + 2 * page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == (char*)MAP_FAILED || mprotect(m, page, PROT_NONE) != 0 || mprotect(m + page + body, page, PROT_NONE) != 0) return 1;
    char* start = m + page + body - (n + 15) / 16 * 16;
    for (size_t i = 0; i + 4 <= n; i += 4) { const unsigned nan = 0x7fc00000u; memcpy(start + i, &nan, 4); }
    *p = (float*)start;
    return 0;
}
static inline hipError_t hipFree(void*) { return 0; }
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
extern "C" void hipemu_add_coresident_kernel(const char* substr);   // kernels with grid-wide barriers: one OS thread per workgroup
extern "C" void hipemu_set_wave_schedule(int mode, unsigned seed);  // order of a workgroup's waves between synchronisation points
extern "C" void hipemu_set_threads(int n);                          // OS threads the workgroups of a launch are spread over
extern "C" const char* hipemu_last_message();
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
// per-launch time of `n` back-to-back launches on one stream (what a dependent launch costs inside a replayed graph), min of 5 trains
static float train_us(const std::function<void()>& f, int n = 20) {
#ifdef ABI_CHECK_HOST
    n = 1;   // times mean nothing on the execution model: one call keeps the launch paths exercised
#endif
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    if (REPS > 1) f();   // warm-up launch (hardware only)
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < (REPS > 1 ? 5 : 1); ++rep) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < n; ++i) f();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms * 1e3f / n);
    }
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return best;
}
static int failures = 0;
// FNV-1a over the bit patterns: equal digests = bit-identical results (the execution-model runs compare them across wave schedules)
static unsigned long long digest(const std::vector<float>& v, unsigned long long h = 1469598103934665603ull) {
    for (float f : v) { unsigned u; memcpy(&u, &f, 4); for (int b = 0; b < 4; ++b) { h ^= (u >> (8 * b)) & 0xff; h *= 1099511628211ull; } }
    return h;
}
static void report(const char* what, const char* kernel, double r, double tol, float us_new, float us_old, double mbytes = 0) {
    const bool ok = r <= tol;
    if (!ok) ++failures;
    printf("%-34s %-18s rel %.2e (<= %.0e) %-4s  forward/new %8.1f us   %8.1f us", what, kernel, r, tol, ok ? "ok" : "FAIL", us_new, us_old);
    if (mbytes > 0) printf("   %.0f GB/s of %.1f MB", mbytes / us_new * 1e3, mbytes);
    printf("\n");
    fflush(stdout);
}

// ---- one-launch InstanceNorm for small tensors against a host fp64 evaluation of nn.InstanceNorm2d -> LeakyReLU -> Dropout mask ----
static void norm_small_case(const char* name, int G, int P, int C, int act, bool use_mask) {
    if (migan_norm_small_ok(G, P, C) != 1) {
        printf("%-34s not a small-norm shape\n", name);
        return;
    }
    const size_t n = (size_t)G * P * C;
    Buf x(n), dy(n), mask(n), y(n, 0, false), mean((size_t)G * C, 0, false), invstd((size_t)G * C, 0, false), dx(n, 0, false);
    std::vector<float> hx = x.host(), hdy = dy.host(), hm = mask.host();
    for (auto& v : hm) v = v > 0.f ? 2.f : 0.f;   // Dropout(0.5) scaled by 1/(1-p)
    CK(hipMemcpy(mask.d, hm.data(), n * 4, hipMemcpyHostToDevice));
    const float slope = 0.2f, eps = 1e-5f;
    float t_f = time_us([&] {
        RC(migan_norm_fwd_small(x.d, y.d, mean.d, invstd.d, nullptr, nullptr, nullptr, use_mask ? mask.d : nullptr, G, P, C, act, slope, eps, nullptr));
    });
    float t_b = time_us([&] {
        RC(migan_norm_bwd_small(x.d, dy.d, use_mask ? mask.d : nullptr, mean.d, invstd.d, nullptr, nullptr, dx.d, G, P, C, act, slope, nullptr, nullptr));
    });
    std::vector<float> ry(n), rdx(n);
    for (int g = 0; g < G; ++g)
        for (int c = 0; c < C; ++c) {
            double m = 0, v = 0;
            for (int p = 0; p < P; ++p) m += hx[((size_t)g * P + p) * C + c];
            m /= P;
            for (int p = 0; p < P; ++p) { double d = hx[((size_t)g * P + p) * C + c] - m; v += d * d; }
            const double is = 1.0 / std::sqrt(v / P + eps);
            double s0 = 0, s1 = 0;
            std::vector<double> dz(P), xh(P);
            for (int p = 0; p < P; ++p) {
                const size_t i = ((size_t)g * P + p) * C + c;
                xh[p] = (hx[i] - m) * is;
                const double a = act == 1 ? (xh[p] > 0 ? xh[p] : xh[p] * slope) : act == 2 ? (xh[p] > 0 ? xh[p] : 0.0) : xh[p];
                const double da = act == 1 ? (xh[p] > 0 ? 1.0 : slope) : act == 2 ? (xh[p] > 0 ? 1.0 : 0.0) : 1.0;
                const double mk = use_mask ? hm[i] : 1.0;
                ry[i] = (float)(a * mk);
                dz[p] = hdy[i] * mk * da;
                s0 += dz[p];
                s1 += dz[p] * xh[p];
            }
            for (int p = 0; p < P; ++p) rdx[((size_t)g * P + p) * C + c] = (float)(is * (dz[p] - s0 / P - xh[p] * s1 / P));
        }
    report(name, "norm_small fwd", rel(y.host(), ry), 2e-6, t_f, 0.f);
    report(name, "norm_small bwd", rel(dx.host(), rdx), 2e-5, t_b, 0.f);
}

// ---- fused WGAN-GP kernels against host fp64, per-launch cost ----
static void critic_case(int B, int Din, int H1, int H2) {
    if (migan_critic_fused_ok(B, Din, H1, H2) != 1) { printf("critic_fused: shape not taken\n"); return; }
    Buf real((size_t)B * Din), fake((size_t)B * Din), alpha(B), w1((size_t)H1 * Din, 0.03f), b1(H1, 0.1f), w2((size_t)H2 * H1, 0.05f), b2(H2, 0.1f),
        w3(H2, 0.1f), b3(1, 0.1f);
    std::vector<float> ha = alpha.host();
    for (auto& v : ha) v = 0.5f * (v + 1.f);
    CK(hipMemcpy(alpha.d, ha.data(), B * 4, hipMemcpyHostToDevice));
    const size_t wsb = migan_critic_fused_workspace(B, Din, H1, H2);
    Buf ws(wsb / 4, 0, false);
    // host fp64: mean D(real), mean D(fake) (out[2], out[3])
    auto hr = real.host(), hf = fake.host(), hw1 = w1.host(), hb1 = b1.host(), hw2 = w2.host(), hb2 = b2.host(), hw3 = w3.host(), hb3 = b3.host();
    auto dmean = [&](const std::vector<float>& xin) {
        double tot = 0;
        std::vector<double> h1(H1), h2(H2);
        for (int r = 0; r < B; ++r) {
            for (int j = 0; j < H1; ++j) {
                double a = hb1[j];
                for (int k = 0; k < Din; ++k) a += (double)xin[(size_t)r * Din + k] * hw1[(size_t)j * Din + k];
                h1[j] = a > 0 ? a : 0.2 * a;
            }
            for (int j = 0; j < H2; ++j) {
                double a = hb2[j];
                for (int k = 0; k < H1; ++k) a += h1[k] * hw2[(size_t)j * H1 + k];
                h2[j] = a > 0 ? a : 0.2 * a;
            }
            double o = hb3[0];
            for (int k = 0; k < H2; ++k) o += h2[k] * hw3[k];
            tot += o;
        }
        return tot / B;
    };
    const double mr = dmean(hr), mf = dmean(hf);
    {
        Buf gw1((size_t)H1 * Din, 0.f), gb1(H1, 0.f), gw2((size_t)H2 * H1, 0.f), gb2(H2, 0.f), gw3(H2, 0.f), gb3(1, 0.f), out(4, 0.f);
        auto run = [&](int phase) {
            RC(migan_critic_fused(real.d, fake.d, alpha.d, w1.d, b1.d, w2.d, b2.d, w3.d, b3.d, gw1.d, gb1.d, gw2.d, gb2.d, gw3.d, gb3.d, out.d, ws.d,
                                  wsb, B, Din, H1, H2, 0.2f, 10.f, 0, phase, nullptr));
        };
        run(0);
        CK(hipDeviceSynchronize());
        std::vector<float> ho = out.host();
        const bool fin = std::isfinite(ho[0]) && std::isfinite(ho[1]);
        const double e = std::max(std::fabs(ho[2] - mr), std::fabs(ho[3] - mf)) / std::max(1.0, std::fabs(mr));
        const bool ok = fin && e <= 1e-4;
        if (!ok) ++failures;
        unsigned long long dg = digest(ho);
        for (Buf* gbuf : {&gw1, &gb1, &gw2, &gb2, &gw3, &gb3}) dg = digest(gbuf->host(), dg);
        printf("critic_fused gradients + losses digest %016llx\n", dg);
        printf("critic_fused B%d %d-%d-%d  d_loss %.6g gp %.6g  |mean D - fp64| %.1e  %s  single call between events %8.1f us\n", B, Din, H1, H2, ho[0],
               ho[1], e, ok ? "ok" : "FAIL", time_us([&] { run(0); }));
        printf("critic_fused, 20 calls back to back: %.1f us per call (6 launches)\n", train_us([&] { run(0); }));
        printf("critic_fused per launch (train of 20 of the same launch):");
        for (int ph = 1; ph <= 6; ++ph) printf("  p%d %.1f us", ph, train_us([&] { run(ph); }));
        printf("\n");
        fflush(stdout);
    }
}

static void mlp_case(int B) {
    const int L = 5;
    const int K[L] = {100, 128, 256, 512, 1024}, Nn[L] = {128, 256, 512, 1024, 1024}, bn[L] = {0, 1, 1, 1, 0}, actc[L] = {1, 1, 1, 1, 3};
    int dims[4 * L];
    float fpar[3 * L];
    for (int l = 0; l < L; ++l) {
        dims[4 * l] = K[l]; dims[4 * l + 1] = Nn[l]; dims[4 * l + 2] = bn[l]; dims[4 * l + 3] = actc[l];
        fpar[3 * l] = 0.2f; fpar[3 * l + 1] = 0.8f; fpar[3 * l + 2] = 0.1f;
    }
    if (migan_mlp_fused_ok(B, L, dims) != 1) { printf("mlp_fused: shape not taken\n"); return; }
    std::vector<Buf*> keep;
    void* ptrs[7 * L];
    std::vector<std::vector<float>> hW(L), hb(L), hg(L), hbe(L);
    long long* nbt;
    CK(hipMalloc((float**)&nbt, 8 * L));
    CK(hipMemset(nbt, 0, 8 * L));
    for (int l = 0; l < L; ++l) {
        Buf* W = new Buf((size_t)Nn[l] * K[l], 1.0f / std::sqrt((float)K[l]));
        Buf* b = new Buf(Nn[l], 0.1f);
        keep.push_back(W); keep.push_back(b);
        hW[l] = W->host(); hb[l] = b->host();
        ptrs[7 * l] = W->d; ptrs[7 * l + 1] = b->d;
        for (int q = 2; q < 7; ++q) ptrs[7 * l + q] = nullptr;
        if (bn[l]) {
            Buf* g = new Buf(Nn[l], 0.2f); Buf* be = new Buf(Nn[l], 0.1f); Buf* rm = new Buf(Nn[l], 0.f); Buf* rv = new Buf(Nn[l], 0.f);
            keep.insert(keep.end(), {g, be, rm, rv});
            hg[l] = g->host(); hbe[l] = be->host();
            for (auto& v : hg[l]) v += 1.f;
            CK(hipMemcpy(g->d, hg[l].data(), Nn[l] * 4, hipMemcpyHostToDevice));
            ptrs[7 * l + 2] = g->d; ptrs[7 * l + 3] = be->d; ptrs[7 * l + 4] = rm->d; ptrs[7 * l + 5] = rv->d; ptrs[7 * l + 6] = nbt + l;
        }
    }
    Buf x((size_t)B * K[0]);
    // host fp64 forward: Linear -> BatchNorm1d(train, eps 0.8) -> LeakyReLU(0.2) | Tanh
    std::vector<double> a(x.n);
    { auto hx = x.host(); for (size_t i = 0; i < hx.size(); ++i) a[i] = hx[i]; }
    for (int l = 0; l < L; ++l) {
        std::vector<double> o((size_t)B * Nn[l]);
        for (int r = 0; r < B; ++r)
            for (int j = 0; j < Nn[l]; ++j) {
                double s_ = hb[l][j];
                for (int k = 0; k < K[l]; ++k) s_ += a[(size_t)r * K[l] + k] * hW[l][(size_t)j * K[l] + k];
                o[(size_t)r * Nn[l] + j] = s_;
            }
        if (bn[l])
            for (int j = 0; j < Nn[l]; ++j) {
                double m = 0, v = 0;
                for (int r = 0; r < B; ++r) m += o[(size_t)r * Nn[l] + j];
                m /= B;
                for (int r = 0; r < B; ++r) { double d = o[(size_t)r * Nn[l] + j] - m; v += d * d; }
                const double is = 1.0 / std::sqrt(v / B + 0.8);
                for (int r = 0; r < B; ++r) o[(size_t)r * Nn[l] + j] = (o[(size_t)r * Nn[l] + j] - m) * is * hg[l][j] + hbe[l][j];
            }
        for (auto& v : o) v = actc[l] == 1 ? (v > 0 ? v : 0.2 * v) : std::tanh(v);
        a.swap(o);
    }
    std::vector<float> ref(a.begin(), a.end());
    const size_t wsb = migan_mlp_fused_workspace(B, L, dims, 0);
    Buf ws(wsb / 4, 0, false);
    unsigned* tickets;
    CK(hipMalloc((float**)&tickets, 4096));
    CK(hipMemset(tickets, 0, 4096));
    {
        Buf y((size_t)B * Nn[L - 1], 0, false);
        auto run = [&] { RC(migan_mlp_fused_fwd(x.d, y.d, B, L, dims, fpar, ptrs, ws.d, wsb, 0, tickets, 0, nullptr)); };
        run();
        CK(hipDeviceSynchronize());
        const double r = rel(y.host(), ref);
        const bool ok = r <= 1e-5;
        if (!ok) ++failures;
        printf("mlp_fused_fwd B%d 100-128-256-512-1024-1024  rel vs fp64 %.2e  %s  digest %016llx  single call between events %8.1f us\n", B, r,
               ok ? "ok" : "FAIL", digest(y.host()), time_us(run));
        fflush(stdout);
    }
    {
        Buf y((size_t)B * Nn[L - 1], 0, false);
        printf("mlp_fused_fwd, 20 calls back to back: %.1f us per call (%d launches: layer 0 inside the launch of layer 1)\n",
               train_us([&] { RC(migan_mlp_fused_fwd(x.d, y.d, B, L, dims, fpar, ptrs, ws.d, wsb, 0, tickets, 0, nullptr)); }), L - 1);
        printf("mlp_fused_fwd per layer:");
        for (int l = 0; l < L; ++l) {
            const float us = train_us([&] { RC(migan_mlp_fused_fwd(x.d, y.d, B, L, dims, fpar, ptrs, ws.d, wsb, 0, tickets, 1 + l, nullptr)); });
            printf("  l%d (%d->%d) %.1f us", l, K[l], Nn[l], us);
        }
        printf("\n");
        fflush(stdout);
    }
    {   // the generator iteration's pair: forward that keeps its activations + backward (top, chain l = 4 .. 1, gradients), timed per phase
        const size_t sb = migan_mlp_fused_workspace(B, L, dims, 1), bb = migan_mlp_fused_bwd_workspace(B, L, dims);
        Buf save(sb / 4, 0, false), bws(bb / 4, 0, false), y((size_t)B * Nn[L - 1], 0, false), dy((size_t)B * Nn[L - 1], 0.01f);
        void* gptrs[4 * L];
        std::vector<Buf*> gk;
        for (int l = 0; l < L; ++l) {
            Buf* gW = new Buf((size_t)Nn[l] * K[l], 0, false); Buf* gb = new Buf(Nn[l], 0, false);
            gk.push_back(gW); gk.push_back(gb);
            gptrs[4 * l] = gW->d; gptrs[4 * l + 1] = gb->d; gptrs[4 * l + 2] = gptrs[4 * l + 3] = nullptr;
            if (bn[l]) {
                Buf* gg = new Buf(Nn[l], 0, false); Buf* gbe = new Buf(Nn[l], 0, false);
                gk.push_back(gg); gk.push_back(gbe);
                gptrs[4 * l + 2] = gg->d; gptrs[4 * l + 3] = gbe->d;
            }
        }
        RC(migan_mlp_fused_fwd(x.d, y.d, B, L, dims, fpar, ptrs, save.d, sb, 1, tickets, 0, nullptr));
        auto bwd = [&](int only) {
            RC(migan_mlp_fused_bwd(x.d, y.d, dy.d, save.d, nullptr, B, L, dims, fpar, ptrs, gptrs, bws.d, bb, 0, only, nullptr));
        };
        bwd(0);
        CK(hipDeviceSynchronize());
        bool fin = true;
        double asum = 0;
        unsigned long long dg = 1469598103934665603ull;
        for (Buf* g : gk) { const auto hv = g->host(); dg = digest(hv, dg); for (float v : hv) { fin = fin && std::isfinite(v); asum += std::fabs(v); } }
        if (!fin || asum == 0) ++failures;
        printf("mlp_fused_bwd B%d  gradients finite %s (sum |g| %.4g)  digest %016llx  20 calls back to back: %.1f us per call\n", B,
               fin ? "ok" : "FAIL", asum, dg, train_us([&] { bwd(0); }));
        printf("mlp_fused_bwd per phase:  top %.1f us", train_us([&] { bwd(1); }));
        for (int ph = 1; ph < L; ++ph) printf("  chain l%d (%d<-%d) %.1f us", L - ph, K[L - ph], Nn[L - ph], train_us([&] { bwd(1 + ph); }));
        printf("  gradients %.1f us\n", train_us([&] { bwd(L + 2); }));
        fflush(stdout);
        for (Buf* g : gk) delete g;
    }
    (void)hipFree(nbt);
    (void)hipFree(tickets);
    for (Buf* b : keep) delete b;
}

int main(int argc, char** argv) {
    int dev = 0;
#ifdef ABI_CHECK_HOST
    // HIPEMU_SCHED=fwd|rev|rand[:seed], HIPEMU_THREADS=n: the ticketed hand-offs and K-slice reductions must give the same numbers in
    // every wave order and workgroup arrival order (tests/test_kernels_emu_cpu.py runs the sections under several)
    if (const char* e = getenv("HIPEMU_SCHED")) {
        const std::string v = e;
        const size_t c = v.find(':');
        const std::string m = v.substr(0, c);
        hipemu_set_wave_schedule(m == "rev" ? 1 : m == "rand" ? 2 : 0, c == std::string::npos ? 1u : (unsigned)atoi(v.c_str() + c + 1));
    }
    if (const char* e = getenv("HIPEMU_THREADS")) hipemu_set_threads(atoi(e));
#endif
    CK(hipSetDevice(dev));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, dev));
    printf("%s | %s | %d CUs\n", migan_version(), p.name, p.multiProcessorCount);
    const std::string only = argc > 1 ? argv[1] : "";
    auto want = [&](const char* s) { return only.empty() || only == s; };
    if (want("norm")) {
        norm_small_case("IN 512ch 2x2 lrelu+mask", 1, 4, 512, 1, true);
        norm_small_case("IN 512ch 8x8 relu+mask", 1, 64, 512, 2, true);
        norm_small_case("IN 256ch 32x32 lrelu", 1, 1024, 256, 1, false);
        norm_small_case("IN 64ch 16x16 b8", 8, 256, 64, 0, false);
    }
    if (want("floor")) {   // what a dependent launch costs on this device: trains of trivial / small streaming launches of the library
        Buf a(1 << 22), y(1 << 22, 0, false);
        for (size_t n : {(size_t)1, (size_t)65536, (size_t)(1 << 20), (size_t)(1 << 22)})
            printf("launch floor: axpby n=%-8zu %.2f us per launch in a train of 20 (single launch between events: %.2f us)\n", n,
                   train_us([&] { RC(migan_axpby(a.d, 1.f, nullptr, 0.f, y.d, n, nullptr)); }),
                   time_us([&] { RC(migan_axpby(a.d, 1.f, nullptr, 0.f, y.d, n, nullptr)); }));
        fflush(stdout);
    }
    if (want("critic")) critic_case(64, 1024, 512, 256);
    if (want("mlp")) mlp_case(64);
    printf(failures ? "FAILED: %d comparison(s) out of bound\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
