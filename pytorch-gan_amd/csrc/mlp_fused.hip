// Forward of an MLP generator - Linear [-> BatchNorm1d (training mode)] [-> LeakyReLU | Tanh | ...] per layer - at <= 64 rows in ONE
// persistent launch: wgan_gp.py:42-65 / gan.py:38-61 (100 -> 128 -> 256 -> 512 -> 1024 -> prod(img_shape), BatchNorm1d(out, 0.8)
// on layers 2-4, LeakyReLU(0.2), Tanh), which the critic iterations of wgan_gp.py:163 run under no_grad: 5 Linear + 3 x (statistics,
// finalize, apply) = 14 launches of 3-6 us op by op.  One phase per layer, a grid-wide barrier between layers (the bounded-spin
// barrier of critic_fused.hip).  A workgroup owns 16 output columns and ALL rows, so the batch statistics of a BatchNorm1d column
// never leave the workgroup: its 8 waves are (row group) x (K slice), the K slices are combined through LDS in a fixed order, the
// column mean and the two-pass variance over the <= 64 rows are taken across the row-group waves through LDS, and the running
// statistics (momentum, unbiased variance) and num_batches_tracked are updated as nn.BatchNorm1d does in training mode.
#include "common.h"

#define MF_WAVES 8
#define MF_THREADS (64 * MF_WAVES)
#define MF_MAX_LAYERS 8
#define MF_SPIN_LIMIT (1u << 16)

struct MlpLayer {
    const float *W, *b, *gamma, *beta;
    float *rmean, *rvar;
    long long* nbt;
    int K, N, bn, act;
    float slope, eps, momentum;
};
struct MlpFused {
    int B, RB, nlayers, maxN;
    const float* x;
    float* y;
    float* ws;        // two [RB][maxN] activation buffers
    unsigned* sync;   // [0] arrivals, [1] exits, [2] error flag
    MlpLayer L[MF_MAX_LAYERS];
};

__device__ __forceinline__ f32x4 mf_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// K slice [kbeg, kend) of a 16x16 NT tile; ap / wp point at this lane's row at k = 0; K % 4 == 0, so a lane's float4 is inside or outside
__device__ __forceinline__ f32x4 mf_nt_partial(const float* __restrict__ ap, const float* __restrict__ wp, int kbeg, int kend, int kq) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int k0 = kbeg;
    for (; k0 + 64 <= kend; k0 += 64) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(ap + k0 + 16 * u + 4 * kq);
            b[u] = *reinterpret_cast<const f32x4*>(wp + k0 + 16 * u + 4 * kq);
        }
#pragma unroll
        for (int u = 0; u < 4; u += 2)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc0 = mf_mfma(a[u][s], b[u][s], acc0);
                acc1 = mf_mfma(a[u + 1][s], b[u + 1][s], acc1);
            }
    }
    for (; k0 < kend; k0 += 16) {
        const int k = k0 + 4 * kq;
        const bool in = k < kend;
        const int kc = in ? k : kbeg;  // any valid address
        f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + kc), b0 = *reinterpret_cast<const f32x4*>(wp + kc);
        if (!in) a0 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc0 = mf_mfma(a0[s], b0[s], acc0);
    }
    return acc0 + acc1;
}

__device__ __forceinline__ bool mf_grid_barrier(unsigned* sync, unsigned& target, int* give_up) {
    __syncthreads();
    if (threadIdx.x == 0) {
        target += gridDim.x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int bad = 0;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || ++spins > MF_SPIN_LIMIT) {
                __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *give_up = bad;
    }
    __syncthreads();
    return *give_up == 0;
}

__global__ __launch_bounds__(MF_THREADS) void mlp_fused_fwd_kernel(const MlpFused p) {
    __shared__ f32x4 part[(MF_WAVES - 1) * 64];   // partial tiles of the waves with ks > 0: [(ks - 1) * RGW + rg][lane]
    __shared__ float red[4][16];
    __shared__ int give_up;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rr = lane & 15, kq = lane >> 4;
    const int B = p.B, RB = p.RB;
    const int RG = RB / 16;
    const int RGW = RG >= 3 ? 4 : RG;             // row-group waves (1, 2 or 4); the rest of the 8 waves are K slices
    const int KSL = MF_WAVES / RGW;
    const int rg = wave % RGW, ks = wave / RGW;
    const bool rows_live = rg < RG;
    float* const buf0 = p.ws;
    float* const buf1 = p.ws + (size_t)RB * p.maxN;
    unsigned target = 0;
    if (threadIdx.x == 0) give_up = 0;

    for (int l = 0; l < p.nlayers; ++l) {
        const MlpLayer& Ly = p.L[l];
        const int K = Ly.K, N = Ly.N;
        const float* in = l == 0 ? p.x : ((l - 1) & 1 ? buf1 : buf0);
        float* out = l == p.nlayers - 1 ? p.y : (l & 1 ? buf1 : buf0);
        const int klen = ((K + KSL - 1) / KSL + 15) / 16 * 16;
        const int kbeg = ks * klen < K ? ks * klen : K, kend = (ks + 1) * klen < K ? (ks + 1) * klen : K;
        for (int t = blockIdx.x; t < N / 16; t += gridDim.x) {
            const int col0 = t * 16, col = col0 + rr;
            const int arow = rg * 16 + rr < B ? rg * 16 + rr : B - 1;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (rows_live && kbeg < kend) acc = mf_nt_partial(in + (size_t)arow * K, Ly.W + (size_t)col * K, kbeg, kend, kq);
            if (ks > 0) part[((ks - 1) * RGW + rg) * 64 + lane] = acc;
            __syncthreads();
            float v[4], d[4];
            bool ok[4];
            if (ks == 0) {
                for (int q = 1; q < KSL; ++q) acc += part[((q - 1) * RGW + rg) * 64 + lane];
                const float bv = Ly.b ? Ly.b[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ok[r] = rows_live && rg * 16 + kq * 4 + r < B;
                    v[r] = acc[r] + bv;
                }
            }
            float mean = 0.f, var = 0.f;
            if (Ly.bn) {  // layer-uniform: every thread takes the same barriers
                if (ks == 0) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) s += ok[r] ? v[r] : 0.f;
                    s += __shfl_xor(s, 16);
                    s += __shfl_xor(s, 32);
                    if (kq == 0) red[rg][rr] = s;
                }
                __syncthreads();
                if (ks == 0) {
                    float s = 0.f;
                    for (int q = 0; q < RGW; ++q) s += red[q][rr];
                    mean = s / (float)B;
                }
                __syncthreads();
                if (ks == 0) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        d[r] = v[r] - mean;
                        s += ok[r] ? d[r] * d[r] : 0.f;
                    }
                    s += __shfl_xor(s, 16);
                    s += __shfl_xor(s, 32);
                    if (kq == 0) red[rg][rr] = s;
                }
                __syncthreads();
                if (ks == 0) {
                    float s = 0.f;
                    for (int q = 0; q < RGW; ++q) s += red[q][rr];
                    var = s / (float)B;
                }
            }
            if (ks == 0) {
                float scale = 1.f, shift = 0.f;
                if (Ly.bn) {
                    const float invstd = 1.f / sqrtf(var + Ly.eps);
                    const float gm = Ly.gamma ? Ly.gamma[col] : 1.f, bt = Ly.beta ? Ly.beta[col] : 0.f;
                    scale = invstd * gm;
                    shift = bt;
                    if (rg == 0 && kq == 0 && Ly.rmean) {  // nn.BatchNorm1d, training: momentum update, unbiased variance
                        const float unb = B > 1 ? var * (float)B / (float)(B - 1) : var;
                        Ly.rmean[col] = (1.f - Ly.momentum) * Ly.rmean[col] + Ly.momentum * mean;
                        Ly.rvar[col] = (1.f - Ly.momentum) * Ly.rvar[col] + Ly.momentum * unb;
                    }
                }
                if (rows_live) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rg * 16 + kq * 4 + r;
                        const float pre = Ly.bn ? d[r] * scale + shift : v[r];
                        if (l == p.nlayers - 1) {
                            if (ok[r]) out[(size_t)row * N + col] = act_apply(pre, Ly.act, Ly.slope);
                        } else {
                            out[(size_t)row * N + col] = ok[r] ? act_apply(pre, Ly.act, Ly.slope) : 0.f;
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (Ly.bn && Ly.nbt && blockIdx.x == 0 && threadIdx.x == 0) *Ly.nbt += 1;
        if (l + 1 < p.nlayers && !mf_grid_barrier(p.sync, target, &give_up)) return;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(p.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// dims: K, N, has_bn, act per layer;  fpar: slope, eps, momentum per layer;  ptrs: W, b, gamma, beta, running_mean, running_var,
// num_batches_tracked per layer (device pointers in a HOST array; NULL where absent).  Returns 1 when the kernel takes the shape.
MIGAN_API int migan_mlp_fused_ok(int B, int nlayers, const int* dims) {
    if (B < 1 || B > 64 || nlayers < 1 || nlayers > MF_MAX_LAYERS) return 0;
    for (int l = 0; l < nlayers; ++l) {
        const int K = dims[4 * l], N = dims[4 * l + 1];
        if (K < 4 || K % 4 != 0 || N < 16 || N % 16 != 0) return 0;
        if (l > 0 && K != dims[4 * (l - 1) + 1]) return 0;
        if (dims[4 * l + 2] && B < 2) return 0;   // BatchNorm1d in training mode needs more than one row
    }
    return 1;
}
MIGAN_API size_t migan_mlp_fused_workspace(int B, int nlayers, const int* dims) {
    if (!migan_mlp_fused_ok(B, nlayers, dims)) return 0;
    int maxN = 0;
    for (int l = 0; l < nlayers; ++l) maxN = dims[4 * l + 1] > maxN ? dims[4 * l + 1] : maxN;
    return (size_t)2 * ((B + 15) / 16 * 16) * maxN * sizeof(float);
}
// y[B][N_last] = MLP(x[B][K_0]), BatchNorm1d layers in training mode (batch statistics; running statistics and counters updated).
// ws: migan_mlp_fused_workspace() bytes; sync: 4 unsigned ints zeroed once (sync[2] != 0 afterwards: the grid barrier gave up).
MIGAN_API int migan_mlp_fused_fwd(const float* x, float* y, int B, int nlayers, const int* dims, const float* fpar,
                                  void* const* ptrs, float* ws, size_t ws_bytes, unsigned* sync, int grid, void* stream) {
    if (!migan_mlp_fused_ok(B, nlayers, dims) || ws_bytes < migan_mlp_fused_workspace(B, nlayers, dims)) return (int)hipErrorInvalidValue;
    MlpFused p;
    p.B = B; p.RB = (B + 15) / 16 * 16; p.nlayers = nlayers; p.maxN = 0;
    p.x = x; p.y = y; p.ws = ws; p.sync = sync;
    int tiles = 0;
    for (int l = 0; l < nlayers; ++l) {
        MlpLayer& L = p.L[l];
        L.K = dims[4 * l]; L.N = dims[4 * l + 1]; L.bn = dims[4 * l + 2]; L.act = dims[4 * l + 3];
        L.slope = fpar[3 * l]; L.eps = fpar[3 * l + 1]; L.momentum = fpar[3 * l + 2];
        L.W = (const float*)ptrs[7 * l]; L.b = (const float*)ptrs[7 * l + 1];
        L.gamma = (const float*)ptrs[7 * l + 2]; L.beta = (const float*)ptrs[7 * l + 3];
        L.rmean = (float*)ptrs[7 * l + 4]; L.rvar = (float*)ptrs[7 * l + 5]; L.nbt = (long long*)ptrs[7 * l + 6];
        if (!L.W || (L.rmean == nullptr) != (L.rvar == nullptr)) return (int)hipErrorInvalidValue;
        p.maxN = L.N > p.maxN ? L.N : p.maxN;
        tiles = L.N / 16 > tiles ? L.N / 16 : tiles;
    }
    int g = grid > 0 ? grid : tiles;
    static const int grid_env = getenv("MIGAN_K7_GRID") ? atoi(getenv("MIGAN_K7_GRID")) : 0;
    if (grid <= 0 && grid_env > 0) g = grid_env;
    if (g > tiles) g = tiles;
    if (g > 128) g = 128;
    hipLaunchKernelGGL(mlp_fused_fwd_kernel, dim3(g), dim3(MF_THREADS), 0, (hipStream_t)stream, p);
    HIP_LAUNCH_CHECK();
    return 0;
}
