// Reductions, losses, gradient-penalty helpers and the fused multi-tensor Adam.
// Reference semantics:
//   torch.nn.BCELoss (dcgan.py:103; log clamped at -100, mean), BCEWithLogitsLoss (relativistic_gan.py:95), MSELoss / L1Loss
//   (cyclegan.py:57-59, pix2pix.py:50-51, srgan.py:71-72), torch.mean (wgan_gp.py:171,189),
//   gradients.norm(2, dim=1) and ((.-1)**2).mean() (wgan_gp.py:136-137),
//   torch.optim.Adam single-tensor arithmetic (dcgan.py:134-135).
#include "common.h"
#include <string.h>

#define REDUCE_BLOCKS 1024
static int grid_for(size_t nvec, int cap = 4096) {
    size_t b = (nvec + 255) / 256;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}
#define GRID_STRIDE(i, n)                                                        \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);      \
         i += (size_t)gridDim.x * blockDim.x)

__device__ __forceinline__ float block_sum(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    return red[0];
}
__global__ void final_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out, double scale) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * scale);
}

// ------------------------------------------------------------------ column sum: x[P][C] -> out[C]  (bias grads)
__global__ void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part, size_t P, int C,
                                      size_t chunk) {
    __shared__ float red[256];
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    int c = blockIdx.x * 64 + tx;
    size_t p0 = blockIdx.y * chunk, p1 = p0 + chunk;
    if (p1 > P) p1 = P;
    float acc = 0.f;
    if (c < C)
        for (size_t p = p0 + ty; p < p1; p += 4) acc += x[p * C + c];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (ty == 0 && c < C) part[(size_t)blockIdx.y * C + c] = red[tx] + red[64 + tx] + red[128 + tx] + red[192 + tx];
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           int C, int nchunks, int accum) {
    const int c = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;  // one wave per channel
    if (c >= C) return;
    double s = 0.0;
    for (int k = lane; k < nchunks; k += 8 * 64) {   // eight chunks per round of loads (clamped index, guarded add: same order, same sum)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = k + u * 64;
            v[u] = part[(size_t)(kk < nchunks ? kk : nchunks - 1) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k + u * 64 < nchunks) s += (double)v[u];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[c] = accum ? out[c] + (float)s : (float)s;
}
static void colsum_plan(size_t P, int C, size_t& chunk, int& nchunks) {
    long gx = cdiv(C, 64);
    long want = cdiv(1024, gx);
    long maxc = cdiv((long)P, 16);
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    chunk = (P + want - 1) / want;
    if (chunk < 1) chunk = 1;
    nchunks = (int)((P + chunk - 1) / chunk);
    if (nchunks < 1) nchunks = 1;
}
MIGAN_API size_t migan_colsum_workspace(size_t P, int C) {
    size_t chunk; int nchunks;
    colsum_plan(P, C, chunk, nchunks);
    return (size_t)nchunks * C * sizeof(float);
}
MIGAN_API int migan_colsum(const float* x, float* out, size_t P, int C, float* ws, size_t ws_bytes, int accumulate, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    size_t chunk; int nchunks;
    colsum_plan(P, C, chunk, nchunks);
    if (ws_bytes < (size_t)nchunks * C * sizeof(float)) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(colsum_partial_kernel, dim3(cdiv(C, 64), nchunks), dim3(256), 0, st, x, ws, P, C, chunk);
    HIP_LAUNCH_CHECK();
    MIGAN_LAUNCH(colsum_final_kernel, dim3(cdiv((long)C * 64, 256)), dim3(256), 0, st, ws, out, C, nchunks,
                       accumulate);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ losses (mean reduction)
enum { LOSS_BCE = 0, LOSS_MSE = 1, LOSS_L1 = 2, LOSS_MEAN = 3, LOSS_BCE_LOGITS = 4 };
__device__ __forceinline__ float loss_term(int kind, float x, float t) {
    switch (kind) {
        case LOSS_BCE: {
            float lx = fmaxf(logf(x), -100.f), l1x = fmaxf(logf(1.f - x), -100.f);
            return -(t * lx + (1.f - t) * l1x);
        }
        case LOSS_MSE: { float d = x - t; return d * d; }
        case LOSS_L1: return fabsf(x - t);
        case LOSS_BCE_LOGITS:  // torch: (1 - t) * x - log_sigmoid(x), log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))
            return (1.f - t) * x - (fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
        default: return x;
    }
}
__device__ __forceinline__ float loss_grad(int kind, float x, float t) {
    switch (kind) {
        case LOSS_BCE: {
            // torch: (x - t) / max((1-x)*x, 1e-12)
            return (x - t) / fmaxf((1.f - x) * x, 1e-12f);
        }
        case LOSS_MSE: return 2.f * (x - t);
        case LOSS_L1: { float d = x - t; return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
        case LOSS_BCE_LOGITS: return 1.f / (1.f + expf(-x)) - t;  // sigmoid(x) - t
        default: return 1.f;
    }
}
__global__ void loss_partial_kernel(int kind, const float* __restrict__ x, const float* __restrict__ t,
                                    float tconst, float* __restrict__ part, size_t n) {
    __shared__ float red[256];
    float acc = 0.f;
    GRID_STRIDE(i, n) acc += loss_term(kind, x[i], t ? t[i] : tconst);
    float s = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// n <= 16384 (the (B,1) validity vectors of dcgan.py:165, PatchGAN maps): one block does the whole mean
__global__ void loss_small_kernel(int kind, const float* __restrict__ x, const float* __restrict__ t, float tconst,
                                  float* __restrict__ out, int n) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)loss_term(kind, x[i], t ? t[i] : tconst);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)n);
}
// dx = g[0] * scale * dloss/dx
__global__ void loss_bwd_kernel(int kind, const float* __restrict__ x, const float* __restrict__ t, float tconst,
                                const float* __restrict__ g, float scale, float* __restrict__ dx, size_t n) {
    const float gs = g[0] * scale;
    GRID_STRIDE(i, n) dx[i] = gs * loss_grad(kind, x[i], t ? t[i] : tconst);
}
MIGAN_API int migan_loss_fwd(int kind, const float* x, const float* t, float tconst, float* out, size_t n, float* ws,
                             size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0 || ws_bytes < REDUCE_BLOCKS * sizeof(float)) return (int)hipErrorInvalidValue;
    if (n <= 16384) {
        MIGAN_LAUNCH(loss_small_kernel, dim3(1), dim3(256), 0, st, kind, x, t, tconst, out, (int)n);
        HIP_LAUNCH_CHECK();
        return 0;
    }
    int blocks = grid_for(n, REDUCE_BLOCKS);
    MIGAN_LAUNCH(loss_partial_kernel, dim3(blocks), dim3(256), 0, st, kind, x, t, tconst, ws, n);
    HIP_LAUNCH_CHECK();
    MIGAN_LAUNCH(final_sum_kernel, dim3(1), dim3(256), 0, st, ws, blocks, out, 1.0 / (double)n);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_loss_bwd(int kind, const float* x, const float* t, float tconst, const float* g, float* dx,
                             size_t n, void* stream) {
    if (n == 0) return 0;
    MIGAN_LAUNCH(loss_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, kind, x, t, tconst, g,
                       (float)(1.0 / (double)n), dx, n);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ row L2 norm (gradient penalty)
// out[b] = ||x[b,:]||_2 ; one block per row
__global__ void rownorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int D) {
    __shared__ float red[256];
    const float* r = x + (size_t)blockIdx.x * D;
    float acc = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) acc += r[i] * r[i];
    float s = block_sum(acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = sqrtf(s);
}
// dx[b,:] = x[b,:] * (dn[b] / n[b])     (torch: zero where n == 0)
__global__ void rownorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ nrm,
                                   const float* __restrict__ dn, float* __restrict__ dx, int D) {
    const size_t b = blockIdx.x;
    float n = nrm[b];
    float sc = n > 0.f ? dn[b] / n : 0.f;
    for (int i = threadIdx.x; i < D; i += 256) dx[b * D + i] = x[b * D + i] * sc;
}
MIGAN_API int migan_rownorm_fwd(const float* x, float* out, int B, int D, void* stream) {
    if (B == 0) return 0;
    MIGAN_LAUNCH(rownorm_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, out, D);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_rownorm_bwd(const float* x, const float* nrm, const float* dn, float* dx, int B, int D,
                                void* stream) {
    if (B == 0) return 0;
    MIGAN_LAUNCH(rownorm_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, nrm, dn, dx, D);
    HIP_LAUNCH_CHECK();
    return 0;
}
// y[b,:] = x[b,:]*s[b]  (WGAN-GP interpolation of detached samples, wgan_gp.py:125)
__global__ void rowscale_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y,
                                int D) {
    const size_t b = blockIdx.x;
    float sc = s[b];
    for (int i = threadIdx.x; i < D; i += 256) y[b * D + i] = x[b * D + i] * sc;
}
// ebgan.py:142-148 pullaway_loss(embeddings [B][D]) = (sum_ij <n_i, n_j> - B) / (B (B - 1)), n_i = e_i / |e_i| - the mean
// off-diagonal cosine similarity of the batch.  sum_ij <n_i, n_j> = |sum_i n_i|^2, so one workgroup suffices: forward
// leaves s[D] = sum_i n_i and inv[B] = 1 / |e_i| in ws (D + B floats) for the backward:
//   d loss / d e_k = g * 2 / (B (B-1)) * inv_k * (s - n_k <n_k, s>).
__global__ __launch_bounds__(256) void pullaway_fwd_kernel(const float* __restrict__ e, float* __restrict__ loss,
                                                           float* __restrict__ ws, int B, int D) {
    float* s = ws;
    float* inv = ws + D;
    __shared__ float red[256];
    for (int t = threadIdx.x; t < B; t += 256) {
        float q = 0.f;
        for (int d = 0; d < D; ++d) q += e[(size_t)t * D + d] * e[(size_t)t * D + d];
        inv[t] = 1.f / sqrtf(q);
    }
    __syncthreads();
    float part = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float a = 0.f;
        for (int t = 0; t < B; ++t) a += e[(size_t)t * D + d] * inv[t];
        s[d] = a;
        part += a * a;
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (red[0] - (float)B) / ((float)B * (float)(B - 1));
}
__global__ __launch_bounds__(256) void pullaway_bwd_kernel(const float* __restrict__ e, const float* __restrict__ ws,
                                                           const float* __restrict__ g, float* __restrict__ de, int B, int D) {
    const float* s = ws;
    const float* inv = ws + D;
    const float c = g[0] * 2.f / ((float)B * (float)(B - 1));
    for (int k = blockIdx.x * 256 + threadIdx.x; k < B; k += gridDim.x * 256) {
        const float iv = inv[k];
        float dot = 0.f;
        for (int d = 0; d < D; ++d) dot += e[(size_t)k * D + d] * iv * s[d];
        for (int d = 0; d < D; ++d) de[(size_t)k * D + d] = c * iv * (s[d] - e[(size_t)k * D + d] * iv * dot);
    }
}
MIGAN_API int migan_pullaway_fwd(const float* e, float* loss, float* ws, int B, int D, void* stream) {
    if (B < 2 || D < 1) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(pullaway_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, e, loss, ws, B, D);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_pullaway_bwd(const float* e, const float* ws, const float* g, float* de, int B, int D, void* stream) {
    MIGAN_LAUNCH(pullaway_bwd_kernel, dim3(cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, e, ws, g, de, B, D);
    HIP_LAUNCH_CHECK();
    return 0;
}

MIGAN_API int migan_rowscale(const float* x, const float* s, float* y, int B, int D, void* stream) {
    if (B == 0) return 0;
    MIGAN_LAUNCH(rowscale_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, s, y, D);
    HIP_LAUNCH_CHECK();
    return 0;
}
// ------------------------------------------------------------------ fused multi-tensor Adam
// One launch updates every tensor of an optimizer.  `tab` holds per-tensor {param, grad, exp_avg,
// exp_avg_sq, numel}; `blk` maps each block to (tensor, chunk).  `step` lives on the device so a captured
// hipGraph advances it on replay.  Arithmetic follows torch/optim/adam.py::_single_tensor_adam:
//   m.lerp_(g, 1-b1); v = v*b2 + (1-b2)*g*g; denom = sqrt(v)/sqrt(1-b2^t) + eps; p += (-lr/(1-b1^t)) * m/denom
struct AdamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
};
struct AdamBlock {
    int tensor;
    int chunk;
};
#define ADAM_CHUNK 4096
__global__ __launch_bounds__(256) void adam_kernel(const AdamTensor* __restrict__ tab,
                                                   const AdamBlock* __restrict__ blk, float* __restrict__ step,
                                                   unsigned* __restrict__ ticket, const float* __restrict__ lr_dev,
                                                   float lr_host, float b1, float b2, float eps, float grad_scale) {
    const AdamBlock bi = blk[blockIdx.x];
    const AdamTensor t = tab[bi.tensor];
    // Every block reads the OLD counter and computes with old + 1; the last block to finish publishes old + 1 (no
    // block reads `step` after its first instruction, and the next launch is a kernel boundary away), so the step
    // counter needs no launch of its own and still advances under hipGraph replay.
    const float st_f = step[0] + 1.f;
    const double st = (double)st_f;
    const float lr = lr_dev ? lr_dev[0] : lr_host;  // device scalar: a captured graph follows LambdaLR (cyclegan.py:275-277)
    const double bc1 = 1.0 - pow((double)b1, st);
    const double bc2 = 1.0 - pow((double)b2, st);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float w = 1.f - b1;
    long long i0 = (long long)bi.chunk * ADAM_CHUNK;
    long long i1 = i0 + ADAM_CHUNK;
    if (i1 > t.n) i1 = t.n;
    auto update = [&](float g, float& m, float& v, float& pv) {
        g *= grad_scale;
        // torch lerp: weight < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
        const float d = g - m;
        m = (w < 0.5f) ? m + w * d : g - d * (1.f - w);
        v = v * b2 + (1.f - b2) * g * g;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        pv = pv + (-step_size) * (m / denom);
    };
    // 16 bytes per lane and array; the chunk's 4 rounds x 4 arrays are all issued before the first update (a chunk used to be 16 dependent
    // rounds of 4-byte loads: 12 us for the 0.66 M parameters of the WGAN-GP critic, profiles/r04_wgan_gp_graph_kernel_stats.txt).
    // Slots of the bucket are 256 B aligned (optim.bucket_layout) and parameters are torch allocations: i0 is a multiple of 4096.
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.m) |
                       reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    if (vec && i1 - i0 == ADAM_CHUNK) {
        constexpr int R = ADAM_CHUNK / 1024;
        f32x4 g4[R], m4[R], v4[R], p4[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long i = i0 + r * 1024 + 4 * threadIdx.x;
            g4[r] = *reinterpret_cast<const f32x4*>(t.g + i);
            m4[r] = *reinterpret_cast<const f32x4*>(t.m + i);
            v4[r] = *reinterpret_cast<const f32x4*>(t.v + i);
            p4[r] = *reinterpret_cast<const f32x4*>(t.p + i);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long i = i0 + r * 1024 + 4 * threadIdx.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m = m4[r][e], v = v4[r][e], pv = p4[r][e];
                update(g4[r][e], m, v, pv);
                m4[r][e] = m;
                v4[r][e] = v;
                p4[r][e] = pv;
            }
            *reinterpret_cast<f32x4*>(t.p + i) = p4[r];
            *reinterpret_cast<f32x4*>(t.m + i) = m4[r];
            *reinterpret_cast<f32x4*>(t.v + i) = v4[r];
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
            float m = t.m[i], v = t.v[i], pv = t.p[i];
            update(t.g[i], m, v, pv);
            t.p[i] = pv;
            t.m[i] = m;
            t.v[i] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned arrived = atomicAdd(ticket, 1u);
        if (arrived == gridDim.x - 1) {
            step[0] = st_f;
            ticket[0] = 0u;
        }
    }
}
__global__ void step_inc_kernel(float* step) { step[0] += 1.f; }

MIGAN_API int migan_adam_chunk() { return ADAM_CHUNK; }
// `step`: device float (torch keeps Adam's step as an fp32 tensor too), `ticket`: device uint32 (zero-initialised),
// `lr_dev`: optional device float overriding `lr` (so a captured hipGraph sees learning-rate schedules).
MIGAN_API int migan_adam_step(const void* tab, const void* blk, int nblocks, float* step, unsigned* ticket,
                              const float* lr_dev, float lr, float b1, float b2, float eps, float grad_scale,
                              void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (nblocks > 0 && ticket) {
        MIGAN_LAUNCH(adam_kernel, dim3(nblocks), dim3(256), 0, st, (const AdamTensor*)tab,
                           (const AdamBlock*)blk, step, ticket, lr_dev, lr, b1, b2, eps, grad_scale);
        HIP_LAUNCH_CHECK();
    } else {
        if (nblocks > 0) return (int)hipErrorInvalidValue;
        MIGAN_LAUNCH(step_inc_kernel, dim3(1), dim3(1), 0, st, step);
        HIP_LAUNCH_CHECK();
    }
    return 0;
}

MIGAN_API const char* migan_version() { return "migan 0.1 gfx950"; }

// ---- debug launch counters (common.h) -----------------------------------------------------------------------------------
// Launches issued by this library since the last reset, summed over the launch sites whose kernel expression contains `substr`
// (NULL or "" = all sites).  Host-side bookkeeping only (one relaxed increment per launch).
MIGAN_API long migan_debug_launch_count(const char* substr) {
    long total = 0;
    for (migan_dbg::Site* s = migan_dbg::sites().load(std::memory_order_acquire); s; s = s->next)
        if (substr == nullptr || substr[0] == 0 || strstr(s->name, substr)) total += s->n.load(std::memory_order_relaxed);
    return total;
}
MIGAN_API void migan_debug_launch_reset() {
    for (migan_dbg::Site* s = migan_dbg::sites().load(std::memory_order_acquire); s; s = s->next) s->n.store(0, std::memory_order_relaxed);
}
MIGAN_API const char* migan_error_string(int code) { return hipGetErrorString((hipError_t)code); }
