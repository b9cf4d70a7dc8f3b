// Layers of the DCGAN-block clones: nn.Embedding (acgan.py:50, cgan/infogan label embeddings),
// nn.Softmax over class logits (acgan.py:100, sgan.py), nn.CrossEntropyLoss with integer targets (acgan.py:113,
// infogan.py categorical loss).  nn.BCEWithLogitsLoss (relativistic_gan.py:95) is kind 4 of migan_loss_fwd/bwd.
// Class counts are tiny (10 classes, batch 64): these kernels are latency-bound; what matters is that the path stays
// on the device (graph-capturable) and deterministic.
#include "common.h"

// y[i][:] = w[idx[i]][:]
__global__ void embedding_fwd_kernel(const float* __restrict__ w, const long long* __restrict__ idx,
                                     float* __restrict__ y, int n, int D, int V) {
    const size_t total = (size_t)n * D;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / D), d = (int)(e - (size_t)i * D);
        long long v = idx[i];
        v = v < 0 ? 0 : (v >= V ? V - 1 : v);  // memory safety only (torch device-asserts; a host check would cost a sync per call): forward and backward clamp alike
        y[e] = w[(size_t)v * D + d];
    }
}
// dw[v][d] (+)= sum over i with idx[i] == v of dy[i][d], i ascending (deterministic; aten uses atomics or a sort)
__global__ void embedding_bwd_kernel(const float* __restrict__ dy, const long long* __restrict__ idx,
                                     float* __restrict__ dw, int n, int D, int V, int accum) {
    const size_t total = (size_t)V * D;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(e / D), d = (int)(e - (size_t)v * D);
        float acc = 0.f;
        for (int i = 0; i < n; ++i)
            if (idx[i] == v) acc += dy[(size_t)i * D + d];
        dw[e] = accum ? dw[e] + acc : acc;
    }
}
MIGAN_API int migan_embedding_fwd(const float* w, const long long* idx, float* y, int n, int D, int V, void* stream) {
    if ((size_t)n * D == 0) return 0;
    int blocks = cdiv((long)n * D, 256);
    if (blocks > 2048) blocks = 2048;
    MIGAN_LAUNCH(embedding_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, idx, y, n, D, V);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_embedding_bwd(const float* dy, const long long* idx, float* dw, int n, int D, int V, int accumulate,
                                  void* stream) {
    if ((size_t)V * D == 0) return 0;
    int blocks = cdiv((long)V * D, 256);
    if (blocks > 2048) blocks = 2048;
    MIGAN_LAUNCH(embedding_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, idx, dw, n, D, V,
                       accumulate);
    HIP_LAUNCH_CHECK();
    return 0;
}

// softmax over the last dim of [B][C]: one wave per row (C is the class count)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                          int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* xr = x + (size_t)row * C;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(xr[c] - m);
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int c = lane; c < C; c += 64) y[(size_t)row * C + c] = expf(xr[c] - m) * inv;
}
// dx = y * (dy - sum_c dy*y)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int B, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* yr = y + (size_t)row * C;
    const float* gr = dy + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += gr[c] * yr[c];
    s = wave_sum(s);
    for (int c = lane; c < C; c += 64) dx[(size_t)row * C + c] = yr[c] * (gr[c] - s);
}
MIGAN_API int migan_softmax_fwd(const float* x, float* y, int B, int C, void* stream) {
    if ((size_t)B * C == 0) return 0;
    MIGAN_LAUNCH(softmax_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, x, y, B, C);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_softmax_bwd(const float* y, const float* dy, float* dx, int B, int C, void* stream) {
    if ((size_t)B * C == 0) return 0;
    MIGAN_LAUNCH(softmax_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, B, C);
    HIP_LAUNCH_CHECK();
    return 0;
}

// CrossEntropyLoss(reduction='mean') on logits [B][C] with int64 class targets: mean_i (logsumexp(x_i) - x_i[t_i]).
// Pass 1: one wave per row -> rowloss[i] (and the row's logsumexp in lse[i], kept for backward); pass 2: one block sums
// the rows in double.  Backward: dx[i][c] = g/B * (exp(x[i][c] - lse[i]) - [c == t_i]).
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ x, const long long* __restrict__ t,
                                                      float* __restrict__ rowloss, float* __restrict__ lse, int B,
                                                      int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* xr = x + (size_t)row * C;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(xr[c] - m);
    s = wave_sum(s);
    if (lane == 0) {
        const float l = m + logf(s);
        long long tt = t[row];
        tt = tt < 0 ? 0 : (tt >= C ? C - 1 : tt);
        lse[row] = l;
        rowloss[row] = l - xr[tt];
    }
}
__global__ void ce_mean_kernel(const float* __restrict__ rowloss, float* __restrict__ out, int B) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) acc += (double)rowloss[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)B);
}
__global__ void ce_bwd_kernel(const float* __restrict__ x, const long long* __restrict__ t,
                              const float* __restrict__ lse, const float* __restrict__ g, float* __restrict__ dx, int B,
                              int C) {
    const float gs = g[0] / (float)B;
    const size_t total = (size_t)B * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / C), c = (int)(e - (size_t)i * C);
        const float p = expf(x[e] - lse[i]);
        long long tt = t[i];
        tt = tt < 0 ? 0 : (tt >= C ? C - 1 : tt);  // the same clamp as the forward: loss and gradient stay consistent
        dx[e] = gs * (p - (tt == c ? 1.f : 0.f));
    }
}
// ws: 2*B floats (row losses, row logsumexp); lse = ws + B is what migan_cross_entropy_bwd takes
MIGAN_API int migan_cross_entropy_fwd(const float* x, const long long* target, float* out, float* ws, int B, int C,
                                      void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (B <= 0 || C <= 0) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(ce_rows_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, x, target, ws, ws + B, B, C);
    HIP_LAUNCH_CHECK();
    MIGAN_LAUNCH(ce_mean_kernel, dim3(1), dim3(256), 0, st, ws, out, B);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_cross_entropy_bwd(const float* x, const long long* target, const float* lse, const float* g,
                                      float* dx, int B, int C, void* stream) {
    if ((size_t)B * C == 0) return 0;
    int blocks = cdiv((long)B * C, 256);
    if (blocks > 2048) blocks = 2048;
    MIGAN_LAUNCH(ce_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, target, lse, g, dx, B, C);
    HIP_LAUNCH_CHECK();
    return 0;
}
