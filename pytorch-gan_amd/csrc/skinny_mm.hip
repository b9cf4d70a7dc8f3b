// Skinny GEMMs for the MLP critic / generator of wgan_gp.py:42-83 and gan.py:38-81 at the reference batch size (64 rows):
// Linear forward  C[M][N] = act(A[M][K] W[N][K]^T + b)   and   Linear input gradient  C[M][K] = G[M][N] W[N][K].
// With M = 64 the tiled implicit-GEMM kernels run 8-16 workgroups that each walk the whole K serially (23 us per
// launch, 467 us of a 0.92 ms critic iteration); these shapes are 67 MFLOP - latency, not throughput.  Here:
//   * no LDS, no barriers: a wave owns a 16-row x 16(32)-column output tile and feeds v_mfma_f32_16x16x4_f32 straight
//     from 16-byte global loads (the operands are L2-resident: activations 256 KB, weights <= 4 MB);
//   * the k order inside a 16-deep chunk is permuted (MFMA step s takes k = 4*(lane>>4) + s from every lane's float4),
//     identically for both operands, so one float4 per lane feeds four MFMA steps;
//   * two independent accumulators per wave (16x16x4 has a 40-cycle dependent latency against a 32-cycle issue);
//   * N/16 (N/32) workgroups of 4 waves = 4 row groups: 32-64 workgroups instead of 8.
// The NN form reads W in its stored [N][K] layout, so the input gradient needs no transposed weight copy.
#include "common.h"

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// C[M][N] = act(A[M][K] * W[N][K]^T + bias);  M <= 64, N % 16 == 0, K % 16 == 0
__global__ __launch_bounds__(256) void skinny_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ C, int M,
                                                        int N, int K, int act, float slope) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rr = lane & 15, kq = lane >> 4;
    const int r0 = wave * 16, col0 = blockIdx.x * 16;
    if (r0 >= M) return;  // wave-uniform; no barriers in this kernel
    const int arow = r0 + rr < M ? r0 + rr : M - 1;
    const float* ap = A + (size_t)arow * K + kq * 4;
    const float* wp = W + (size_t)(col0 + rr) * K + kq * 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int k0 = 0;
    for (; k0 + 64 <= K; k0 += 64) {  // 8 independent 16-byte loads in flight per lane, then 16 MFMAs
        f32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(ap + k0 + 16 * u);
            b[u] = *reinterpret_cast<const f32x4*>(wp + k0 + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; u += 2)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc0 = mfma16(a[u][s], b[u][s], acc0);
                acc1 = mfma16(a[u + 1][s], b[u + 1][s], acc1);
            }
    }
    for (; k0 < K; k0 += 16) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + k0), b0 = *reinterpret_cast<const f32x4*>(wp + k0);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc0 = mfma16(a0[s], b0[s], acc0);
    }
    const int col = col0 + rr;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + kq * 4 + r;
        if (row < M) C[(size_t)row * N + col] = act_apply(acc0[r] + acc1[r] + bv, act, slope);
    }
}

// C[M][Nc] = A[M][R] * W[R][Nc];  M <= 64, R % 16 == 0, Nc % 32 == 0.  Tile j of a wave holds columns col0 + 2*(lane&15) + j.
__global__ __launch_bounds__(256) void skinny_nn_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                        float* __restrict__ C, int M, int R, int Nc) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cc = lane & 15, kq = lane >> 4;
    const int r0 = wave * 16, col0 = blockIdx.x * 32;
    if (r0 >= M) return;
    const int arow = r0 + cc < M ? r0 + cc : M - 1;
    const float* ap = A + (size_t)arow * R + kq * 4;
    const float* wp = W + (size_t)(kq * 4) * Nc + col0 + 2 * cc;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int n0 = 0;
    for (; n0 + 32 <= R; n0 += 32) {  // 2 + 8 independent loads in flight per lane, then 16 MFMAs
        f32x4 a[2];
        f32x2 b[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(ap + n0 + 16 * u);
#pragma unroll
            for (int s = 0; s < 4; ++s) b[u][s] = *reinterpret_cast<const f32x2*>(wp + (size_t)(n0 + 16 * u + s) * Nc);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc0 = mfma16(a[u][s], b[u][s][0], acc0);
                acc1 = mfma16(a[u][s], b[u][s][1], acc1);
            }
    }
    for (; n0 < R; n0 += 16) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ap + n0);
        f32x2 b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) b[s] = *reinterpret_cast<const f32x2*>(wp + (size_t)(n0 + s) * Nc);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc0 = mfma16(a[s], b[s][0], acc0);
            acc1 = mfma16(a[s], b[s][1], acc1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + kq * 4 + r;
        if (row < M) {
            f32x2 o = {acc0[r], acc1[r]};
            *reinterpret_cast<f32x2*>(C + (size_t)row * Nc + col0 + 2 * cc) = o;
        }
    }
}

// 1 when the skinny kernels take the shape (else the caller uses migan_conv2d_fwd / migan_transpose_batched + migan_conv2d_fwd)
MIGAN_API int migan_skinny_nt_ok(int M, int N, int K) { return M >= 1 && M <= 64 && N % 16 == 0 && K % 16 == 0 && K >= 32; }
MIGAN_API int migan_skinny_nn_ok(int M, int R, int Nc) { return M >= 1 && M <= 64 && R % 16 == 0 && Nc % 32 == 0 && R >= 16; }

// nn.Linear forward for <= 64 rows (wgan_gp.py:46-56,73-77): y[M][N] = act(x[M][K] w[N][K]^T + bias)
MIGAN_API int migan_skinny_nt(const float* a, const float* w, const float* bias, float* c, int M, int N, int K, int act,
                              float slope, void* stream) {
    if (!migan_skinny_nt_ok(M, N, K)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(skinny_nt_kernel, dim3(N / 16), dim3(256), 0, (hipStream_t)stream, a, w, bias, c, M, N, K, act, slope);
    HIP_LAUNCH_CHECK();
    return 0;
}
// nn.Linear input gradient for <= 64 rows: dx[M][K] = dy[M][N] w[N][K]  (w in its stored layout; Nc = K, R = N)
MIGAN_API int migan_skinny_nn(const float* a, const float* w, float* c, int M, int R, int Nc, void* stream) {
    if (!migan_skinny_nn_ok(M, R, Nc)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(skinny_nn_kernel, dim3(Nc / 32), dim3(256), 0, (hipStream_t)stream, a, w, c, M, R, Nc);
    HIP_LAUNCH_CHECK();
    return 0;
}
