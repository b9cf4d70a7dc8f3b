// Skinny GEMMs for the MLP critic / generator of wgan_gp.py:42-83 and gan.py:38-81 at the reference batch size (64 rows):
// Linear forward  C[M][N] = act(A[M][K] W[N][K]^T + b)   and   Linear input gradient  C[M][K] = G[M][N] W[N][K].
// With M = 64 the tiled implicit-GEMM kernels run 8-16 workgroups that each walk the whole K serially (23 us per
// launch, 467 us of a 0.92 ms critic iteration); these shapes are 67 MFLOP - latency, not throughput.  Here:
//   * no LDS, no barriers: a wave owns a 16-row x 16(32)-column output tile and feeds v_mfma_f32_16x16x4_f32 straight
//     from 16-byte global loads (the operands are L2-resident: activations 256 KB, weights <= 4 MB);
//   * the k order inside a 16-deep chunk is permuted (MFMA step s takes k = 4*(lane>>4) + s from every lane's float4),
//     identically for both operands, so one float4 per lane feeds four MFMA steps;
//   * two independent accumulators per wave (16x16x4 has a 40-cycle dependent latency against a 32-cycle issue);
//   * a workgroup is ONE 16-row group x one 16 (32)-column tile, its waves are up to 8 K-slices: (N/16) x (M/16)
//     workgroups = 128-256 for the critic layers, so every CU pulls a share of the (L2-resident) operands - the second
//     version (4 row groups x 4 slices in one 1024-thread workgroup, N/16 = 32 workgroups) was bound by the L2 -> CU
//     bandwidth of the 32 CUs it ran on (13.4 us per launch, profiles/r02_wgan_kernel_stats.txt) - and the dependent
//     chain per wave is one or two load rounds.
// The NN form reads W in its stored [N][K] layout, so the input gradient needs no transposed weight copy.
#include "common.h"

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// The reduction dimension is cut into KS slices, one wave each (64 * KS threads): a wave's serial chain
// of dependent load -> MFMA rounds is what bounds these launches (first version, one slice: 16 rounds of ~0.8 us for
// K = 1024, 14.5 us per launch), and each round issues all its loads (128 k = 16 float4 per lane) before the MFMAs.
// The KS partial tiles are combined through LDS in a fixed order (deterministic).

// C[M][N] = act(A[M][K] * W[N][K]^T + bias);  M <= 64, N % 16 == 0, K % (16 * KS) == 0
template <int KS>
__global__ __launch_bounds__(64 * KS) void skinny_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ C, int M,
                                                             int N, int K, int act, float slope) {
    __shared__ f32x4 part[KS > 1 ? (KS - 1) * 64 : 1];
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int rg = blockIdx.y;
    const int rr = lane & 15, kq = lane >> 4;
    const int r0 = rg * 16, col0 = blockIdx.x * 16;
    const int Kslice = K / KS;
    const int arow = r0 + rr < M ? r0 + rr : M - 1;
    const float* ap = A + (size_t)arow * K + ks * Kslice + kq * 4;
    const float* wp = W + (size_t)(col0 + rr) * K + ks * Kslice + kq * 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int k0 = 0;
    for (; k0 + 128 <= Kslice; k0 += 128) {  // 16 independent 16-byte loads in flight per lane, then 32 MFMAs
        f32x4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(ap + k0 + 16 * u);
            b[u] = *reinterpret_cast<const f32x4*>(wp + k0 + 16 * u);
        }
        __builtin_amdgcn_sched_barrier(0);  // every load of the round is issued before its first MFMA
#pragma unroll
        for (int u = 0; u < 8; u += 2)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc0 = mfma16(a[u][s], b[u][s], acc0);
                acc1 = mfma16(a[u + 1][s], b[u + 1][s], acc1);
            }
    }
    for (; k0 + 64 <= Kslice; k0 += 64) {
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
    for (; k0 < Kslice; k0 += 16) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + k0), b0 = *reinterpret_cast<const f32x4*>(wp + k0);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc0 = mfma16(a0[s], b0[s], acc0);
    }
    f32x4 acc = acc0 + acc1;
    if (KS > 1) {
        if (ks > 0) part[(ks - 1) * 64 + lane] = acc;
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int q = 1; q < KS; ++q) acc += part[(q - 1) * 64 + lane];
    }
    const int col = col0 + rr;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + kq * 4 + r;
        if (row < M) C[(size_t)row * N + col] = act_apply(acc[r] + bv, act, slope);
    }
}

// C[M][Nc] = A[M][R] * W[R][Nc];  M <= 64, R % (16 * KS) == 0, Nc % 32 == 0.  Tile j of a wave holds columns col0 + 2*(lane&15) + j.
template <int KS>
__global__ __launch_bounds__(64 * KS) void skinny_nn_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                             float* __restrict__ C, int M, int R, int Nc) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ f32x4 part[KS > 1 ? 2 * (KS - 1) * 64 : 1];
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int rg = blockIdx.y;
    const int cc = lane & 15, kq = lane >> 4;
    const int r0 = rg * 16, col0 = blockIdx.x * 32;
    const int Rslice = R / KS;
    const int arow = r0 + cc < M ? r0 + cc : M - 1;
    const float* ap = A + (size_t)arow * R + ks * Rslice + kq * 4;
    const float* wp = W + (size_t)(ks * Rslice + kq * 4) * Nc + col0 + 2 * cc;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int n0 = 0;
    for (; n0 + 64 <= Rslice; n0 += 64) {  // 4 + 16 independent loads in flight per lane, then 32 MFMAs
        f32x4 a[4];
        f32x2 b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(ap + n0 + 16 * u);
#pragma unroll
            for (int s = 0; s < 4; ++s) b[u][s] = *reinterpret_cast<const f32x2*>(wp + (size_t)(n0 + 16 * u + s) * Nc);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc0 = mfma16(a[u][s], b[u][s][0], acc0);
                acc1 = mfma16(a[u][s], b[u][s][1], acc1);
            }
    }
    for (; n0 < Rslice; n0 += 16) {
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
    if (KS > 1) {
        if (ks > 0) {
            part[((ks - 1) * 64 + lane) * 2] = acc0;
            part[((ks - 1) * 64 + lane) * 2 + 1] = acc1;
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int q = 1; q < KS; ++q) {
            acc0 += part[((q - 1) * 64 + lane) * 2];
            acc1 += part[((q - 1) * 64 + lane) * 2 + 1];
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

// dW[N][K] (+)= dy[M][N]^T * x[M][K]  (+ optionally db[N] (+)= column sums of dy);  M <= 64, N % 16 == 0, K % 64 == 0.
// The nn.Linear weight (and bias) gradient at <= 64 rows: the reduction dimension is the 64 rows, so there is nothing to
// split - a wave owns a 16 (n) x 64 (k) output tile (4 accumulators; tile t holds columns k0 + 4*(lane&15) + t), walks
// the rows 4 at a time (one 4-byte dy load + one 16-byte x load per lane per step) and writes - or adds into the
// optimiser's gradient bucket - directly: no partial slabs, no reduction launch, no separate column-sum launches.
__global__ __launch_bounds__(256) void skinny_tn_kernel(const float* __restrict__ DY, const float* __restrict__ X,
                                                        float* __restrict__ dW, float* __restrict__ db, int M, int N,
                                                        int K, int accum, int db_accum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ii = lane & 15, kq = lane >> 4;
    const int ktiles = K / 64;
    const int wt = blockIdx.x * 4 + wave;          // wave tile index: n-tile major
    const int nt = wt / ktiles, kt = wt - nt * ktiles;
    if (nt * 16 >= N) return;
    const int n0 = nt * 16, k0 = kt * 64;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[16];
    f32x4 b[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {   // all 32 loads of the tile in flight together
        const int m = 4 * s + kq;
        const int mc = m < M ? m : M - 1;
        const float av = DY[(size_t)mc * N + n0 + ii];
        const f32x4 bv = *reinterpret_cast<const f32x4*>(X + (size_t)mc * K + k0 + 4 * ii);
        a[s] = m < M ? av : 0.f;
        b[s] = bv;
    }
    __builtin_amdgcn_sched_barrier(0);
    float colsum = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        colsum += a[s];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(a[s], b[s][t], acc[t]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + kq * 4 + r;
        float* o = dW + (size_t)n * K + k0 + 4 * ii;
        f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        if (accum) v += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = v;
    }
    if (db && kt == 0) {  // wave-uniform: the first k-tile of each n-tile also owns the bias gradient
        colsum += __shfl_xor(colsum, 16);
        colsum += __shfl_xor(colsum, 32);
        if (kq == 0) db[n0 + ii] = db_accum ? db[n0 + ii] + colsum : colsum;
    }
}
MIGAN_API int migan_skinny_tn_ok(int M, int N, int K) { return M >= 1 && M <= 64 && N % 16 == 0 && K % 64 == 0; }
// nn.Linear weight / bias gradient for <= 64 rows (wgan_gp.py:46-78 at batch 64): dw[N][K] (+)= dy[M][N]^T x[M][K],
// db[N] (+)= sum_m dy[m][n] (db may be NULL)
MIGAN_API int migan_skinny_tn(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, int accumulate,
                              int db_accumulate, void* stream) {
    if (!migan_skinny_tn_ok(M, N, K)) return (int)hipErrorInvalidValue;
    const int wave_tiles = (N / 16) * (K / 64);
    MIGAN_LAUNCH(skinny_tn_kernel, dim3((wave_tiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, x, dw, db, M, N,
                       K, accumulate, db_accumulate);
    HIP_LAUNCH_CHECK();
    return 0;
}

// 1 when the skinny kernels take the shape (else the caller uses migan_conv2d_fwd / migan_transpose_batched + migan_conv2d_fwd)
MIGAN_API int migan_skinny_nt_ok(int M, int N, int K) { return M >= 1 && M <= 64 && N % 16 == 0 && K % 16 == 0 && K >= 32; }
MIGAN_API int migan_skinny_nn_ok(int M, int R, int Nc) { return M >= 1 && M <= 64 && R % 16 == 0 && Nc % 32 == 0 && R >= 16; }

// nn.Linear forward for <= 64 rows (wgan_gp.py:46-56,73-77): y[M][N] = act(x[M][K] w[N][K]^T + bias)
MIGAN_API int migan_skinny_nt(const float* a, const float* w, const float* bias, float* c, int M, int N, int K, int act,
                              float slope, void* stream) {
    if (!migan_skinny_nt_ok(M, N, K)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(N / 16, (M + 15) / 16);
    if (K % 128 == 0 && K >= 1024)
        MIGAN_LAUNCH(skinny_nt_kernel<8>, grid, dim3(512), 0, st, a, w, bias, c, M, N, K, act, slope);
    else if (K % 64 == 0 && K >= 256)
        MIGAN_LAUNCH(skinny_nt_kernel<4>, grid, dim3(256), 0, st, a, w, bias, c, M, N, K, act, slope);
    else if (K % 32 == 0 && K >= 64)
        MIGAN_LAUNCH(skinny_nt_kernel<2>, grid, dim3(128), 0, st, a, w, bias, c, M, N, K, act, slope);
    else
        MIGAN_LAUNCH(skinny_nt_kernel<1>, grid, dim3(64), 0, st, a, w, bias, c, M, N, K, act, slope);
    HIP_LAUNCH_CHECK();
    return 0;
}
// nn.Linear input gradient for <= 64 rows: dx[M][K] = dy[M][N] w[N][K]  (w in its stored layout; Nc = K, R = N)
MIGAN_API int migan_skinny_nn(const float* a, const float* w, float* c, int M, int R, int Nc, void* stream) {
    if (!migan_skinny_nn_ok(M, R, Nc)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(Nc / 32, (M + 15) / 16);
    if (R % 128 == 0 && R >= 512)
        MIGAN_LAUNCH(skinny_nn_kernel<8>, grid, dim3(512), 0, st, a, w, c, M, R, Nc);
    else if (R % 64 == 0 && R >= 256)
        MIGAN_LAUNCH(skinny_nn_kernel<4>, grid, dim3(256), 0, st, a, w, c, M, R, Nc);
    else if (R % 32 == 0 && R >= 64)
        MIGAN_LAUNCH(skinny_nn_kernel<2>, grid, dim3(128), 0, st, a, w, c, M, R, Nc);
    else
        MIGAN_LAUNCH(skinny_nn_kernel<1>, grid, dim3(64), 0, st, a, w, c, M, R, Nc);
    HIP_LAUNCH_CHECK();
    return 0;
}
