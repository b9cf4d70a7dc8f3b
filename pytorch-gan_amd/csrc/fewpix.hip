// Convolutions with a handful of pixels and megabytes of weights: the inner levels of the pix2pix U-Net at batch 1
// (pix2pix/models.py:62-71: Conv2d / ConvTranspose2d 512->512 and 1024->512, 4x4 stride 2, at 8x8 ... 1x1 pixels - 16.8 and
// 33.5 MB of weights per layer against 1-64 pixels).  They are weight STREAMS, not GEMM tiles: the tiled kernels spend them on
// OHWI / IHWO packs of the weight (read + write 2 x 16.8 MB per layer and step), a split-K forward, a split-K input gradient, 17-34 MB
// of weight-gradient slabs and their reduction.  Here every product is one of the <= 64-row skinny GEMMs of skinny_mm.hip on the
// weight IN ITS STORED torch LAYOUT (no pack, no slab, no reduction launch), around two small index kernels:
//
//   Conv2d forward          col = im2col(x)            [M_out][Ci*T]   y  = act(col W^T + b)        skinny_nt, W as [Co][Ci*T]
//   Conv2d input gradient   ycol = dy W                [M_out][Ci*T]   dx = col2im(ycol)            skinny_nn
//   Conv2d weight gradient  dW (+)= dy^T col           [Co][Ci*T]                                   skinny_tn (col kept from the forward)
//   ConvTranspose2d forward ycol = x W                 [M_in][Co*T]    y  = act(col2im(ycol) + b)   skinny_nn, W as [Ci][Co*T]
//   ConvTranspose2d dgrad   dycol = im2col(dy)         [M_in][Co*T]    dx = dycol W^T               skinny_nt
//   ConvTranspose2d wgrad   dW (+)= x^T dycol          [Ci][Co*T]                                   skinny_tn
//
// (T = R*S taps; column order (channel, tap) = the weight's own trailing dimensions.)  The col matrices are at most
// 64 x 16384 floats (4 MB, L2 / MALL resident); the weights are read once per product at streaming rates.
// Measured (profiles/r03_abi_check.txt): per 16.8 MB layer forward 46.6 -> 13.8 us, input gradient 50.5 -> 9.8 us, weight gradient
// 39.8 -> 12.7 us against the pack + split-K kernels that served these shapes before.
#include "common.h"

extern "C" int migan_skinny_nt_ok(int M, int N, int K);
extern "C" int migan_skinny_nt(const float* a, const float* w, const float* bias, float* c, int M, int N, int K, int act, float slope,
                               void* stream);

// col[m][c*T + t] = x[n][ho*stride - pt + r][wo*stride - pl + s][c]  (0 outside the image); m = (n*Ho + ho)*Wo + wo, t = r*S + s.
// One thread per element of col, consecutive threads = consecutive columns: the stores are contiguous, the loads hit a <= 64-pixel
// (cache-resident) source.
__global__ __launch_bounds__(256) void im2col_small_kernel(const float* __restrict__ x, float* __restrict__ col, int H, int W, int C,
                                                           int Ho, int Wo, int R, int S, int stride, int pt, int pl, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int T = R * S, K = C * T;
    const int m = (int)(idx / K), k = (int)(idx - (long)m * K);
    const int c = k / T, t = k - c * T;
    const int r = t / S, s = t - r * S;
    const int wo = m % Wo, q = m / Wo;
    const int ho = q % Ho, n = q / Ho;
    const int h = ho * stride - pt + r, w = wo * stride - pl + s;
    float v = 0.f;
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) v = x[(((long)n * H + h) * W + w) * C + c];
    col[idx] = v;
}

// out[n][h][w][j] = act(bias[j] + sum over taps t = (r, s) and pixels (ho, wo) with ho*stride - pt + r == h, wo*stride - pl + s == w
//                                 of ycol[(n*Ho + ho)*Wo + wo][j*T + t]),   taps in increasing order (deterministic).
// One thread per output element, consecutive threads = consecutive channels j.
__global__ __launch_bounds__(256) void col2im_small_kernel(const float* __restrict__ ycol, const float* __restrict__ bias,
                                                           float* __restrict__ out, int H, int W, int J, int Ho, int Wo, int R, int S,
                                                           int stride, int pt, int pl, int act, float slope, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int T = R * S;
    const int j = (int)(idx % J);
    const long p = idx / J;
    const int w = (int)(p % W);
    const long q = p / W;
    const int h = (int)(q % H), n = (int)(q / H);
    float acc = bias ? bias[j] : 0.f;
    for (int r = 0; r < R; ++r) {
        const int hn = h + pt - r;
        if (hn < 0 || hn % stride != 0) continue;
        const int ho = hn / stride;
        if (ho >= Ho) continue;
        for (int s = 0; s < S; ++s) {
            const int wn = w + pl - s;
            if (wn < 0 || wn % stride != 0) continue;
            const int wo = wn / stride;
            if (wo >= Wo) continue;
            acc += ycol[(((long)n * Ho + ho) * Wo + wo) * ((long)J * T) + (long)j * T + r * S + s];
        }
    }
    out[idx] = act_apply(acc, act, slope);
}

// ---- the NT product at streaming rate ------------------------------------------------------------------------------------------
// skinny_nt_kernel (skinny_mm.hip) gives one workgroup a 16-column tile and ALL of K: N / 16 = 32-64 workgroups for these layers,
// each walking 8-16 dependent load rounds - fine for the 2.6 MB L2-resident critic it was written for, a sixth of the chip's memory
// parallelism for a 16.8-33.5 MB weight stream.  Here the same wave-level loop (16-byte loads of both operands straight into
// v_mfma_f32_16x16x4_f32, k order permuted identically for both) also splits K over blockIdx.z: N/16 x row groups x Z workgroups
// (256-512), one or two load rounds per wave, raw partial tiles into part[Z][M][N], and a second small launch adds the Z partials
// in z order (deterministic) with bias and activation.
__device__ __forceinline__ f32x4 fp_mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// part[z][m][n] = sum over k in [z*Kc, (z+1)*Kc) of A[m][k] W[n][k];  M <= 64, N % 16 == 0, Kc % (16 * KS) == 0
template <int KS>
__global__ __launch_bounds__(64 * KS) void fewpix_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                             float* __restrict__ part, int M, int N, int K, int Kc) {
    __shared__ f32x4 red[KS > 1 ? (KS - 1) * 64 : 1];
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int rr = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.y * 16, col0 = blockIdx.x * 16, z = blockIdx.z;
    const int Kslice = Kc / KS;
    const int arow = r0 + rr < M ? r0 + rr : M - 1;
    const size_t kbase = (size_t)z * Kc + (size_t)ks * Kslice + kq * 4;
    const float* ap = A + (size_t)arow * K + kbase;
    const float* wp = W + (size_t)(col0 + rr) * K + kbase;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int k0 = 0;
    for (; k0 + 128 <= Kslice; k0 += 128) {  // 16 independent 16-byte loads in flight per lane, then 32 MFMAs
        f32x4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(ap + k0 + 16 * u);
            b[u] = *reinterpret_cast<const f32x4*>(wp + k0 + 16 * u);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u += 2)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc0 = fp_mfma16(a[u][t], b[u][t], acc0);
                acc1 = fp_mfma16(a[u + 1][t], b[u + 1][t], acc1);
            }
    }
    for (; k0 < Kslice; k0 += 16) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + k0), b0 = *reinterpret_cast<const f32x4*>(wp + k0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc0 = fp_mfma16(a0[t], b0[t], acc0);
    }
    f32x4 acc = acc0 + acc1;
    if (KS > 1) {
        if (ks > 0) red[(ks - 1) * 64 + lane] = acc;
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int q = 1; q < KS; ++q) acc += red[(q - 1) * 64 + lane];
    }
    const int col = col0 + rr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + kq * 4 + r;
        if (row < M) part[((size_t)z * M + row) * N + col] = acc[r];
    }
}

// out[m][n] = act(bias[n] + part[0][m][n] + part[1][m][n] + ... ) - the partials in z order
__global__ __launch_bounds__(256) void fewpix_nt_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                               float* __restrict__ out, int MN, int N, int Z, int act, float slope) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float v = part[i];
    for (int z = 1; z < Z; ++z) v += part[(size_t)z * MN + i];
    if (bias) v += bias[i % N];
    out[i] = act_apply(v, act, slope);
}

// K-split of the NT product: the largest Z in {16, 8, 4, 2} that leaves every wave whole 128-deep load rounds (8 waves x 128 =
// 1024 of K per workgroup and round) and keeps the launch at or under 512 workgroups; 1 = no split (skinny_nt_kernel's own launch)
static int fewpix_nt_split(int M, int N, int K) {
    const long tiles = (long)(N / 16) * ((M + 15) / 16);
    for (int Z = 16; Z >= 2; Z >>= 1)
        if (K % (Z * 1024) == 0 && tiles * Z <= 512) return Z;
    return 1;
}
// bytes of the partial-tile workspace migan_fewpix_nt needs (0: no split)
MIGAN_API size_t migan_fewpix_nt_workspace(int M, int N, int K) {
    const int Z = fewpix_nt_split(M, N, K);
    return Z > 1 ? (size_t)Z * M * N * sizeof(float) : 0;
}
// out[M][N] = act(a[M][K] w[N][K]^T + bias): the forward of a few-pixel Conv2d (a = its im2col) and the input gradient of a
// few-pixel ConvTranspose2d (pix2pix/models.py:23,39), K split over workgroups; ws from migan_fewpix_nt_workspace (may be NULL when 0).
MIGAN_API int migan_fewpix_nt(const float* a, const float* w, const float* bias, float* out, float* ws, size_t ws_bytes, int M, int N,
                              int K, int act, float slope, void* stream) {
    if (migan_skinny_nt_ok(M, N, K) != 1) return (int)hipErrorInvalidValue;
    const int Z = fewpix_nt_split(M, N, K);
    if (Z == 1) return migan_skinny_nt(a, w, bias, out, M, N, K, act, slope, stream);
    if (ws == nullptr || ws_bytes < (size_t)Z * M * N * sizeof(float)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int Kc = K / Z;
    MIGAN_LAUNCH(fewpix_nt_kernel<8>, dim3(N / 16, (M + 15) / 16, Z), dim3(512), 0, st, a, w, ws, M, N, K, Kc);
    HIP_LAUNCH_CHECK();
    MIGAN_LAUNCH(fewpix_nt_reduce_kernel, dim3(cdiv((long)M * N, 256L)), dim3(256), 0, st, ws, bias, out, M * N, N, Z, act, slope);
    HIP_LAUNCH_CHECK();
    return 0;
}

// 1 when a conv whose GEMM has `rows` pixel rows (Conv2d: N*Ho*Wo output pixels; ConvTranspose2d: N*Hin*Win input pixels),
// `n` = the weight's leading dimension (Conv2d: Co; ConvTranspose2d: Ci) and `k` = the product of its trailing ones takes this
// path: the three skinny GEMM forms take the shape and the weight is large enough to be a stream.
MIGAN_API int migan_fewpix_ok(int rows, int n, int k) {
    return rows >= 1 && rows <= 64 && n % 16 == 0 && n >= 16 && k % 64 == 0 && k >= 1024 &&
                   (long)n * k >= (1L << 20)
               ? 1
               : 0;
}

// im2col of an NHWC image batch x[N][H][W][C] for a conv R x S / stride / pads (pt, pl) with Ho x Wo output pixels:
// col[N*Ho*Wo][C*R*S], column (c, r, s).  Conv2d forward operand (pix2pix/models.py:23) and the dy operand of ConvTranspose2d's
// gradients (pix2pix/models.py:39: there (H, W) is the transposed conv's OUTPUT and (Ho, Wo) its input).
MIGAN_API int migan_im2col_small(const float* x, float* col, int N, int H, int W, int C, int Ho, int Wo, int R, int S, int stride,
                                 int pt, int pl, void* stream) {
    if (N < 1 || H < 1 || W < 1 || C < 1 || Ho < 1 || Wo < 1 || R < 1 || S < 1 || stride < 1) return (int)hipErrorInvalidValue;
    const long total = (long)N * Ho * Wo * C * R * S;
    if (total >= (1L << 31)) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(im2col_small_kernel, dim3((unsigned)cdiv(total, 256L)), dim3(256), 0, (hipStream_t)stream, x, col, H, W, C, Ho,
                       Wo, R, S, stride, pt, pl, total);
    HIP_LAUNCH_CHECK();
    return 0;
}

// The adjoint: out[N][H][W][J] = act(bias + col2im(ycol[N*Ho*Wo][J*R*S])).  Conv2d input gradient (bias NULL, act 0) and
// ConvTranspose2d forward (its output is the (H, W) side).
MIGAN_API int migan_col2im_small(const float* ycol, const float* bias, float* out, int N, int H, int W, int J, int Ho, int Wo, int R,
                                 int S, int stride, int pt, int pl, int act, float slope, void* stream) {
    if (N < 1 || H < 1 || W < 1 || J < 1 || Ho < 1 || Wo < 1 || R < 1 || S < 1 || stride < 1) return (int)hipErrorInvalidValue;
    const long total = (long)N * H * W * J;
    if (total >= (1L << 31) || (long)N * Ho * Wo * J * R * S >= (1L << 31)) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(col2im_small_kernel, dim3((unsigned)cdiv(total, 256L)), dim3(256), 0, (hipStream_t)stream, ycol, bias, out, H,
                       W, J, Ho, Wo, R, S, stride, pt, pl, act, slope, total);
    HIP_LAUNCH_CHECK();
    return 0;
}
