// Thin-N convolutions on the MFMA kernels through a width-Toeplitz expansion.
//
// The image-output layers (cyclegan/models.py:82 ReflectionPad2d(3) + Conv2d(64, 3, 7); srgan/models.py:62
// Conv2d(64, 3, 9, 1, 4)) are GEMMs with N = Co = 3 columns: an MFMA tile
// is >= 87 % padding and the direct VALU kernel (thin_conv_kernel, conv_igemm.hip) reaches 16 % of the fp32 peak
// (profiles/r02_conv_microbench.txt: srgan conv3 2.92 ms forward, 4.11 ms weight gradient).  The kernel COLUMN index s
// can be moved from the reduction dimension into the GEMM's N dimension:
//
//     P[n][h][u][(s, co)] = sum_{r, c} x[n][h + r - pad_t][u][c] * w[co][c][r][s]        (an R x 1 convolution with
//                                                                                         Co' = S*Co "channels")
//     y[n][h][w][co]      = act(bias[co] + sum_s P[n][h][map(w + s - pad_l)][(s, co)])   (shifted diagonal sum; map =
//                                                                                         zero / reflection padding)
//
// so the forward is one implicit GEMM with Co' = 27 (9x9) or 21 (7x7) columns - 84 % / 66 % of a 32-wide MFMA tile
// instead of 9 % - on the existing igemm_pipe_kernel<128, 32>, plus one streaming pass over P.  The backward uses the
// transposed expansion Q[n][h][u][(s, co)] = sum_{w: map(w + s - pad_l) = u} dy[n][h][w][co]:
//     dw[co][c][r][s] = wgrad of the R x 1 convolution (x, Q) -> dwt[(s, co)][c][r], then a fold of 10^4 elements
//     dx              = dgrad of the R x 1 convolution (Q, wd)
// P and Q ([N*Ho*W][Co'] floats, 264 MB for srgan conv3 at batch 16) live in the caller's workspace.
#include "common.h"

MIGAN_API int migan_conv2d_fwd(const float* x, const float* w_ohwi, const float* bias, float* y, int N, int Hi, int Wi, int Ci,
                               int Ho, int Wo, int Co, int R, int S, int stride, int pad_t, int pad_l, int gather, int act,
                               float slope, void* stream);
MIGAN_API int migan_conv2d_dgrad(const float* dy, const float* w_ihwo, const float* bias, float* dx, int N, int Hi, int Wi,
                                 int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t, int pad_l, int act,
                                 float slope, void* stream);
MIGAN_API size_t migan_conv2d_wgrad_workspace(int N, int Ho, int Wo, int Co, int R, int S, int Ci);
MIGAN_API int migan_conv2d_wgrad(const float* x, const float* dy, float* dw_oihw, float* ws, size_t ws_bytes, int N, int Hi,
                                 int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t, int pad_l,
                                 int gather, int accumulate, float* db, int db_accumulate, const float* db_slabs,
                                 int db_nslab, void* stream);
MIGAN_API int migan_gather2d_bwd(const float* dy, float* dx, int N, int Hi, int Wi, int C, int Ho, int Wo, int pad_t,
                                 int pad_l, int mode, void* stream);

static inline int toep_cols(int Co, int S) { return (S * Co + 3) / 4 * 4; }

// 1 when the expansion applies AND pays: <= 4 output channels, stride 1, 16 <= S*Co <= 32 columns (at least half of a
// 32-wide MFMA tile: 7x7 and 9x9 kernels with 3 channels; a 3x3 kernel fills 28 % and stays on thin_conv_kernel), a
// vector-loadable source with a reduction worth a GEMM (Ci % 4 == 0, >= 16), zero or reflection padding
MIGAN_API int migan_thin_toeplitz_ok(int Co, int R, int S, int Ci, int stride, int gather) {
    return Co >= 1 && Co <= 4 && stride == 1 && S * Co >= 16 && S * Co <= 32 && R >= 1 && R <= 16 && Ci % 4 == 0 && Ci >= 16 &&
           (gather == GATHER_ZERO || gather == GATHER_REFLECT);
}
// Co' = S*Co rounded up to a multiple of 4 (row length of wt / P / Q)
MIGAN_API int migan_thin_toeplitz_cols(int Co, int S) { return toep_cols(Co, S); }
// bytes of the P (forward) / Q (backward) buffer
MIGAN_API size_t migan_thin_toeplitz_workspace(int N, int Ho, int Wi, int Co, int S) {
    return (size_t)N * Ho * Wi * toep_cols(Co, S) * sizeof(float);
}

// w_oihw [Co][Ci][R][S] -> wt [Co'][R][Ci] (forward operand, rows (s, co), zero rows up to Co') and wd [Ci][R][Co'] (dgrad)
__global__ void toep_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, float* __restrict__ wd, int Co, int Ci,
                                 int R, int S, int Cop) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cop * R * Ci) return;
    const int c = idx % Ci, r = (idx / Ci) % R, j = idx / (Ci * R);
    float v = 0.f;
    if (j < S * Co) {
        const int s = j / Co, co = j - s * Co;
        v = w[(((size_t)co * Ci + c) * R + r) * S + s];
    }
    wt[idx] = v;
    wd[((size_t)c * R + r) * Cop + j] = v;
}
MIGAN_API int migan_thin_toeplitz_pack(const float* w_oihw, float* wt, float* wd, int Co, int Ci, int R, int S, void* stream) {
    const int Cop = toep_cols(Co, S), n = Cop * R * Ci;
    MIGAN_LAUNCH(toep_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, wt, wd, Co, Ci, R, S,
                       Cop);
    HIP_LAUNCH_CHECK();
    return 0;
}

// y[n][h][w][co] = act(bias[co] + sum_s P[n][h][map(w + s - pl)][s*Co + co]): a workgroup takes 64 output columns of one
// row, stages the 64+S-1 P rows it touches in LDS (each P element is used exactly once: a streaming pass) and thread
// (w, co) adds its S terms.  LDS row stride Co'+1 (odd) -> the column walk of 64 lanes is conflict-free.
#define TOEP_TW 64
__global__ __launch_bounds__(256) void toep_sum_kernel(const float* __restrict__ P, const float* __restrict__ bias,
                                                       float* __restrict__ y, int Ho, int Wo, int Co, int W, int Cop, int S,
                                                       int pl, int gather, int act, float slope) {
    extern __shared__ float lds[];
    const int w0 = blockIdx.x * TOEP_TW, h = blockIdx.y, n = blockIdx.z;
    const int LD = Cop + 1, q4 = Cop >> 2;
    const int nw = Wo - w0 < TOEP_TW ? Wo - w0 : TOEP_TW;  // valid outputs of this workgroup
    const int nv = nw + S - 1;
    const float* Prow = P + ((size_t)(n * Ho + h) * W) * Cop;
    for (int e = threadIdx.x; e < nv * q4; e += 256) {
        const int vi = e / q4, q = e - vi * q4;
        int u;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (map_coord(w0 + vi - pl, W, gather, u)) v = *reinterpret_cast<const f32x4*>(Prow + (size_t)u * Cop + 4 * q);
        float* d = lds + vi * LD + 4 * q;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    const int wl = threadIdx.x & 63, co = threadIdx.x >> 6;
    if (co >= Co || wl >= nw) return;
    float acc = bias ? bias[co] : 0.f;
    for (int s = 0; s < S; ++s) acc += lds[(wl + s) * LD + s * Co + co];
    y[((size_t)(n * Ho + h) * Wo + w0 + wl) * Co + co] = act_apply(acc, act, slope);
}

// forward: x [N][Hi][Wi][Ci], wt from migan_thin_toeplitz_pack, y [N][Ho][Wo][Co]; ws >= migan_thin_toeplitz_workspace()
MIGAN_API int migan_thin_toeplitz_fwd(const float* x, const float* wt, const float* bias, float* y, float* ws, size_t ws_bytes,
                                      int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int pad_t, int pad_l,
                                      int gather, int act, float slope, void* stream) {
    if (!migan_thin_toeplitz_ok(Co, R, S, Ci, 1, gather) || N < 1 || N > 65535 || Ho < 1 || Ho > 65535 || Wo < 1 ||
        ws_bytes < migan_thin_toeplitz_workspace(N, Ho, Wi, Co, S))
        return (int)hipErrorInvalidValue;
    if (gather == GATHER_REFLECT && (pad_l >= Wi || Wo + S - 1 - pad_l - Wi >= Wi)) return (int)hipErrorInvalidValue;
    const int Cop = toep_cols(Co, S);
    if (int rc = migan_conv2d_fwd(x, wt, nullptr, ws, N, Hi, Wi, Ci, Ho, Wi, Cop, R, 1, 1, pad_t, 0, gather, ACT_NONE, 0.f, stream))
        return rc;
    const size_t lds = (size_t)(TOEP_TW + S - 1) * (Cop + 1) * sizeof(float);
    MIGAN_LAUNCH(toep_sum_kernel, dim3((Wo + TOEP_TW - 1) / TOEP_TW, Ho, N), dim3(256), lds, (hipStream_t)stream, ws, bias,
                       y, Ho, Wo, Co, Wi, Cop, S, pad_l, gather, act, slope);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Q[n][h][u][(s, co)] = sum over the output columns w whose tap s reads source column u: the direct one (w = u - s + pl)
// and, under reflection padding, the mirrored ones on either side.  Columns >= S*Co are zero.
__global__ __launch_bounds__(256) void toep_expand_kernel(const float* __restrict__ dy, float* __restrict__ Q, int Ho, int Wo,
                                                          int Co, int W, int Cop, int S, int pl, int gather) {
    const int u0 = blockIdx.x * TOEP_TW, h = blockIdx.y, n = blockIdx.z;
    const float* drow = dy + (size_t)(n * Ho + h) * Wo * Co;
    float* qrow = Q + ((size_t)(n * Ho + h) * W + u0) * Cop;
    const int nu = W - u0 < TOEP_TW ? W - u0 : TOEP_TW;
    const int SC = S * Co;
    for (int e = threadIdx.x; e < nu * Cop; e += 256) {
        const int ul = e / Cop, j = e - ul * Cop;
        float v = 0.f;
        if (j < SC) {
            const int s = j / Co, co = j - s * Co;
            const int u = u0 + ul;
            const int w1 = u - s + pl;
            if ((unsigned)w1 < (unsigned)Wo) v = drow[w1 * Co + co];
            if (gather == GATHER_REFLECT) {
                const int w2 = -u - s + pl;               // w + s - pl = -u        (left mirror, u > 0)
                const int w3 = 2 * (W - 1) - u - s + pl;  // w + s - pl = 2(W-1)-u  (right mirror, u < W-1)
                if (u > 0 && (unsigned)w2 < (unsigned)Wo) v += drow[w2 * Co + co];
                if (u < W - 1 && (unsigned)w3 < (unsigned)Wo) v += drow[w3 * Co + co];
            }
        }
        qrow[e] = v;
    }
}
// dy [N][Ho][Wo][Co] (gradient w.r.t. the pre-activation output) -> q [N][Ho][Wi][Co']
MIGAN_API int migan_thin_toeplitz_expand(const float* dy, float* q, int N, int Ho, int Wo, int Co, int Wi, int S, int pad_l,
                                         int gather, void* stream) {
    if (N < 1 || N > 65535 || Ho < 1 || Ho > 65535 || Co < 1 || Co > 4 || S * Co > 32) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(toep_expand_kernel, dim3((Wi + TOEP_TW - 1) / TOEP_TW, Ho, N), dim3(256), 0, (hipStream_t)stream, dy, q, Ho,
                       Wo, Co, Wi, toep_cols(Co, S), S, pad_l, gather);
    HIP_LAUNCH_CHECK();
    return 0;
}

// dw[co][c][r][s] (+)= dwt[(s*Co + co)][c][r]
__global__ void toep_fold_dw_kernel(const float* __restrict__ dwt, float* __restrict__ dw, int Co, int Ci, int R, int S,
                                    int accum) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Co * Ci * R * S) return;
    const int s = idx % S, r = (idx / S) % R, c = (idx / (S * R)) % Ci, co = idx / (S * R * Ci);
    const float v = dwt[((size_t)(s * Co + co) * Ci + c) * R + r];
    dw[idx] = accum ? dw[idx] + v : v;
}
MIGAN_API size_t migan_thin_toeplitz_wgrad_workspace(int N, int Ho, int Wi, int Ci, int Co, int R, int S) {
    const int Cop = toep_cols(Co, S);
    return (size_t)Cop * Ci * R * sizeof(float) + migan_conv2d_wgrad_workspace(N, Ho, Wi, Cop, R, 1, Ci);
}
// weight gradient: x [N][Hi][Wi][Ci], q from migan_thin_toeplitz_expand, dw_oihw [Co][Ci][R][S] (accumulate != 0: +=)
MIGAN_API int migan_thin_toeplitz_wgrad(const float* x, const float* q, float* dw_oihw, float* ws, size_t ws_bytes, int N, int Hi,
                                        int Wi, int Ci, int Ho, int Co, int R, int S, int pad_t, int gather, int accumulate,
                                        void* stream) {
    if (!migan_thin_toeplitz_ok(Co, R, S, Ci, 1, gather) || ws_bytes < migan_thin_toeplitz_wgrad_workspace(N, Ho, Wi, Ci, Co, R, S))
        return (int)hipErrorInvalidValue;
    const int Cop = toep_cols(Co, S);
    float* dwt = ws;
    float* ws2 = ws + (size_t)Cop * Ci * R;
    if (int rc = migan_conv2d_wgrad(x, q, dwt, ws2, ws_bytes - (size_t)Cop * Ci * R * sizeof(float), N, Hi, Wi, Ci, Ho, Wi, Cop, R, 1,
                                    1, pad_t, 0, gather, 0, nullptr, 0, nullptr, 0, stream))
        return rc;
    const int n = Co * Ci * R * S;
    MIGAN_LAUNCH(toep_fold_dw_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dwt, dw_oihw, Co, Ci, R, S,
                       accumulate);
    HIP_LAUNCH_CHECK();
    return 0;
}

// bytes of the row-padded intermediate of the reflection-padded input gradient (0 for zero padding)
MIGAN_API size_t migan_thin_toeplitz_dgrad_workspace(int N, int Hi, int Wi, int Ci, int Ho, int R, int gather) {
    return gather == GATHER_REFLECT ? (size_t)N * (Ho + R - 1) * Wi * Ci * sizeof(float) : 0;
}
// input gradient: q [N][Ho][Wi][Co'], wd from migan_thin_toeplitz_pack, dx [N][Hi][Wi][Ci]
MIGAN_API int migan_thin_toeplitz_dgrad(const float* q, const float* wd, float* dx, float* ws, size_t ws_bytes, int N, int Hi,
                                        int Wi, int Ci, int Ho, int Co, int R, int S, int pad_t, int gather, void* stream) {
    if (!migan_thin_toeplitz_ok(Co, R, S, Ci, 1, gather)) return (int)hipErrorInvalidValue;
    const int Cop = toep_cols(Co, S);
    if (gather == GATHER_ZERO)
        return migan_conv2d_dgrad(q, wd, nullptr, dx, N, Hi, Wi, Ci, Ho, Wi, Cop, R, 1, 1, pad_t, 0, ACT_NONE, 0.f, stream);
    const int Hp = Ho + R - 1;  // rows of the reflection-padded source
    if (ws_bytes < migan_thin_toeplitz_dgrad_workspace(N, Hi, Wi, Ci, Ho, R, gather)) return (int)hipErrorInvalidValue;
    if (int rc = migan_conv2d_dgrad(q, wd, nullptr, ws, N, Hp, Wi, Ci, Ho, Wi, Cop, R, 1, 1, 0, 0, ACT_NONE, 0.f, stream)) return rc;
    return migan_gather2d_bwd(ws, dx, N, Hi, Wi, Ci, Hp, Wi, pad_t, 0, GATHER_REFLECT, stream);
}
