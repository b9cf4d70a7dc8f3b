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

// Row length Co' of wt / P / Q: S*Co (16..32) padded to 32.  (Round 5: it was the next multiple of 4 - 28 for the 9x9, 24 for the 7x7
// layer.  With 32 the input-gradient GEMM - source channels = Co' - is taken by the LDS-DMA kernels, which want >= 32 source channels and
// whole 32-channel K-tiles; the forward GEMM's N tile is 32 wide either way.  P / Q grow by 14 % / 33 %.)
static inline int toep_cols(int Co, int S) { return S * Co <= 32 ? 32 : (S * Co + 3) / 4 * 4; }

// 1 when the expansion applies AND pays: <= 4 output channels, stride 1, 16 <= S*Co <= 32 columns (at least half of a
// 32-wide MFMA tile: 7x7 and 9x9 kernels with 3 channels; a 3x3 kernel fills 28 % and stays on thin_conv_kernel), a
// vector-loadable source with a reduction worth a GEMM (Ci % 4 == 0, >= 16), zero or reflection padding
MIGAN_API int migan_thin_toeplitz_ok(int Co, int R, int S, int Ci, int stride, int gather) {
    return Co >= 1 && Co <= 4 && stride == 1 && S * Co >= 16 && S * Co <= 32 && R >= 1 && R <= 16 && Ci % 4 == 0 && Ci >= 16 &&
           (gather == GATHER_ZERO || gather == GATHER_REFLECT);
}
// Co' = 32 (row length of wt / P / Q)
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

// ------------------------------------------------------------------------------------------------
// Weight gradient of the R x 1 convolution with the image rows kept in LDS (round 5).
// The general weight-gradient kernel sees dwt[(s,co)][(r,c)] = sum_p Q[p][(s,co)] * x[p + (r - pad_t) rows][c] as a GEMM with M = Co' = 28
// rows - its 64-row tiles are 56 % padding - and N = R * Ci columns cut into tiles per tap r, each of which fetches "its" source row per
// K-tile: every image row is fetched R times (profiles/r04_pmc_kernels.json: 1.45 ms, 5.0 GB against 0.87 GB of operands, MFMA busy on
// twice the necessary work).  Here a workgroup owns a 16-pixel-wide column strip of one image and WALKS DOWN it: the R source rows of the
// current output row sit in an LDS ring of R + 1 slots, each step fetches ONE new row (4 KB) and one Q row (2 KB) while the previous
// ones are multiplied, and the whole N = R * 64 columns belong to the workgroup.  M = 32 rows: 27 / 32 of the MFMA work is real.
// Partial [32][R*64] slabs per workgroup and pixel half, one fixed-order fold.
// ------------------------------------------------------------------------------------------------
#define TWR_PX 16
struct ToepRingGeom {
    int N, Hi, Wi, Ho, Co, Cop, S, pad_t, reflect;
    int rows_per_seg, segs;
};
// Four waves: wave = (g, j); g = which 8 of the K-tile's 16 pixels it contracts, j = which channel half: its R column blocks are
// (tap r, half j), r = 0 .. R-1.  (A first version gave six / seven waves two or three blocks each over all 16 pixels: 1.16 ms for the
// 9 x 9 layer - a workgroup of six waves puts two waves on two SIMDs and one on the others, the MFMA work of a CU is 2:2:1:1.)
// The two pixel halves leave separate slabs; the fold adds them like slabs of different workgroups.
template <int R>
__global__ __launch_bounds__(256, 2) void toep_wgrad_ring_kernel(const ToepRingGeom g, const float* __restrict__ x,
                                                                  const float* __restrict__ q, float* __restrict__ part) {
    constexpr int CI = 64, NSLOT = R + 1, ROWF = TWR_PX * CI;   // floats per ring slot
    constexpr int NT = 256, HS = TWR_PX / 4;                     // MFMA steps (2 pixels each) per wave and K-tile
    __shared__ __attribute__((aligned(16))) float ring[NSLOT * ROWF];
    __shared__ __attribute__((aligned(16))) float qs[2 * TWR_PX * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kk = lane >> 5;
    const int pg = wave >> 1, ch = wave & 1;
    const int strips = g.Wi / TWR_PX;
    int wg = blockIdx.x;
    const int seg = wg % g.segs;
    wg /= g.segs;
    const int strip = wg % strips, n = wg / strips;
    const int u0 = strip * TWR_PX;
    const int oi0 = seg * g.rows_per_seg;
    const int oi1 = oi0 + g.rows_per_seg < g.Ho ? oi0 + g.rows_per_seg : g.Ho;

    // loader roles: every thread one 16-byte chunk of an image row (pixel tid / 16, channels 4 * (tid % 16) ..), the first
    // 16 * Cop / 4 threads also one chunk of a Q row
    const int q4 = g.Cop >> 2;
    const bool q_role = tid < TWR_PX * q4;
    const int xp = tid >> 4, xc = (tid & 15) * 4;
    const int qp = q_role ? tid / q4 : 0, qc = q_role ? (tid - qp * q4) * 4 : 0;
    auto load_x = [&](int v) -> f32x4 {   // virtual (padded) row v of the strip
        f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
        int ih = v;
        bool ok = true;
        if (g.reflect) {
            ih = ih < 0 ? -ih : ih;
            ih = ih >= g.Hi ? 2 * g.Hi - 2 - ih : ih;
        } else {
            ok = (unsigned)ih < (unsigned)g.Hi;
        }
        if (ok) r4 = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * g.Hi + ih) * g.Wi + u0 + xp) * CI + xc);
        return r4;
    };
    auto load_q = [&](int oi) -> f32x4 {
        f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
        if (q_role && oi < oi1) r4 = *reinterpret_cast<const f32x4*>(q + ((size_t)(n * g.Ho + oi) * g.Wi + u0 + qp) * g.Cop + qc);
        return r4;
    };
    auto slot_of = [&](int v) { return (v + 4 * NSLOT) % NSLOT; };   // v >= -pad_t > -4 * NSLOT

    for (int e = tid; e < 2 * TWR_PX * 32; e += NT) qs[e] = 0.f;   // columns Cop .. 31 stay zero
    __syncthreads();
    // the R rows of the first output row, and its Q row
    if (oi0 < oi1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f32x4 t = load_x(oi0 - g.pad_t + r);
            *reinterpret_cast<f32x4*>(ring + slot_of(oi0 - g.pad_t + r) * ROWF + xp * CI + xc) = t;
        }
        const f32x4 t = load_q(oi0);
        if (q_role) *reinterpret_cast<f32x4*>(qs + qp * 32 + qc) = t;
    }
    __syncthreads();

    f32x16 acc[R];
#pragma unroll
    for (int b = 0; b < R; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    int cur = 0;
    // fetch distance TWO output rows: the row (and Q row) that output row oi + 2 adds is requested while row oi is multiplied and written
    // to LDS at the end of row oi + 1 - one row of MFMAs (~1 us) is less than a loaded memory system's latency (with distance one:
    // 0.96 ms for the 9 x 9 layer, profiles/r05_ab.txt)
    f32x4 nx = {0.f, 0.f, 0.f, 0.f}, nq = {0.f, 0.f, 0.f, 0.f};
    if (oi0 + 1 < oi1) {
        nx = load_x(oi0 - g.pad_t + R);
        nq = load_q(oi0 + 1);
    }
    for (int oi = oi0; oi < oi1; ++oi) {
        const int vnew = oi - g.pad_t + R;   // the row output row oi + 1 adds: in nx / nq since the previous iteration
        const bool more = oi + 1 < oi1;
        f32x4 fx = {0.f, 0.f, 0.f, 0.f}, fq = {0.f, 0.f, 0.f, 0.f};
        if (oi + 2 < oi1) {
            fx = load_x(vnew + 1);
            fq = load_q(oi + 2);
        }
        const float* qb = qs + cur * (TWR_PX * 32) + (pg * 2 * HS + kk) * 32 + l31;
        float a[HS];
#pragma unroll
        for (int st = 0; st < HS; ++st) a[st] = qb[2 * st * 32];
        int slot = slot_of(oi - g.pad_t);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float* rb = ring + slot * ROWF + (pg * 2 * HS + kk) * CI + ch * 32 + l31;
#pragma unroll
            for (int st = 0; st < HS; ++st)
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], rb[2 * st * CI], acc[r], 0, 0, 0);
            slot = slot + 1 == NSLOT ? 0 : slot + 1;
        }
        if (more) {   // the free slot (the row that left the window with the previous output row) and the other Q buffer
            *reinterpret_cast<f32x4*>(ring + slot_of(vnew) * ROWF + xp * CI + xc) = nx;
            if (q_role) *reinterpret_cast<f32x4*>(qs + (cur ^ 1) * (TWR_PX * 32) + qp * 32 + qc) = nq;
        }
        nx = fx;
        nq = fq;
        __syncthreads();
        cur ^= 1;
    }
    // slab [32][R * 64] per (workgroup, pixel half): this wave's column blocks
    float* out = part + ((size_t)blockIdx.x * 2 + pg) * 32 * (R * CI);
#pragma unroll
    for (int b = 0; b < R; ++b) {
        const int col = b * CI + ch * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            out[(size_t)row * (R * CI) + col] = acc[b][r];
        }
    }
}
// dw[co][c][r][s] (+)= sum_wg part[wg][s * Co + co][r * 64 + c], slabs added in index order
__global__ __launch_bounds__(256) void toep_ring_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int nslab, int Co,
                                                             int R, int S, int accum) {
    __shared__ float red[256];
    const int lo = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int NC = R * 64, total = S * Co * NC;
    const int e = blockIdx.x * 16 + lo;
    float s = 0.f;
    if (e < total) {
        const size_t stride = (size_t)32 * NC;
        const int per = (nslab + 15) / 16;
        const int b = grp * per, en = b + per < nslab ? b + per : nslab;
        for (int w = b; w < en; ++w) s += part[(size_t)w * stride + e];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && e < total) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += red[k * 16 + lo];
        const int row = e / NC, rem = e - row * NC;
        const int r = rem >> 6, c = rem & 63;
        const int s_ = row / Co, co = row - s_ * Co;
        float* o = dw + (((size_t)co * 64 + c) * R + r) * S + s_;
        *o = accum ? *o + s : s;
    }
}
static bool toep_ring_ok(int Co, int R, int S, int Ci, int Wi, int Cop) {
    return Ci == 64 && (R == 7 || R == 9) && R == S && Wi % TWR_PX == 0 && Cop <= 32 && S * Co <= 32;
}
static void toep_ring_plan(int N, int Wi, int Ho, int& segs, int& rows) {
    const int strips = N * (Wi / TWR_PX);
    segs = (512 + strips - 1) / strips;          // two workgroups per CU (accumulators: R x 16 registers per lane)
    if (segs < 1) segs = 1;
    if (segs > Ho / 16) segs = Ho / 16 > 0 ? Ho / 16 : 1;   // a walk re-fetches R - 1 rows at its start: at least 16 rows long
    rows = (Ho + segs - 1) / segs;
    segs = (Ho + rows - 1) / rows;
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
    size_t general = (size_t)Cop * Ci * R * sizeof(float) + migan_conv2d_wgrad_workspace(N, Ho, Wi, Cop, R, 1, Ci);
    if (toep_ring_ok(Co, R, S, Ci, Wi, Cop)) {
        int segs, rows;
        toep_ring_plan(N, Wi, Ho, segs, rows);
        const size_t ring = (size_t)N * (Wi / TWR_PX) * segs * 2 * 32 * R * 64 * sizeof(float);
        if (ring > general) general = ring;
    }
    return general;
}
// weight gradient: x [N][Hi][Wi][Ci], q from migan_thin_toeplitz_expand, dw_oihw [Co][Ci][R][S] (accumulate != 0: +=)
MIGAN_API int migan_thin_toeplitz_wgrad(const float* x, const float* q, float* dw_oihw, float* ws, size_t ws_bytes, int N, int Hi,
                                        int Wi, int Ci, int Ho, int Co, int R, int S, int pad_t, int gather, int accumulate,
                                        void* stream) {
    if (!migan_thin_toeplitz_ok(Co, R, S, Ci, 1, gather) || ws_bytes < migan_thin_toeplitz_wgrad_workspace(N, Ho, Wi, Ci, Co, R, S))
        return (int)hipErrorInvalidValue;
    const int Cop = toep_cols(Co, S);
    static const int ring_env = getenv("MIGAN_TOEP_RING") ? atoi(getenv("MIGAN_TOEP_RING")) : 1;   // A/B knob (round 5)
    if (ring_env && toep_ring_ok(Co, R, S, Ci, Wi, Cop)) {
        ToepRingGeom g = {N, Hi, Wi, Ho, Co, Cop, S, pad_t, gather == GATHER_REFLECT, 0, 0};
        toep_ring_plan(N, Wi, Ho, g.segs, g.rows_per_seg);
        const int wgs = N * (Wi / TWR_PX) * g.segs;
        if (R == 9) MIGAN_LAUNCH((toep_wgrad_ring_kernel<9>), dim3(wgs), dim3(256), 0, (hipStream_t)stream, g, x, q, ws);
        else MIGAN_LAUNCH((toep_wgrad_ring_kernel<7>), dim3(wgs), dim3(256), 0, (hipStream_t)stream, g, x, q, ws);
        HIP_LAUNCH_CHECK();
        const int total = S * Co * R * 64;
        MIGAN_LAUNCH(toep_ring_fold_kernel, dim3((total + 15) / 16), dim3(256), 0, (hipStream_t)stream, ws, dw_oihw, 2 * wgs, Co, R, S,
                     accumulate);
        HIP_LAUNCH_CHECK();
        return 0;
    }
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
