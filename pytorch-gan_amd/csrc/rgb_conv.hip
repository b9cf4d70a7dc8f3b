// Image-input convolutions (3 source channels, stride 1) on the MFMA units without an im2col buffer and without the general
// kernels' 32-channel K-tiles.
//
// Reference call sites: srgan/models.py:85 (discriminator Conv2d(3, 64, 3, 1, 1) + LeakyReLU on 384 x 384 images - three forwards and
// two weight gradients per step), torchvision vgg19.features[0] behind srgan/models.py:11 (Conv2d(3, 64, 3, padding=1) + ReLU, two
// forwards per step), cyclegan/models.py:49-50 (ReflectionPad2d(3) + Conv2d(3, 64, 7)).
//
// With Ci = 3 the reduction of a 3x3 layer is K = 27: the general implicit-GEMM kernels pad every tap to a 32-channel tile (10.7x the
// work) and the VALU kernels that served these layers ran at 2-15 % of peak while writing or reading a 604 MB activation
// (profiles/r04_pmc_kernels.json: forward 277 us against a 120 us store floor, weight gradient 508 us, activation backward + bias
// column sums 351 us in a launch of their own).  In NHWC with 3 channels the S taps of one kernel row are 3*S CONTIGUOUS floats, so
// the whole patch of a pixel is R runs: the GEMM operand is read straight out of R staged image rows in LDS.
//
//   forward   y[p][co] = act(b[co] + sum_k patch[p][k] * w[k][co]),  k = (r*S + s)*3 + c:
//             M = 32 pixels per wave, N = Co in 32-column blocks, K = 27 -> 14 steps of v_mfma_f32_32x32x2_f32; a workgroup owns
//             128 pixels of an output row (x TH rows), stages the R source rows once, keeps w [K][Co] in LDS.  Bound by the store.
//   weight gradient, with the activation backward and the bias gradient inside:
//             dw[co][k] = sum_p g[p][co] * patch[p][k],  g = dy * act'(y):  M = Co, N = K + 1 columns (column K is the constant 1:
//             the bias gradient is one more column of the same product), contraction over pixels.  The A operand - 32 output
//             channels of one pixel per half-wave - is read straight from dy and y in the MFMA's own layout (128 B per half-wave),
//             the B operand from the staged rows.  Partial [Co][32*NBK] slabs per workgroup, then a fixed-order reduction
//             (deterministic) that also writes the OIHW layout.  Bound by reading dy and y once.
#include "common.h"

#define RGB_TW 128   // output pixels of one row per workgroup (4 waves x 32)

__device__ __forceinline__ int rgb_map(int v, int L, int reflect) {   // source coordinate or -1 (zero); both forms computed, one selected
    int vr = v < 0 ? -v : v;
    vr = vr >= L ? 2 * L - 2 - vr : vr;
    const int vz = (unsigned)v < (unsigned)L ? v : -1;
    return reflect ? vr : vz;
}

struct RgbGeom {
    int N, H, W, Ho, Wo, Co;
    int pad_t, pad_l, reflect, act;
    float slope;
    int rows_per_wg;   // output rows a workgroup walks
    int flip;          // forward kernel: tap t of the staged weight is read from tap T-1-t of w (the input gradient of a thin-output layer)
};

// The R source rows of output row oh, columns [ow0 - pad_l, ow0 - pad_l + TW + S - 1), 3 channels -> xs[R][XW].  Two halves: load() puts
// this thread's elements into registers with EVERY load issued before the first use (a padded element reads word 0 of the tensor and is
// replaced by zero: no branch, no wait between loads), store() writes them to LDS - callers put the previous tile's arithmetic between
// the two, so the fetch of tile t+1 runs under the MFMAs and stores of tile t.
template <int R, int S, int C = 3>
struct RgbStage {
    static constexpr int XP = RGB_TW + S - 1, XW = XP * C + 1, NE = R * XP * C, NL = (NE + 255) / 256;
    float v[NL];
    __device__ __forceinline__ void load(const RgbGeom& g, const float* __restrict__ x, int n, int oh, int ow0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int e = (int)threadIdx.x + i * 256;
            e = e < NE ? e : NE - 1;
            const int r = e / (XP * C), q = e - r * (XP * C);
            const int j = q / C, c = q - j * C;
            const int ih = rgb_map(oh + r - g.pad_t, g.H, g.reflect), iw = rgb_map(ow0 + j - g.pad_l, g.W, g.reflect);
            const bool ok = (ih | iw) >= 0;
            const size_t idx = ok ? ((size_t)(n * g.H + ih) * g.W + iw) * C + c : 0;
            const float t = x[idx];
            v[i] = ok ? t : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* xs) const {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = (int)threadIdx.x + i * 256;
            if (e < NE) {
                const int r = e / (XP * C), q = e - r * (XP * C);
                xs[r * XW + q] = v[i];
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------ forward
template <int R, int S, int NB, int C = 3>
__global__ __launch_bounds__(256) void rgb_conv_fwd_kernel(const RgbGeom g, const float* __restrict__ x, const float* __restrict__ wk,
                                                           const float* __restrict__ bias, float* __restrict__ y) {
    constexpr int K = R * S * C, K2 = (K + 1) & ~1;
    constexpr int XP = RGB_TW + S - 1, XW = XP * C + 1;
    extern __shared__ float rgb_lds[];
    float* wl = rgb_lds;                 // [K2][NB * 32]
    float* xs = rgb_lds + K2 * NB * 32;  // [R][XW]
    const int Co = NB * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kk = lane >> 5;
    const int ow0 = blockIdx.x * RGB_TW, n = blockIdx.z;
    const int oh_begin = blockIdx.y * g.rows_per_wg;
    const int oh_end = oh_begin + g.rows_per_wg < g.Ho ? oh_begin + g.rows_per_wg : g.Ho;

    for (int e = threadIdx.x; e < K2 * Co; e += 256) {   // wk is [K][Co]; row K (odd K) = 0; flip: tap t <- tap T-1-t
        const int k = e / Co, col = e - k * Co;
        const int ks = g.flip ? (R * S - 1 - k / C) * C + k % C : k;
        wl[e] = k < K ? wk[ks * Co + col] : 0.f;
    }
    float bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bv[nb] = bias ? bias[nb * 32 + l31] : 0.f;
    const int a_base = (wave * 32 + l31) * C;
    // none / LeakyReLU / ReLU as one select: negative-side factor 1 / slope / 0
    const float ns = g.act == ACT_NONE ? 1.f : (g.act == ACT_LRELU ? g.slope : 0.f);
    const bool relu = g.act == ACT_RELU;   // negative side = the constant 0 (v * 0 would turn -inf into NaN)

    RgbStage<R, S, C> stage;
    if (oh_begin < oh_end) stage.load(g, x, n, oh_begin, ow0);
    for (int oh = oh_begin; oh < oh_end; ++oh) {
        __syncthreads();   // the previous row's reads of xs are over (first time round: nothing)
        stage.store(xs);
        __syncthreads();   // xs (and, first time round, wl) complete
        if (oh + 1 < oh_end) stage.load(g, x, n, oh + 1, ow0);   // in flight under this row's MFMAs and stores
        f32x16 acc[NB];   // starts at the bias: the epilogue needs no loaded value
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = bv[nb];
#pragma unroll
        for (int st = 0; st < K2 / 2; ++st) {
            // k = 2*st + kk; patch element k of pixel p sits at xs[(k / (3S)) * XW + p*3 + k % (3S)]; k >= K multiplies the zero row of wl
            constexpr int S3 = C * S;
            const int k0 = 2 * st, k1 = 2 * st + 1;
            const int o0 = (k0 / S3) * XW + k0 % S3;
            const int o1 = k1 < K ? (k1 / S3) * XW + k1 % S3 : o0;   // (the product with the zero row must not read a NaN: any valid word)
            const float a = xs[a_base + (kk ? o1 : o0)];
            const int krow = (2 * st + kk) * Co;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float b = wl[krow + nb * 32 + l31];
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nb], 0, 0, 0);
            }
        }
        const size_t rowbase = (size_t)(n * g.Ho + oh) * g.Wo;
        // none / LeakyReLU / ReLU only (the launcher refuses the others).  Full tiles (Wo % 128 == 0: every layer of the reference) store
        // unconditionally: ONE basic block of 32 selects and stores.  (With a per-element switch over five activations, and then with a
        // bounds branch around every store, the compiler put `s_waitcnt vmcnt(0)` in front of each store - every store waited for the one
        // before it and for the next row's fetch: 215 us for the 604 MB of SRGAN's first layer.)
        if (g.Wo % RGB_TW == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* o = y + (rowbase + ow0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk) * Co + l31;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float v = acc[nb][r];
                    o[nb * 32] = v > 0.f ? v : (relu ? 0.f : v * ns);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ow = ow0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                float* o = y + (rowbase + ow) * Co + l31;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float v = acc[nb][r];
                    if (ow < g.Wo) o[nb * 32] = v > 0.f ? v : (relu ? 0.f : v * ns);
                }
            }
        }
    }
}

// 1 when the image-input kernels take this layer: 3 source channels, stride 1, square 3 / 7 kernel, 32 or 64 output channels,
// zero or reflection padding smaller than the image, and enough pixels to be worth a launch of 128-pixel row tiles
MIGAN_API int migan_rgb_conv_ok(int Ci, int Co, int R, int S, int stride, int gather, long long pixels) {
    // (9 x 9 - srgan/models.py:38 on 96 x 96 images - was measured too: 89-120 us against 94 us for the general kernel; not taken)
    if ((Ci != 3 && !(Ci == 1 && R == 3)) || stride != 1 || R != S || (R != 3 && R != 7) || (Co != 32 && Co != 64)) return 0;
    if (gather != GATHER_ZERO && gather != GATHER_REFLECT) return 0;
    return pixels >= 16384 ? 1 : 0;
}

static size_t rgb_fwd_lds(int R, int S, int Co, int C) {
    const int K2 = (R * S * C + 1) & ~1;
    return ((size_t)K2 * Co + (size_t)R * ((RGB_TW + S - 1) * C + 1)) * sizeof(float);
}

// x [N][H][W][Ci] (Ci = 3; 1 with a 3x3 kernel), w_hwio [R][S][Ci][Co] (= the OIHW weight permuted (2,3,1,0)), bias [Co] or NULL,
// y [N][Ho][Wo][Co].  flip != 0: the taps are read in reverse order - with x = dy [N][H][W][c] and w_hwio = the weight [c][Co][R][S] of
// a thin-output layer permuted (2,3,0,1) this is that layer's INPUT gradient (dcgan.py:62: 64 channels back from 1 or 3)
MIGAN_API int migan_rgb_conv_fwd(const float* x, const float* w_hwio, const float* bias, float* y, int N, int H, int W, int Ci, int Ho,
                                 int Wo, int Co, int R, int S, int pad_t, int pad_l, int gather, int act, float slope, int flip,
                                 void* stream) {
    if (!migan_rgb_conv_ok(Ci, Co, R, S, 1, gather, (long long)N * Ho * Wo) || N < 1 || N > 65535) return (int)hipErrorInvalidValue;
    if (act != ACT_NONE && act != ACT_LRELU && act != ACT_RELU) return (int)hipErrorInvalidValue;   // the activations image-input layers carry
    if (gather == GATHER_REFLECT && (pad_t >= H || pad_l >= W || Ho + R - 1 - pad_t - H >= H || Wo + S - 1 - pad_l - W >= W))
        return (int)hipErrorInvalidValue;
    RgbGeom g = {N, H, W, Ho, Wo, Co, pad_t, pad_l, gather == GATHER_REFLECT, act, slope, 0, flip != 0};
    // rows per workgroup: the weights are staged once per workgroup - long enough walks to amortise that, enough workgroups for 8 / CU
    const int xt = (Wo + RGB_TW - 1) / RGB_TW;
    int th = 8;
    while (th > 1 && (long)xt * ((Ho + th - 1) / th) * N < 2048) th >>= 1;
    g.rows_per_wg = th;
    const dim3 grid(xt, (Ho + th - 1) / th, N);
    const size_t lds = rgb_fwd_lds(R, S, Co, Ci);
#define RGB_FWD(R_, NB_, C_)                                                                                                   \
    do {                                                                                                                       \
        /* once per DEVICE (the attribute is per device: a second GPU of the same process needs it too) */                     \
        static bool attr_set[64] = {};                                                                                         \
        int dev__ = 0;                                                                                                         \
        (void)hipGetDevice(&dev__);                                                                                            \
        dev__ &= 63;                                                                                                           \
        if (!attr_set[dev__] && lds > 48 * 1024) {                                                                             \
            hipFuncSetAttribute((const void*)rgb_conv_fwd_kernel<R_, R_, NB_, C_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                                     \
            attr_set[dev__] = true;                                                                                            \
        }                                                                                                                      \
        MIGAN_LAUNCH((rgb_conv_fwd_kernel<R_, R_, NB_, C_>), grid, dim3(256), lds, (hipStream_t)stream, g, x, w_hwio, bias, y); \
    } while (0)
    if (Ci == 1) {
        if (Co == 64) RGB_FWD(3, 2, 1); else RGB_FWD(3, 1, 1);
    } else if (Co == 64) {
        if (R == 3) RGB_FWD(3, 2, 3); else RGB_FWD(7, 2, 3);
    } else {
        if (R == 3) RGB_FWD(3, 1, 3); else RGB_FWD(7, 1, 3);
    }
#undef RGB_FWD
    HIP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ weight gradient (+ activation backward, + bias)
// slab layout: part[wg][co][NBK * 32]; column j < K: dw of patch element j, column K: the bias gradient, beyond: zero
// HAS_ACT: g = dy * (y > 0 ? 1 : ns) with ns = slope (LeakyReLU) or 0 (ReLU) - the only activations an image-input layer of the
// reference carries (srgan/models.py:85 LeakyReLU(0.2), vgg19.features[1] ReLU); false: dy is the gradient of the pre-activation
template <int R, int S, int MB, int NBK, bool HAS_ACT>
__global__ __launch_bounds__(256) void rgb_conv_wgrad_kernel(const RgbGeom g, const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ yact, float* __restrict__ part, int tiles_x,
                                                             int tiles_total, float ns) {
    constexpr int K = R * S * 3, S3 = 3 * S;
    constexpr int XP = RGB_TW + S - 1, XW = XP * 3 + 1;
    constexpr int J = NBK * 32;
    static_assert(K + 1 <= J, "the bias column needs room");
    extern __shared__ float rgb_lds[];
    float* xs = rgb_lds;   // [R][XW]; after the walk: the workgroup's slab [MB * 32][J]
    const int Co = MB * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kk = lane >> 5;

    f32x16 acc[MB][NBK];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NBK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // B operand of column block nbk: patch element j = nbk*32 + l31 of pixel p = wave*32 + 2*step + kk
    int b_off[NBK];      // LDS word of that element for p = wave*32 + kk (step adds 6), or -1: the constant column / a zero column
    float b_const[NBK];
#pragma unroll
    for (int nbk = 0; nbk < NBK; ++nbk) {
        const int j = nbk * 32 + l31;
        b_off[nbk] = j < K ? (j / S3) * XW + j % S3 + (wave * 32 + kk) * 3 : -1;
        b_const[nbk] = j == K ? 1.f : 0.f;
    }

    // a workgroup walks the row tiles t = blockIdx.x, blockIdx.x + gridDim.x, ... (tile = 128 pixels of one output row)
    RgbStage<R, S> stage;
    int t = blockIdx.x;
    if (t < tiles_total) {
        const int row = t / tiles_x;
        stage.load(g, x, row / g.Ho, row % g.Ho, (t % tiles_x) * RGB_TW);
    }
    for (; t < tiles_total; t += gridDim.x) {
        const int tx = t % tiles_x, row = t / tiles_x;   // row = n * Ho + oh
        const int ow0 = tx * RGB_TW;
        // this wave's 32 pixels: g = dy * act'(y) in the MFMA A layout (lane: channel l31 of block mb, pixel 2*step + kk).  Every load is
        // issued before the first use; a pixel beyond the row reads a valid word and is replaced by zero (no branch between loads)
        const int pw = ow0 + wave * 32;
        const size_t gbase = ((size_t)row * g.Wo + pw) * Co + l31;
        float av[16][MB], yv[HAS_ACT ? 16 : 1][MB];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int p = 2 * st + kk;
            const size_t idx = pw + p < g.Wo ? gbase + (size_t)p * Co : (size_t)l31;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                av[st][mb] = dy[idx + mb * 32];
                if (HAS_ACT) yv[st][mb] = yact[idx + mb * 32];
            }
        }
        __syncthreads();   // the previous tile's reads of xs are over
        stage.store(xs);
        __syncthreads();
        const int tn = t + (int)gridDim.x;
        if (tn < tiles_total) {   // the next tile's image rows: in flight under this tile's MFMAs
            const int rown = tn / tiles_x;
            stage.load(g, x, rown / g.Ho, rown % g.Ho, (tn % tiles_x) * RGB_TW);
        }
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const bool ok = pw + 2 * st + kk < g.Wo;
            float b[NBK];
#pragma unroll
            for (int nbk = 0; nbk < NBK; ++nbk) b[nbk] = b_off[nbk] >= 0 ? xs[b_off[nbk] + 6 * st] : b_const[nbk];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                float a = ok ? av[st][mb] : 0.f;
                if (HAS_ACT) a *= yv[st][mb] > 0.f ? 1.f : ns;
#pragma unroll
                for (int nbk = 0; nbk < NBK; ++nbk)
                    acc[mb][nbk] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[nbk], acc[mb][nbk], 0, 0, 0);
            }
        }
    }
    // the four waves' partial products, added in wave order through LDS (a fixed order), then the slab
    float* slab = rgb_lds;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nbk = 0; nbk < NBK; ++nbk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                        float* o = slab + co * J + nbk * 32 + l31;
                        *o = w == 0 ? acc[mb][nbk][r] : *o + acc[mb][nbk][r];
                    }
        }
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * Co * J;
    for (int e = threadIdx.x; e < Co * J; e += 256) out[e] = slab[e];
}

// dw_oihw[co][c][r][s] (+)= sum_wg part[wg][co][(r*S + s)*3 + c];  db[co] (+)= sum_wg part[wg][co][K]  - slabs added in index order
__global__ __launch_bounds__(256) void rgb_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                               int nslab, int Co, int J, int R, int S, int accum_w, int accum_b) {
    __shared__ float red[256];
    const int K = R * S * 3;
    const int lo = threadIdx.x & 15, grp = threadIdx.x >> 4;   // 16 outputs x 16 slab groups
    const int idx = blockIdx.x * 16 + lo;                      // (co, j), j <= K
    const int co = idx / (K + 1), j = idx - co * (K + 1);
    float s = 0.f;
    if (co < Co) {
        const float* p = part + (size_t)co * J + j;
        const size_t stride = (size_t)Co * J;
        const int per = (nslab + 15) / 16;
        const int b = grp * per, e = b + per < nslab ? b + per : nslab;
        for (int q = b; q < e; ++q) s += p[(size_t)q * stride];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && co < Co) {
#pragma unroll
        for (int q = 1; q < 16; ++q) s += red[q * 16 + lo];
        if (j == K) {
            if (db) db[co] = accum_b ? db[co] + s : s;
        } else {
            const int c = j % 3, t = j / 3, r = t / S, s_ = t - r * S;
            float* o = dw + (((size_t)co * 3 + c) * R + r) * S + s_;
            *o = accum_w ? *o + s : s;
        }
    }
}

#define RGB_WGRAD_WGS 1024
MIGAN_API size_t migan_rgb_conv_wgrad_workspace(int Co, int R, int S) {
    const int J = (R * S * 3 + 1 + 31) / 32 * 32;
    return (size_t)RGB_WGRAD_WGS * Co * J * sizeof(float);
}
// 1 when migan_rgb_conv_wgrad takes the layer (the 9 x 9 kernel's 8 column blocks do not fit a wave's accumulators)
MIGAN_API int migan_rgb_conv_wgrad_ok(int Ci, int Co, int R, int S, int stride, int gather, long long pixels) {
    return Ci == 3 && migan_rgb_conv_ok(Ci, Co, R, S, stride, gather, pixels) && R <= 7 && (R == 3 || Co == 64);
}
// dw_oihw [Co][3][R][S] (accumulate_w: +=) and, when db != NULL, db [Co] (accumulate_b: +=) of
//   y = act(conv(x, w) + b):  g = dy * act'(y_act)  (y_act = the layer's OUTPUT, NULL with act = ACT_NONE: dy is used as it is),
//   dw = sum_p g[p] (x) patch[p],  db = sum_p g[p]
// - the activation backward and the bias column sums happen inside the weight-gradient launch: dy and y are read once, nothing is
// written but the slabs.  ws >= migan_rgb_conv_wgrad_workspace() bytes.
MIGAN_API int migan_rgb_conv_wgrad(const float* x, const float* dy, const float* y_act, float* dw_oihw, float* db, float* ws,
                                   size_t ws_bytes, int N, int H, int W, int Ho, int Wo, int Co, int R, int S, int pad_t, int pad_l,
                                   int gather, int act, float slope, int accumulate_w, int accumulate_b, void* stream) {
    if (!migan_rgb_conv_wgrad_ok(3, Co, R, S, 1, gather, (long long)N * Ho * Wo) || ws_bytes < migan_rgb_conv_wgrad_workspace(Co, R, S))
        return (int)hipErrorInvalidValue;
    if (act != ACT_NONE && (!y_act || (act != ACT_LRELU && act != ACT_RELU))) return (int)hipErrorInvalidValue;
    if (gather == GATHER_REFLECT && (pad_t >= H || pad_l >= W || Ho + R - 1 - pad_t - H >= H || Wo + S - 1 - pad_l - W >= W))
        return (int)hipErrorInvalidValue;
    RgbGeom g = {N, H, W, Ho, Wo, Co, pad_t, pad_l, gather == GATHER_REFLECT, act, slope, 0, 0};
    const int tiles_x = (Wo + RGB_TW - 1) / RGB_TW;
    const long tiles = (long)tiles_x * Ho * N;
    if (tiles > 0x7fffffffL) return (int)hipErrorInvalidValue;
    const int wgs = tiles < RGB_WGRAD_WGS ? (int)tiles : RGB_WGRAD_WGS;
    const int J = (R * S * 3 + 1 + 31) / 32 * 32;
    const size_t rows_lds = (size_t)R * ((RGB_TW + S - 1) * 3 + 1) * sizeof(float), slab_lds = (size_t)Co * J * sizeof(float);
    const size_t lds = rows_lds > slab_lds ? rows_lds : slab_lds;
    const float ns = act == ACT_LRELU ? slope : 0.f;
#define RGB_WG(R_, MB_, NBK_)                                                                                                     \
    do {                                                                                                                          \
        if (act != ACT_NONE)                                                                                                      \
            MIGAN_LAUNCH((rgb_conv_wgrad_kernel<R_, R_, MB_, NBK_, true>), dim3(wgs), dim3(256), lds, (hipStream_t)stream, g, x, dy, \
                         y_act, ws, tiles_x, (int)tiles, ns);                                                                     \
        else                                                                                                                      \
            MIGAN_LAUNCH((rgb_conv_wgrad_kernel<R_, R_, MB_, NBK_, false>), dim3(wgs), dim3(256), lds, (hipStream_t)stream, g, x,    \
                         dy, y_act, ws, tiles_x, (int)tiles, ns);                                                                 \
    } while (0)
    if (R == 3 && Co == 64) RGB_WG(3, 2, 1);
    else if (R == 3) RGB_WG(3, 1, 1);
    else RGB_WG(7, 2, 5);
#undef RGB_WG
    HIP_LAUNCH_CHECK();
    const int outs = Co * (R * S * 3 + 1);
    MIGAN_LAUNCH(rgb_wgrad_reduce_kernel, dim3((outs + 15) / 16), dim3(256), 0, (hipStream_t)stream, ws, dw_oihw, db, wgs, Co, J, R, S,
                 accumulate_w, accumulate_b);
    HIP_LAUNCH_CHECK();
    return 0;
}

// (Round 5 also built the thin-OUTPUT 3x3 layers - dcgan.py:62 forward, and the input gradients of the image-input layers above - as an
// MFMA kernel: nine taps as GEMM columns, the A operand read from global memory in the MFMA's own layout (a lane = one pixel's 32
// contiguous channels), a 9-term gather out of an LDS ring.  Measured and removed: 115.6 us against 62.0 us for the one-pixel-per-lane
// VALU kernel on dcgan.py:62, 553.9 us against 354 + 162 us on SRGAN's first-layer gradient, the DCGAN step 2.62 vs 2.53 ms
// (profiles/r05_ab.txt call 16) - 64 lanes reading 16 bytes each from 64 different cache lines per instruction thrash the L1.)
