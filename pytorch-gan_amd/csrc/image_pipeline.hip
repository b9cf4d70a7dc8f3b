// Device side of the reference's input pipeline: what the DataLoader workers do per image between the
// decoded uint8 bitmap and the fp32 training batch - cyclegan.py:111-117 (Resize(int(h*1.12), BICUBIC), RandomCrop,
// RandomHorizontalFlip, ToTensor, Normalize), srgan/datasets.py:16-33 (two BICUBIC resizes of every image), dcgan.py:120-131
// (Resize(img_size) of MNIST, bilinear), pix2pix/datasets.py (BICUBIC resize + flip) - as three streaming kernels over a batch
// of equally sized images.  Byte / integer work, HBM bound; results are BIT-EXACT with Pillow + torchvision:
//   * resample: Pillow's ImagingResample (src/libImaging/Resample.c) for 8-bit images - separable, horizontal pass first,
//     int32 fixed point with 22 fractional bits, the intermediate image rounded to uint8 (clip8).  The per-output-column
//     coefficient rows and source bounds are computed on the host exactly as precompute_coeffs / normalize_coeffs_8bpc do
//     (pytorch_gan_amd/data.py) - they depend on the sizes only and are uploaded once per (in, out, filter).
//   * u8 -> f32: crop window + horizontal flip + ToTensor (x / 255) + Normalize ((x - mean[c]) / std[c]) with one IEEE
//     rounding per operation in torchvision's order, written NHWC (this package's activation layout) or NCHW.
#include "common.h"

#define RESAMPLE_PRECISION_BITS (32 - 8 - 2)

__device__ __forceinline__ unsigned char resample_clip8(int v) {  // Resample.c clip8()
    if (v >= (1 << RESAMPLE_PRECISION_BITS << 8)) return 255;
    if (v <= 0) return 0;
    return (unsigned char)(v >> RESAMPLE_PRECISION_BITS);
}

// horizontal pass: dst[row][xx][c] = clip8(2^21 + sum_x src[row][xmin + x][c] * kk[xx][x]); rows = N * H
template <int C>
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                         const int* __restrict__ kk, const int* __restrict__ bounds, int ksize,
                                                         size_t rows, int Wi, int Wo) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * (size_t)Wo) return;
    const size_t row = idx / Wo;
    const int xx = (int)(idx - row * Wo);
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    const unsigned char* s = src + (row * Wi + xmin) * C;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (RESAMPLE_PRECISION_BITS - 1);
    for (int x = 0; x < n; ++x) {
        const int kv = k[x];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += (int)s[x * C + c] * kv;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dst[idx * C + c] = resample_clip8(acc[c]);
}

// vertical pass: dst[n][yy][e] = clip8(2^21 + sum_y src[n][ymin + y][e] * kk[yy][y]); e runs over the W*C bytes of a row
__global__ __launch_bounds__(256) void resample_v_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                         const int* __restrict__ kk, const int* __restrict__ bounds, int ksize, int N,
                                                         int Hi, int Ho, size_t rowbytes) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)N * Ho * rowbytes) return;
    const size_t e = idx % rowbytes, t = idx / rowbytes;
    const int yy = (int)(t % Ho);
    const size_t n = t / Ho;
    const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
    const int* k = kk + (size_t)yy * ksize;
    const unsigned char* s = src + (n * Hi + ymin) * rowbytes + e;
    int acc = 1 << (RESAMPLE_PRECISION_BITS - 1);
    for (int y = 0; y < cnt; ++y) acc += (int)s[(size_t)y * rowbytes] * k[y];
    dst[idx] = resample_clip8(acc);
}

// One separable pass of Pillow's 8-bit resample over a batch [N][Hi][Wi][C] (C <= 4) of uint8 images.
// axis 1: width Wi -> out (dst [N][Hi][out][C]); axis 0: height Hi -> out (dst [N][out][Wi][C]).
// kk: [out][ksize] int32 fixed-point coefficients, bounds: [out][2] = (first source index, tap count).
MIGAN_API int migan_resample_u8(const unsigned char* src, unsigned char* dst, const int* kk, const int* bounds, int ksize, int N,
                                int Hi, int Wi, int C, int out, int axis, void* stream) {
    if (N < 1 || Hi < 1 || Wi < 1 || C < 1 || C > 4 || out < 1 || ksize < 1 || (axis != 0 && axis != 1))
        return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (axis == 1) {
        const size_t rows = (size_t)N * Hi, total = rows * out;
        const dim3 grid((unsigned)((total + 255) / 256));
        switch (C) {
            case 1: MIGAN_LAUNCH(resample_h_kernel<1>, grid, dim3(256), 0, st, src, dst, kk, bounds, ksize, rows, Wi, out); break;
            case 2: MIGAN_LAUNCH(resample_h_kernel<2>, grid, dim3(256), 0, st, src, dst, kk, bounds, ksize, rows, Wi, out); break;
            case 3: MIGAN_LAUNCH(resample_h_kernel<3>, grid, dim3(256), 0, st, src, dst, kk, bounds, ksize, rows, Wi, out); break;
            default: MIGAN_LAUNCH(resample_h_kernel<4>, grid, dim3(256), 0, st, src, dst, kk, bounds, ksize, rows, Wi, out); break;
        }
    } else {
        const size_t rowbytes = (size_t)Wi * C, total = (size_t)N * out * rowbytes;
        MIGAN_LAUNCH(resample_v_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, dst, kk, bounds, ksize, N,
                           Hi, out, rowbytes);
    }
    HIP_LAUNCH_CHECK();
    return 0;
}

// crop (per-image top-left corner) + optional horizontal flip + ToTensor + Normalize: dst[n][y][x][c] (nchw == 0) or
// dst[n][c][y][x] (nchw != 0) = (src[n][cy + y][cx + (flip ? w-1-x : x)][c] / 255 - mean[c]) / std[c]
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst,
                                                        const int* __restrict__ crop_yx, const unsigned char* __restrict__ flip,
                                                        const float* __restrict__ mean, const float* __restrict__ stdv, int N, int Hi,
                                                        int Wi, int C, int h, int w, int nchw) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)h * w * C;
    if (idx >= (size_t)N * per) return;
    const size_t n = idx / per;
    size_t r = idx - n * per;
    int y, x, c;
    if (nchw) {
        c = (int)(r / ((size_t)h * w));
        r -= (size_t)c * h * w;
        y = (int)(r / w);
        x = (int)(r - (size_t)y * w);
    } else {
        y = (int)(r / ((size_t)w * C));
        r -= (size_t)y * w * C;
        x = (int)(r / C);
        c = (int)(r - (size_t)x * C);
    }
    const int cy = crop_yx ? crop_yx[2 * n] : 0, cx = crop_yx ? crop_yx[2 * n + 1] : 0;
    const int sx = cx + ((flip && flip[n]) ? w - 1 - x : x);
    const unsigned char px = src[((n * Hi + cy + y) * Wi + sx) * C + c];
    float v = __fdiv_rn((float)px, 255.f);              // ToTensor: .div(255)
    if (mean) v = __fdiv_rn(__fsub_rn(v, mean[c]), stdv[c]);  // Normalize: .sub_(mean).div_(std)
    dst[idx] = v;
}
// src [N][Hi][Wi][C] uint8; crop_yx (may be NULL: corner 0,0) [N][2] int32; flip (may be NULL) [N] bytes; mean / std (both or
// neither) [C] fp32; dst fp32 [N][h][w][C] or [N][C][h][w].  The crop window must lie inside the image.
MIGAN_API int migan_u8_to_f32(const unsigned char* src, float* dst, const int* crop_yx, const unsigned char* flip, const float* mean,
                              const float* stdv, int N, int Hi, int Wi, int C, int h, int w, int nchw, void* stream) {
    if (N < 1 || C < 1 || h < 1 || w < 1 || h > Hi || w > Wi || ((mean == nullptr) != (stdv == nullptr)))
        return (int)hipErrorInvalidValue;
    const size_t total = (size_t)N * h * w * C;
    MIGAN_LAUNCH(u8_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, crop_yx,
                       flip, mean, stdv, N, Hi, Wi, C, h, w, nchw);
    HIP_LAUNCH_CHECK();
    return 0;
}
