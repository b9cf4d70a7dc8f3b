// HBM-bound pointwise / index-remap kernels of the GAN hot path (NHWC for 4-D tensors).
// Reference layers: nn.LeakyReLU / ReLU / Tanh / Sigmoid / PReLU, nn.Dropout2d(0.25) (dcgan.py:78),
// nn.Dropout(0.5) (pix2pix/models.py:27,44), nn.Upsample(scale_factor=2) (dcgan.py:54),
// nn.ReflectionPad2d (cyclegan/models.py:27), nn.ZeroPad2d((1,0,1,0)) (cyclegan/models.py:117),
// nn.PixelShuffle(2) (srgan/models.py:56), MaxPool2d(2,2) of VGG19 (srgan/models.py:11-12),
// torch.cat(.,1) (pix2pix/models.py:50,132), residual adds (cyclegan/models.py:37, srgan/models.py:30,68).
#include "common.h"

static int grid_for(size_t nvec) {
    size_t b = (nvec + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

#define GRID_STRIDE(i, n)                                                        \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);      \
         i += (size_t)gridDim.x * blockDim.x)

// q = i / d, returns i % d.  Flat indices are size_t for safety, but every tensor of the path has < 2^32 elements: take
// the 32-bit divide (~10x fewer instructions than the emulated 64-bit one) whenever the value fits.
__device__ __forceinline__ unsigned divmod(size_t i, unsigned d, size_t& q) {
    if (i <= 0xffffffffull) {
        const unsigned iu = (unsigned)i, qq = iu / d;
        q = qq;
        return iu - qq * d;
    }
    q = i / d;
    return (unsigned)(i - q * d);
}

// ------------------------------------------------------------------ activations
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act,
                               float slope) {
    size_t n4 = n / 4;
    GRID_STRIDE(i, n4) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = act_apply(v[k], act, slope);
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
    size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = act_apply(x[t], act, slope);
}

// dx = dy * act'(.) expressed through the activation output y
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                               float* __restrict__ dx, size_t n, int act, float slope) {
    size_t n4 = n / 4;
    GRID_STRIDE(i, n4) {
        f32x4 d = reinterpret_cast<const f32x4*>(dy)[i];
        f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = d[k] * act_grad_from_out(v[k], act, slope);
        reinterpret_cast<f32x4*>(dx)[i] = o;
    }
    size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dx[t] = dy[t] * act_grad_from_out(y[t], act, slope);
}

// dx[n][p][c] = dy * mask[n][c] * act'(y): backward of act -> Dropout2d in one pass (y is the masked output; where the
// mask is 0 the result is 0, elsewhere y keeps the sign the activation derivative needs).  C % 4 == 0.
__global__ void act_bwd_nc_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                  const float* __restrict__ mask, float* __restrict__ dx, int HW, int C, size_t n4,
                                  int act, float slope) {
    const size_t per_img = (size_t)HW * C / 4;
    const int cq = C / 4;
    GRID_STRIDE(i, n4) {
        size_t img, tmp_;
        divmod(i, (unsigned)per_img, img);
        const int c4 = (int)divmod(i, (unsigned)cq, tmp_);
        const f32x4 d = reinterpret_cast<const f32x4*>(dy)[i];
        const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
        const f32x4 m = reinterpret_cast<const f32x4*>(mask)[img * cq + c4];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = d[k] * m[k] * act_grad_from_out(v[k], act, slope);
        reinterpret_cast<f32x4*>(dx)[i] = o;
    }
}
MIGAN_API int migan_act_bwd_nc(const float* dy, const float* y, const float* mask, float* dx, int N, int HW, int C,
                               int act, float slope, void* stream) {
    size_t total = (size_t)N * HW * C;
    if (total == 0) return 0;
    if (C % 4 != 0) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(act_bwd_nc_kernel, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, dy, y, mask, dx,
                       HW, C, total / 4, act, slope);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Second derivative through the activation backward (gradient penalties differentiate dx = g * f'(y) again):
// out = gg * g * d f'(y)/dy, with f' expressed through the output y: tanh 1 - y^2 -> -2y, sigmoid y(1-y) -> 1 - 2y;
// zero for the piecewise-linear activations (dragan.py:91-92 ends in Sigmoid).
__global__ void act_bwd2_kernel(const float* __restrict__ g, const float* __restrict__ gg, const float* __restrict__ y,
                                float* __restrict__ out, size_t n, int act) {
    GRID_STRIDE(i, n) {
        const float v = y[i];
        const float d2 = act == ACT_TANH ? -2.f * v : (act == ACT_SIGMOID ? 1.f - 2.f * v : 0.f);
        out[i] = gg[i] * g[i] * d2;
    }
}
MIGAN_API int migan_act_bwd2(const float* g, const float* gg, const float* y, float* out, size_t n, int act,
                             void* stream) {
    if (n == 0) return 0;
    MIGAN_LAUNCH(act_bwd2_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, gg, y, out, n, act);
    HIP_LAUNCH_CHECK();
    return 0;
}

// DRAGAN interpolation (dragan.py:147-149): out = alpha*X + (1 - alpha)*(X + 0.5*std(X)*noise), std = the UNBIASED
// standard deviation over all n elements of X, given as device scalars mean/var (biased variance from
// migan_norm_moments on the flattened tensor) so that no host sync is needed.
__global__ void dragan_interp_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                     const float* __restrict__ noise, const float* __restrict__ var,
                                     float* __restrict__ out, size_t n) {
    const double nn = (double)n;
    const float sd = (float)sqrt((double)var[0] * (nn > 1.0 ? nn / (nn - 1.0) : 1.0));
    const float hs = 0.5f * sd;
    GRID_STRIDE(i, n) {
        const float a = alpha[i], xv = x[i];
        out[i] = a * xv + (1.f - a) * (xv + hs * noise[i]);
    }
}
MIGAN_API int migan_dragan_interp(const float* x, const float* alpha, const float* noise, const float* var_biased,
                                  float* out, size_t n, void* stream) {
    if (n == 0) return 0;
    MIGAN_LAUNCH(dragan_interp_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, alpha, noise,
                       var_biased, out, n);
    HIP_LAUNCH_CHECK();
    return 0;
}

// y[n][p] = alpha * a[n][p] + beta * mean_m b[m][p]   (a may be NULL).  The relativistic average GAN logits of
// esrgan.py:137,165-166 / relativistic_gan.py:149-158: D(x) - mean over the batch of D(other), forward (alpha 1, beta -1)
// and the backward into `other` (a NULL, beta -1: the negated batch mean of the upstream gradient, broadcast).
__global__ void batch_mean_axpy_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int N,
                                       size_t P, float alpha, float beta) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float m = 0.f;
    for (int n = 0; n < N; ++n) m += b[(size_t)n * P + p];
    m = beta * (m / (float)N);
    for (int n = 0; n < N; ++n) y[(size_t)n * P + p] = (a ? alpha * a[(size_t)n * P + p] : 0.f) + m;
}
MIGAN_API int migan_batch_mean_axpy(const float* a, const float* b, float* y, int N, size_t P, float alpha, float beta,
                                    void* stream) {
    if (N <= 0 || P == 0) return 0;
    MIGAN_LAUNCH(batch_mean_axpy_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, y, N, P,
                       alpha, beta);
    HIP_LAUNCH_CHECK();
    return 0;
}

MIGAN_API int migan_act_fwd(const float* x, float* y, size_t n, int act, float slope, void* stream) {
    if (n == 0) return 0;
    MIGAN_LAUNCH(act_fwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, y, n, act,
                       slope);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_act_bwd(const float* dy, const float* y, float* dx, size_t n, int act, float slope,
                            void* stream) {
    if (n == 0) return 0;
    MIGAN_LAUNCH(act_bwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n,
                       act, slope);
    HIP_LAUNCH_CHECK();
    return 0;
}

// PReLU with one shared slope (nn.PReLU(), srgan/models.py:24,38,57): y = x>0 ? x : a*x
__global__ void prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                 float* __restrict__ y, size_t n) {
    const float s = a[0];
    GRID_STRIDE(i, n) {
        float v = x[i];
        y[i] = v > 0.f ? v : s * v;
    }
}
// dx = dy * (x>0 ? 1 : a);  per-block partial of da = sum(dy * x * [x<=0])
__global__ void prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                 const float* __restrict__ a, float* __restrict__ dx,
                                 float* __restrict__ part, size_t n) {
    __shared__ float red[256];
    const float s = a[0];
    float acc = 0.f;
    GRID_STRIDE(i, n) {
        float v = x[i], d = dy[i];
        dx[i] = v > 0.f ? d : s * d;
        if (!(v > 0.f)) acc += d * v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void sum_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ out,
                                    float scale) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * (double)scale);
}
#define REDUCE_BLOCKS 1024
MIGAN_API size_t migan_reduce_workspace() { return REDUCE_BLOCKS * sizeof(float); }

MIGAN_API int migan_prelu_fwd(const float* x, const float* a, float* y, size_t n, void* stream) {
    if (n == 0) return 0;
    MIGAN_LAUNCH(prelu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, a, y, n);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_prelu_bwd(const float* x, const float* dy, const float* a, float* dx, float* da,
                              float* ws, size_t n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int blocks = grid_for(n);
    if (blocks > REDUCE_BLOCKS) blocks = REDUCE_BLOCKS;
    MIGAN_LAUNCH(prelu_bwd_kernel, dim3(blocks), dim3(256), 0, st, x, dy, a, dx, ws, n);
    HIP_LAUNCH_CHECK();
    MIGAN_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, st, ws, blocks, da, 1.f);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ y = alpha*a + beta*b, y = a*b
__global__ void axpby_kernel(const float* __restrict__ a, float alpha, const float* __restrict__ b,
                             float beta, float* __restrict__ y, size_t n) {
    size_t n4 = n / 4;
    GRID_STRIDE(i, n4) {
        f32x4 u = reinterpret_cast<const f32x4*>(a)[i];
        f32x4 o;
        if (b) {
            f32x4 v = reinterpret_cast<const f32x4*>(b)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = alpha * u[k] + beta * v[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = alpha * u[k];
        }
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
    size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = alpha * a[t] + (b ? beta * b[t] : 0.f);
}
__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                           size_t n) {
    GRID_STRIDE(i, n) y[i] = a[i] * b[i];
}
// y[n][p][c] = x[n][p][c] * m[n][c]   (Dropout2d: one Bernoulli draw per (n,c) plane)
__global__ void mul_nc_kernel(const float* __restrict__ x, const float* __restrict__ m,
                              float* __restrict__ y, int HW, int C, size_t total) {
    GRID_STRIDE(i, total) {
        size_t pix, n;
        const int c = (int)divmod(i, (unsigned)C, pix);
        divmod(pix, (unsigned)HW, n);
        y[i] = x[i] * m[n * C + c];
    }
}
MIGAN_API int migan_axpby(const float* a, float alpha, const float* b, float beta, float* y, size_t n,
                          void* stream) {
    if (n == 0) return 0;
    MIGAN_LAUNCH(axpby_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, alpha, b,
                       beta, y, n);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_mul(const float* a, const float* b, float* y, size_t n, void* stream) {
    if (n == 0) return 0;
    MIGAN_LAUNCH(mul_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_mul_nc(const float* x, const float* m, float* y, int N, int HW, int C, void* stream) {
    size_t total = (size_t)N * HW * C;
    if (total == 0) return 0;
    MIGAN_LAUNCH(mul_nc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, m, y, HW, C,
                       total);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ dropout masks (Philox4x32-10)
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3,
                                             unsigned k0, unsigned k1) {
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    unsigned n1 = (unsigned)p1;
    unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
// mask[i] = u_i < p ? 0 : 1/(1-p).  The stream position comes from a device counter so a captured
// hipGraph draws fresh numbers on every replay.  counter[0] = position, counter[1] = arrival ticket (zero at rest):
// every block reads the position BEFORE it takes a ticket, and the block that takes the LAST ticket advances the
// position and resets the ticket - so no separate counter-update launch is needed.
__global__ void rand_mask_kernel(float* __restrict__ mask, size_t n, float p, unsigned long long seed,
                                 unsigned long long* counter) {
    const unsigned long long off = counter ? __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : 0ull;
    const float keep = 1.f / (1.f - p);
    size_t nq = (n + 3) / 4;
    GRID_STRIDE(i, nq) {
        unsigned long long ctr = off + i;
        unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0x9E3779B9u, c3 = 0xBB67AE85u;
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        unsigned rr[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            size_t e = i * 4 + k;
            if (e < n) {
                float u = (float)(rr[k] >> 8) * (1.0f / 16777216.0f);
                mask[e] = u < p ? 0.f : keep;
            }
        }
    }
    if (counter) {
        __syncthreads();  // all threads of this block have read the position
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned long long t = atomicAdd(counter + 1, 1ull);
            if (t == (unsigned long long)gridDim.x - 1) {
                counter[1] = 0ull;
                __hip_atomic_store(counter, off + nq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

MIGAN_API int migan_rand_mask(float* mask, size_t n, float p, unsigned long long seed,
                              unsigned long long* counter, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    MIGAN_LAUNCH(rand_mask_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, mask, n, p, seed, counter);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ gather2d: pad / upsample
// y[n][oh][ow][c] = x[n][map(oh-pad_t)][map(ow-pad_l)][c] or 0
__global__ void gather2d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int Hi, int Wi, int C,
                                    int Ho, int Wo, int pad_t, int pad_l, int mode, size_t total) {
    const int HiL = mode == GATHER_UP2 ? 2 * Hi : Hi, WiL = mode == GATHER_UP2 ? 2 * Wi : Wi;
    GRID_STRIDE(i, total) {
        size_t r, n;
        const int c = (int)divmod(i, (unsigned)C, r);
        const int ow = (int)divmod(r, (unsigned)(Wo), r);
        const int oh = (int)divmod(r, (unsigned)(Ho), n);
        int ih, iw;
        float v = 0.f;
        if (map_coord(oh - pad_t, HiL, mode, ih) && map_coord(ow - pad_l, WiL, mode, iw))
            v = x[((n * Hi + ih) * Wi + iw) * C + c];
        y[i] = v;
    }
}
// preimages of a source coordinate h under the map, in padded-output coordinates
__device__ __forceinline__ int preimages(int h, int H, int Hout, int pad, int mode, int* out) {
    int cnt = 0;
    if (mode == GATHER_UP2) {
        for (int a = 0; a < 2; ++a) {
            int o = 2 * h + a + pad;
            if (o >= 0 && o < Hout) out[cnt++] = o;
        }
        return cnt;
    }
    int o = h + pad;
    if (o >= 0 && o < Hout) out[cnt++] = o;
    if (mode == GATHER_REFLECT) {
        int o1 = pad - h;              // logical -h
        if (h >= 1 && o1 >= 0 && o1 < Hout) out[cnt++] = o1;
        int o2 = pad + 2 * (H - 1) - h;  // logical 2(H-1)-h
        if (h <= H - 2 && o2 >= 0 && o2 < Hout) out[cnt++] = o2;
    }
    return cnt;
}
__global__ void gather2d_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int Hi, int Wi,
                                    int C, int Ho, int Wo, int pad_t, int pad_l, int mode, size_t total) {
    GRID_STRIDE(i, total) {
        size_t r, n;
        const int c = (int)divmod(i, (unsigned)C, r);
        const int w = (int)divmod(r, (unsigned)(Wi), r);
        const int h = (int)divmod(r, (unsigned)(Hi), n);
        int ph[3], pw[3];
        int nh = preimages(h, Hi, Ho, pad_t, mode, ph);
        int nw = preimages(w, Wi, Wo, pad_l, mode, pw);
        float s = 0.f;
        for (int a = 0; a < nh; ++a)
            for (int b = 0; b < nw; ++b) s += dy[((n * Ho + ph[a]) * Wo + pw[b]) * C + c];
        dx[i] = s;
    }
}
MIGAN_API int migan_gather2d_fwd(const float* x, float* y, int N, int Hi, int Wi, int C, int Ho, int Wo,
                                 int pad_t, int pad_l, int mode, void* stream) {
    size_t total = (size_t)N * Ho * Wo * C;
    if (total == 0) return 0;
    MIGAN_LAUNCH(gather2d_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, Hi, Wi,
                       C, Ho, Wo, pad_t, pad_l, mode, total);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_gather2d_bwd(const float* dy, float* dx, int N, int Hi, int Wi, int C, int Ho, int Wo,
                                 int pad_t, int pad_l, int mode, void* stream) {
    size_t total = (size_t)N * Hi * Wi * C;
    if (total == 0) return 0;
    MIGAN_LAUNCH(gather2d_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, Hi,
                       Wi, C, Ho, Wo, pad_t, pad_l, mode, total);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ PixelShuffle(r), NHWC
// out[n][h*r+i][w*r+j][c] = in[n][h][w][c*r*r + i*r + j]   (in has C*r*r channels, out has C)
template <bool FWD>
__global__ void pixel_shuffle_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int C,
                                     int r, size_t total) {
    GRID_STRIDE(i, total) {  // i indexes the shuffled (large-spatial) tensor
        size_t q, n;
        const int c = (int)divmod(i, (unsigned)C, q);
        const int ow = (int)divmod(q, (unsigned)((W * r)), q);
        const int oh = (int)divmod(q, (unsigned)((H * r)), n);
        int h = oh / r, ii = oh - h * r, w = ow / r, jj = ow - w * r;
        size_t lo = ((n * H + h) * W + w) * ((size_t)C * r * r) + (size_t)c * r * r + ii * r + jj;
        if (FWD) dst[i] = src[lo];
        else dst[lo] = src[i];
    }
}
MIGAN_API int migan_pixel_shuffle(const float* src, float* dst, int N, int H, int W, int C, int r, int forward,
                                  void* stream) {
    size_t total = (size_t)N * H * W * C * r * r;
    if (total == 0) return 0;
    if (forward)
        MIGAN_LAUNCH((pixel_shuffle_kernel<true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           src, dst, H, W, C, r, total);
    else
        MIGAN_LAUNCH((pixel_shuffle_kernel<false>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           src, dst, H, W, C, r, total);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ MaxPool2d(2,2), NHWC (H, W even)
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C,
                                    size_t total) {
    const int Ho = H / 2, Wo = W / 2;
    GRID_STRIDE(i, total) {
        size_t q, n;
        const int c = (int)divmod(i, (unsigned)C, q);
        const int ow = (int)divmod(q, (unsigned)(Wo), q);
        const int oh = (int)divmod(q, (unsigned)(Ho), n);
        const float* p = x + ((n * H + 2 * oh) * W + 2 * ow) * C + c;
        float m = p[0];
        float v = p[C]; if (v > m) m = v;
        v = p[(size_t)W * C]; if (v > m) m = v;
        v = p[(size_t)W * C + C]; if (v > m) m = v;
        y[i] = m;
    }
}
// dx: first maximal element of each window (scan order h, w) receives dy (torch tie rule).  relu != 0: x is the output of a ReLU
// whose backward rides here - the maximum of a window is > 0 or the whole window is 0, where ReLU' is 0
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    float* __restrict__ dx, int H, int W, int C, size_t total, int relu) {
    const int Ho = H / 2, Wo = W / 2;
    GRID_STRIDE(i, total) {
        size_t q, n;
        const int c = (int)divmod(i, (unsigned)C, q);
        const int ow = (int)divmod(q, (unsigned)(Wo), q);
        const int oh = (int)divmod(q, (unsigned)(Ho), n);
        size_t b = ((n * H + 2 * oh) * W + 2 * ow) * C + c;
        size_t o[4] = {b, b + C, b + (size_t)W * C, b + (size_t)W * C + C};
        int arg = 0;
        float m = x[o[0]];
        for (int k = 1; k < 4; ++k) {
            float v = x[o[k]];
            if (v > m) { m = v; arg = k; }
        }
        float d = dy[i];
        if (relu && !(m > 0.f)) d = 0.f;
        for (int k = 0; k < 4; ++k) dx[o[k]] = (k == arg) ? d : 0.f;
    }
}
MIGAN_API int migan_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    size_t total = (size_t)N * (H / 2) * (W / 2) * C;
    if (total == 0) return 0;
    MIGAN_LAUNCH(maxpool2_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C,
                       total);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_maxpool2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C,
                                 void* stream) {
    size_t total = (size_t)N * (H / 2) * (W / 2) * C;
    if (total == 0) return 0;
    MIGAN_LAUNCH(maxpool2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, H,
                       W, C, total, 0);
    HIP_LAUNCH_CHECK();
    return 0;
}
// MaxPool2d(2) backward fused with the backward of the ReLU that produced x (vgg19.features[3:5], [8:10]: ReLU, MaxPool2d)
MIGAN_API int migan_maxpool2_relu_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C,
                                      void* stream) {
    size_t total = (size_t)N * (H / 2) * (W / 2) * C;
    if (total == 0) return 0;
    MIGAN_LAUNCH(maxpool2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, H,
                       W, C, total, 1);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ channel concat / split (NHWC)
// y[p][0:Ca] = a[p], y[p][Ca:Ca+Cb] = b[p]        (forward=1)   or the inverse scatter (forward=0)
template <bool FWD>
__global__ void cat_c_kernel(float* __restrict__ a, float* __restrict__ b, float* __restrict__ y, int Ca, int Cb,
                             size_t total) {
    const int C = Ca + Cb;
    GRID_STRIDE(i, total) {
        size_t p;
        const int c = (int)divmod(i, (unsigned)C, p);
        if (FWD) y[i] = c < Ca ? a[p * Ca + c] : b[p * Cb + (c - Ca)];
        else {
            if (c < Ca) a[p * Ca + c] = y[i];
            else b[p * Cb + (c - Ca)] = y[i];
        }
    }
}
MIGAN_API int migan_cat_channels(float* a, float* b, float* y, size_t P, int Ca, int Cb, int forward,
                                 void* stream) {
    size_t total = P * (size_t)(Ca + Cb);
    if (total == 0) return 0;
    if (forward)
        MIGAN_LAUNCH((cat_c_kernel<true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, b, y,
                           Ca, Cb, total);
    else
        MIGAN_LAUNCH((cat_c_kernel<false>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, b, y,
                           Ca, Cb, total);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ layout: NCHW <-> NHWC per image
// src viewed as [B][R][Cc] -> dst [B][Cc][R]  (32x32 LDS tiles, both sides coalesced)
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {
    __shared__ float tile[32][33];
    const size_t b = blockIdx.z;
    const float* s = src + b * (size_t)R * Cc;
    float* d = dst + b * (size_t)R * Cc;
    int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        int r = r0 + k, c = c0 + tx;
        if (r < R && c < Cc) tile[k][tx] = s[(size_t)r * Cc + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        int c = c0 + k, r = r0 + tx;
        if (r < R && c < Cc) d[(size_t)c * R + r] = tile[tx][k];
    }
}
MIGAN_API int migan_transpose_batched(const float* src, float* dst, int B, int R, int Cc, void* stream) {
    if ((size_t)B * R * Cc == 0) return 0;
    // grid.y is limited to 65535 blocks: R up to 2M rows
    dim3 grid(cdiv(Cc, 32), cdiv(R, 32), B);
    if (grid.y > 65535 || grid.z > 65535) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, R, Cc);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Row gather from two sources: dst[k][:] = sel[k] >= 0 ? a[sel[k]][:] : b[-1 - sel[k]][:]  (rows of D floats).
// Device-resident CycleGAN image history (cyclegan/utils.py:13-33): out[k] is either the new sample batch[k] or an old
// pool entry, and the pool update pool[j] = batch[k] is the same kernel with the roles swapped - two launches per
// push_and_pop instead of per-sample clones and a torch.cat.
__global__ void select_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst,
                                   const int* __restrict__ sel, const int* __restrict__ dst_row, size_t D) {
    const int k = blockIdx.y;
    const int sidx = sel[k];
    const float* src = sidx >= 0 ? a + (size_t)sidx * D : b + (size_t)(-1 - sidx) * D;
    const int drow = dst_row ? dst_row[k] : k;
    if (drow < 0) return;  // padding entry of a fixed-length table (a recorded hipGraph launches n = batch rows every step)
    float* out = dst + (size_t)drow * D;
    const size_t n4 = D / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(src)[i];
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < D; i += (size_t)gridDim.x * blockDim.x)
        out[i] = src[i];
}
// n rows; sel / dst_row are device int32 arrays (dst_row NULL: dst row k; dst_row[k] < 0: row k is skipped).  Rows must be 16-byte aligned (D % 4 == 0 or
// the scalar tail path handles the remainder only when the row starts are aligned, i.e. D % 4 == 0 for row > 0).
MIGAN_API int migan_select_rows(const float* a, const float* b, float* dst, const int* sel, const int* dst_row, int n,
                                size_t D, void* stream) {
    if (n <= 0 || D == 0) return 0;
    if (D % 4 != 0 || n > 65535) return (int)hipErrorInvalidValue;
    size_t bx = (D / 4 + 255) / 256;
    if (bx > 512) bx = 512;
    MIGAN_LAUNCH(select_rows_kernel, dim3((unsigned)bx, n), dim3(256), 0, (hipStream_t)stream, a, b, dst, sel, dst_row, D);
    HIP_LAUNCH_CHECK();
    return 0;
}

// generic 4-D permute copy (weight packing; small tensors): dst = src.permute(p0,p1,p2,p3).contiguous()
__global__ void permute4_kernel(const float* __restrict__ src, float* __restrict__ dst, int d0, int d1, int d2,
                                int d3, int p0, int p1, int p2, int p3, size_t total) {
    const int dims[4] = {d0, d1, d2, d3};
    const size_t st[4] = {(size_t)d1 * d2 * d3, (size_t)d2 * d3, (size_t)d3, 1};
    const int o0 = dims[p0], o1 = dims[p1], o2 = dims[p2], o3 = dims[p3];
    (void)o0;
    GRID_STRIDE(i, total) {
        size_t q, q0;
        const int i3 = (int)divmod(i, (unsigned)o3, q);
        const int i2 = (int)divmod(q, (unsigned)o2, q);
        const int i1 = (int)divmod(q, (unsigned)o1, q0);
        const int i0 = (int)q0;
        dst[i] = src[i0 * st[p0] + i1 * st[p1] + i2 * st[p2] + i3 * st[p3]];
    }
}
// The two weight packs of a convolution, OIHW -> OHWI (perm 0,2,3,1) and OIHW -> IHWO (perm 1,2,3,0), as LDS-tiled
// transposes.  With R = kh*kw both are  dst[y][r][x] = src[y*sy + x*sx + r]  (OHWI: y = o, x = i, sy = I*R, sx = R;
// IHWO: y = i, x = o, sy = R, sx = I*R): permute4_kernel reads them 4 B at a stride of R floats (27 us for a 4.2 M-element
// pix2pix weight, 22 such launches per step, profiles/r03_pix2pix_kernel_stats.txt).  A workgroup owns (y, 64 x): it reads
// the 64 runs of R contiguous floats (one contiguous 64*R block for OHWI), transposes through a padded LDS tile and writes
// R runs of 64 contiguous floats.  Measured (profiles/r03_abi_check.txt): 4 M / 8 M-element
// packs 17.6-41.5 -> 11.3 / 17.4 us (3.0 / 3.9 TB/s).
#define PTR_X 64
__global__ __launch_bounds__(256) void pack_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int X,
                                                             int R, long long sy, long long sx, int x_tiles) {
    extern __shared__ float ptr_tile[];   // [PTR_X][R + 1]
    const int y = blockIdx.x / x_tiles, x0 = (blockIdx.x - y * x_tiles) * PTR_X;
    const int nx = X - x0 < PTR_X ? X - x0 : PTR_X;
    const float* s = src + (size_t)y * sy + (size_t)x0 * sx;
    for (int e = threadIdx.x; e < nx * R; e += 256) {
        const int xl = e / R, r = e - xl * R;
        ptr_tile[xl * (R + 1) + r] = s[(size_t)xl * sx + r];
    }
    __syncthreads();
    float* d = dst + (size_t)y * R * X + x0;
    for (int e = threadIdx.x; e < R * PTR_X; e += 256) {
        const int r = e / PTR_X, xl = e - r * PTR_X;
        if (xl < nx) d[(size_t)r * X + xl] = ptr_tile[xl * (R + 1) + r];
    }
}

MIGAN_API int migan_permute4d(const float* src, float* dst, int d0, int d1, int d2, int d3, int p0, int p1, int p2,
                              int p3, void* stream) {
    size_t total = (size_t)d0 * d1 * d2 * d3;
    if (total == 0) return 0;
    const int R = d2 * d3;
    const bool ohwi = p0 == 0 && p1 == 2 && p2 == 3 && p3 == 1, ihwo = p0 == 1 && p1 == 2 && p2 == 3 && p3 == 0;
    if ((ohwi || ihwo) && R > 1 && R <= 96 && total >= (1u << 16) && total < (1ull << 31)) {
        const int X = ohwi ? d1 : d0, Y = ohwi ? d0 : d1;
        const long long sy = ohwi ? (long long)d1 * R : R, sx = ohwi ? R : (long long)d1 * R;
        const int x_tiles = (X + PTR_X - 1) / PTR_X;
        MIGAN_LAUNCH(pack_transpose_kernel, dim3((unsigned)Y * x_tiles), dim3(256), (size_t)PTR_X * (R + 1) * 4,
                           (hipStream_t)stream, src, dst, X, R, sy, sx, x_tiles);
        HIP_LAUNCH_CHECK();
        return 0;
    }
    MIGAN_LAUNCH(permute4_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, d0, d1,
                       d2, d3, p0, p1, p2, p3, total);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Multi-tensor form of permute4_kernel: ONE launch packs every weight of a training step (the OHWI / IHWO copies of all
// conv and linear weights: 456 permute launches per SRGAN step, 767 per ESRGAN step, 12 per DCGAN step otherwise).
// Table-driven like adam_kernel: block b handles PACK_CHUNK destination elements `chunk` of tensor `entry` (1024: the DCGAN
// step's 221 k pack elements are 216 workgroups; with 4096 they were 54 and the launch took 18.6 us).
struct PackEntry {
    const float* src;
    float* dst;
    unsigned o1, o2, o3, pad;  // destination extents of dims 1..3 (dim 0 is the quotient)
    long long s[4];            // source element stride of destination dim 0..3
    long long n;
};
struct PackBlock {
    int entry;
    int chunk;
};
#define PACK_CHUNK 1024
__global__ __launch_bounds__(256) void multi_permute4_kernel(const PackEntry* __restrict__ tab, const PackBlock* __restrict__ blk) {
    const PackBlock b = blk[blockIdx.x];
    const PackEntry* e = tab + b.entry;  // block-uniform: scalar loads, no per-thread copy of the entry
    const float* __restrict__ src = e->src;
    float* __restrict__ dst = e->dst;
    const unsigned o1 = e->o1, o2 = e->o2, o3 = e->o3;
    const size_t s0 = (size_t)e->s[0], s1 = (size_t)e->s[1], s2 = (size_t)e->s[2], s3 = (size_t)e->s[3];
    const size_t n = (size_t)e->n, base = (size_t)b.chunk * PACK_CHUNK;
#pragma unroll 4
    for (int k = threadIdx.x; k < PACK_CHUNK; k += 256) {
        const size_t i = base + k;
        if (i >= n) break;
        size_t q, q0;
        const unsigned i3 = divmod(i, o3, q);
        const unsigned i2 = divmod(q, o2, q);
        const unsigned i1 = divmod(q, o1, q0);
        dst[i] = src[q0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
    }
}
// entries / blocks: device arrays of PackEntry (72 bytes, see include/migan.h) and PackBlock {entry, chunk}; the caller lists
// ceil(n / 1024) blocks per entry
MIGAN_API int migan_multi_permute4d(const void* entries, const void* blocks, int nblocks, void* stream) {
    if (nblocks <= 0) return 0;
    MIGAN_LAUNCH(multi_permute4_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const PackEntry*)entries,
                       (const PackBlock*)blocks);
    HIP_LAUNCH_CHECK();
    return 0;
}

// p[0 .. bytes) = 0: optim.Adam.zero_grad() on the flat gradient bucket (dcgan.py:157,175).  A kernel of the library's own, 16 B per lane:
// hipMemsetAsync was measured first - as memset nodes of the recorded CycleGAN step (91 + 2 x 11 MB of buckets) it cost 4 ms per replay
// (36.4 vs 32.4 ms, profiles/r06_ab.txt call 20).  p must be 4-byte aligned, bytes a multiple of 4 (fp32 / int32 buffers).
__global__ __launch_bounds__(256) void zero_kernel(float* __restrict__ p, size_t n) {
    const size_t n4 = n >> 2;
    f32x4* q = reinterpret_cast<f32x4*>(p);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) q[i] = z;
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = 0.f;
}
MIGAN_API int migan_zero(void* p, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    if ((bytes & 3) || ((uintptr_t)p & 3)) return (int)hipErrorInvalidValue;
    const size_t n = bytes >> 2;
    if ((uintptr_t)p & 15) {   // unaligned head: a slower scalar pass (never the optimiser's buckets: 256-byte aligned)
        return (int)hipMemsetAsync(p, 0, bytes, (hipStream_t)stream);
    }
    long blocks = (long)((n / 4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    MIGAN_LAUNCH(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float*)p, n);
    HIP_LAUNCH_CHECK();
    return 0;
}
