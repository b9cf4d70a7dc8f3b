// BatchNorm2d/BatchNorm1d (train) and InstanceNorm2d, NHWC, fused with the following activation.
// Reference semantics: torch.nn.BatchNorm2d(C[, eps]) as used in dcgan.py:53,56,60,80 (eps 1e-5 and
// the positional eps=0.8), srgan/models.py:23,26,47,55,87,90, wgan_gp.py:49 (BatchNorm1d);
// torch.nn.InstanceNorm2d(C) (affine=False, eps 1e-5) in cyclegan/models.py:29,33,51,62,77,108 and
// pix2pix/models.py:25,40,117.  SURVEY.md Appendix B lists the exact formulas.
//
// Data is viewed as [G groups][P pixels][C channels]: BatchNorm has G=1,P=N*H*W; InstanceNorm has
// G=N,P=H*W.  All kernels are HBM-bound; loads are 16 B/lane when C%4==0.
#include "common.h"

// nn.PixelShuffle(2) behind the normalisation (srgan/models.py:53-57: Conv -> BatchNorm2d -> PixelShuffle -> PReLU) as the
// STORE index map of the apply kernel and the LOAD index map of dy in the backward kernels: a thread's 4 consecutive channels
// c = 4*co + 2*i + j of input pixel (n, h, w) are channel co of the 4 output pixels (n, 2h+i, 2w+j) - per instruction the
// lanes of a wave (consecutive co) still touch consecutive addresses.  No shuffled copy of the 604 MB tensor in either
// direction.  on == 0: plain [g][p][c] indexing.
struct PShuf {
    int on, H, W, Co;
    unsigned mg_hw, mg_w;
    int sh_hw, sh_w;
};
__device__ __forceinline__ size_t pshuf_base(const PShuf& s, int p, int co) {  // element index of (i, j) = (0, 0)
    const int n = fastdiv(p, s.mg_hw, s.sh_hw);
    const int rem = p - n * s.H * s.W;
    const int h = fastdiv(rem, s.mg_w, s.sh_w), w = rem - h * s.W;
    return ((size_t)(n * 2 * s.H + 2 * h) * (2 * s.W) + 2 * w) * s.Co + co;
}
__device__ __forceinline__ size_t pshuf_off(const PShuf& s, int k) {  // offset of sub-pixel k = 2*i + j from pshuf_base
    return ((size_t)(k >> 1) * (2 * s.W) + (k & 1)) * s.Co;
}
static PShuf pshuf_make(int H, int W, int C) {
    PShuf s = {};
    if (H > 0 && W > 0) {
        s.on = 1; s.H = H; s.W = W; s.Co = C / 4;
        fastdiv_magic((unsigned)(H * W), s.mg_hw, s.sh_hw);
        fastdiv_magic((unsigned)W, s.mg_w, s.sh_w);
    }
    return s;
}

// Per-block column sums: thread (tx, ty) holds the sums of its VW channels over its pixel lane; the ty lanes are added
// in a fixed order through LDS and lane ty == 0 writes slab[(g * gridDim.y + chunk)][c .. c+VW).
template <int VW>
__device__ __forceinline__ void colsum_slab_store(const float (&cs)[VW], float* red, float* __restrict__ csum, int tid,
                                                  int tx, int CTX, int TY, bool cok, int c, int C) {
#pragma unroll
    for (int v = 0; v < VW; ++v) red[tid * VW + v] = cs[v];
    __syncthreads();
    if (tid / CTX == 0 && cok) {
        float* out = csum + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c;
#pragma unroll
        for (int v = 0; v < VW; ++v) {
            float a = 0.f;
            for (int y = 0; y < TY; ++y) a += red[(y * CTX + tx) * VW + v];
            out[v] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// stats pass 1: per (g, chunk, c) partial (sum, sumsq) [and for backward: sum(dyz), sum(dyz*xhat)]
// thread layout: tx = tid % CTX walks channel vectors, ty = tid / CTX walks pixels.
// ---------------------------------------------------------------------------------------------
template <int VW, bool BWD>
__global__ __launch_bounds__(256) void norm_partial_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ dy,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ part, int P, int C, int CTX,
                                                           int chunk, int nchunks, int act, float slope,
                                                           const float* __restrict__ slope_ptr, const PShuf ps) {
    // slope_ptr (backward only): nn.PReLU()'s single learnable slope (srgan/models.py:24,57) fused behind the norm -
    // LeakyReLU with the slope read from the device, plus a third sum per (g, chunk, c): sum dy * min(z, 0) = d(loss)/d(slope)
    __shared__ float red[BWD ? 3 : 2][256 * VW];
    const bool prelu = BWD && slope_ptr != nullptr;
    if (prelu) {
        slope = *slope_ptr;
        act = ACT_LRELU;
    }
    const int tid = threadIdx.x;
    const int tx = tid % CTX, ty = tid / CTX, TY = 256 / CTX;
    const int g = blockIdx.z, ck = blockIdx.y;
    const int c = (blockIdx.x * CTX + tx) * VW;
    const bool cok = c < C;
    float s0[VW], s1[VW], s2[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) s0[v] = s1[v] = s2[v] = 0.f;
    float mu[VW], is[VW], ga[VW], be[VW];
    if (BWD && cok) {
#pragma unroll
        for (int v = 0; v < VW; ++v) {
            mu[v] = mean[(size_t)g * C + c + v];
            is[v] = invstd[(size_t)g * C + c + v];
            ga[v] = gamma ? gamma[c + v] : 1.f;
            be[v] = beta ? beta[c + v] : 0.f;
        }
    }
    int p0 = ck * chunk, p1 = p0 + chunk;
    if (p1 > P) p1 = P;
    const float* xb = x + (size_t)g * P * C;
    const float* dyb = BWD ? dy + (size_t)g * P * C : nullptr;
    // forward: sums are taken around K = the group's FIRST pixel (shifted-data algorithm: one sample of the
    // distribution as the origin keeps E[d^2]-E[d]^2 free of catastrophic cancellation); every chunk uses the same
    // K, so the finalize kernel only adds the partial sums (in double).
    float shift[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) shift[v] = (!BWD && cok) ? xb[c + v] : 0.f;
    if (cok) {
        // one pixel's contribution, in pixel order (the sums of a thread are sequential whatever the batching of the loads below)
        auto accum = [&](const float (&xv)[VW], const float (&dv)[VW]) {
#pragma unroll
            for (int v = 0; v < VW; ++v) {
                if (BWD) {
                    float xh = (xv[v] - mu[v]) * is[v];
                    float z = xh * ga[v] + be[v];
                    float d = dv[v];
                    if (prelu) s2[v] += z > 0.f ? 0.f : d * z;
                    if (act == ACT_LRELU) d *= (z > 0.f ? 1.f : slope);
                    else if (act == ACT_RELU) d = z > 0.f ? d : 0.f;
                    s0[v] += d;
                    s1[v] += d * xh;
                } else {
                    float d = xv[v] - shift[v];
                    s0[v] += d;
                    s1[v] += d * d;
                }
            }
        };
        auto load = [&](int p, float (&xv)[VW], float (&dv)[VW]) {
            if (VW == 4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(xb + (size_t)p * C + c);
                xv[0] = t[0]; xv[1] = t[1]; xv[2] = t[2]; xv[3] = t[3];
                if (BWD) {
                    if (ps.on) {
                        const size_t q = pshuf_base(ps, p, c >> 2);
#pragma unroll
                        for (int k = 0; k < 4; ++k) dv[k] = dy[q + pshuf_off(ps, k)];
                    } else {
                        f32x4 d = *reinterpret_cast<const f32x4*>(dyb + (size_t)p * C + c);
                        dv[0] = d[0]; dv[1] = d[1]; dv[2] = d[2]; dv[3] = d[3];
                    }
                }
            } else {
                xv[0] = xb[(size_t)p * C + c];
                if (BWD) dv[0] = dyb[(size_t)p * C + c];
            }
        };
        // Software-pipelined walk over the thread's pixels (round 6): batches of U pixels, the NEXT batch's loads issued before the current
        // batch is accumulated, so one memory round trip is exposed per workgroup instead of one per batch (a thread of SRGAN's trunk
        // walks 9 pixels = three dependent rounds of 4 + 4 + 1 before: 34 us for 75 MB, 2.2 TB/s, profiles/r06_ab.txt call 33).  Loads
        // past the chunk are clamped to its last pixel and their accumulation skipped; a thread adds its pixels in ascending order as before.
        constexpr int U = BWD ? 4 : 8;
        auto load_batch = [&](int pb, float (&xv)[U][VW], float (&dv)[U][VW]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = pb + u * TY;
                load(q < p1 ? q : p1 - 1, xv[u], dv[u]);
            }
        };
        auto accum_batch = [&](int pb, const float (&xv)[U][VW], const float (&dv)[U][VW]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (pb + u * TY < p1) accum(xv[u], dv[u]);
        };
        int p = p0 + ty;
        if (p < p1) {
            float xa[U][VW], da[U][VW], xq[U][VW], dq[U][VW];
            load_batch(p, xa, da);
            for (; p < p1; p += 2 * U * TY) {
                const bool more = p + U * TY < p1;
                if (more) load_batch(p + U * TY, xq, dq);
                accum_batch(p, xa, da);
                if (!more) break;
                if (p + 2 * U * TY < p1) load_batch(p + 2 * U * TY, xa, da);
                accum_batch(p + U * TY, xq, dq);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VW; ++v) {
        red[0][tid * VW + v] = s0[v];
        red[1][tid * VW + v] = s1[v];
        if (BWD) red[2][tid * VW + v] = s2[v];
    }
    __syncthreads();
    if (ty == 0 && cok) {
#pragma unroll
        for (int v = 0; v < VW; ++v) {
            float a = 0.f, b = 0.f, c2 = 0.f;
            for (int y = 0; y < TY; ++y) {
                a += red[0][(y * CTX + tx) * VW + v];
                b += red[1][(y * CTX + tx) * VW + v];
                if (BWD) c2 += red[2][(y * CTX + tx) * VW + v];
            }
            // (a chunk-minor layout - contiguous reads for the finalize wave, scattered writes here - was measured: finalize
            // 6.9 -> 6.3 us, this kernel 13.4 -> 14.9 us; a wash, the finalize launches sit on the ~5 us small-kernel floor)
            size_t o = (((size_t)g * nchunks + ck) * C + c + v) * 3;
            part[o] = a;
            part[o + 1] = b;
            part[o + 2] = BWD ? c2 : shift[v];
        }
    }
}

// The finalize kernels are one wave per (group, channel) walking `nchunks` partial records 64 at a time: with one record per lane per
// round a 1024-chunk reduction was 16 dependent round trips (8.5 us for a launch whose floor is 2.4).  NF_B records per lane are loaded
// per round - the record index is clamped, not branched on, the add is guarded: same order, same sums.
#define NF_B 4

// pass 2 (forward): ONE WAVE per (g,c) adds the per-chunk (sum d, sum d^2) pairs in double (lanes stride over
// chunks, butterfly reduce), writes mean / invstd and updates the running statistics.
// chain (BatchNorm over G consecutive sub-batches in one launch - e.g. D(real) and D(fake) of dcgan.py:176-177 run as one
// batch of 2 x 128): one wave per channel walks the groups in order and applies the G running-statistics updates one after
// the other, exactly as G separate forward calls would; num_batches_tracked grows by G.
__global__ __launch_bounds__(256) void norm_finalize_fwd_kernel(const float* __restrict__ part,
                                                                float* __restrict__ mean,
                                                                float* __restrict__ invstd, float* __restrict__ var_out,
                                                                float* running_mean,
                                                                float* running_var, long long* nbt, int G, int P,
                                                                int C, int nchunks, int chunk, float eps,
                                                                float momentum, int chain) {
    const int w = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (w >= (chain ? C : G * C)) return;  // wave-uniform
    const int g0 = chain ? 0 : w / C, c = chain ? w : w - g0 * C, g1 = chain ? G : g0 + 1;
    for (int g = g0; g < g1; ++g) {
        const int i = g * C + c;
        double sd = 0.0, sq = 0.0;
        for (int k = lane; k < nchunks; k += NF_B * 64) {   // NF_B chunks per round of loads (see NF_B)
            float v[NF_B][2];
#pragma unroll
            for (int u = 0; u < NF_B; ++u) {
                const int kk = k + u * 64;
                const size_t o = (((size_t)g * nchunks + (kk < nchunks ? kk : nchunks - 1)) * C + c) * 3;
                v[u][0] = part[o];
                v[u][1] = part[o + 1];
            }
#pragma unroll
            for (int u = 0; u < NF_B; ++u)
                if (k + u * 64 < nchunks) {
                    sd += (double)v[u][0];
                    sq += (double)v[u][1];
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sd += __shfl_xor(sd, off);
            sq += __shfl_xor(sq, off);
        }
        if (lane != 0) continue;
        if (c == 0 && nbt && (chain || g == 0)) nbt[0] += 1;  // BatchNorm's num_batches_tracked
        const double K = (double)part[(((size_t)g * nchunks) * C + c) * 3 + 2];
        const double md = sd / P;
        double M2 = sq - sd * md;
        if (M2 < 0.0) M2 = 0.0;
        const double m = K + md;
        double var = M2 / P;
        mean[i] = (float)m;
        if (invstd) invstd[i] = (float)(1.0 / sqrt(var + (double)eps));
        if (var_out) var_out[i] = (float)var;
        if (running_mean && (G == 1 || chain)) {
            double unb = P > 1 ? M2 / (double)(P - 1) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
}

// pass 2 (backward): one wave per (g,c) -> sums[g][c][2] = (sum dyz, sum dyz*xhat); dgamma/dbeta for G==1, or summed over the
// groups in group order when chain != 0 (see the forward kernel).
__global__ __launch_bounds__(256) void norm_finalize_bwd_kernel(const float* __restrict__ part,
                                                                float* __restrict__ sums, float* dgamma,
                                                                float* dbeta, int G, int C, int nchunks,
                                                                int accum, float* __restrict__ dslope_gc, int chain) {
    const int w = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (w >= (chain ? C : G * C)) return;
    const int g0 = chain ? 0 : w / C, c = chain ? w : w - g0 * C, g1 = chain ? G : g0 + 1;
    float ta = 0.f, tb = 0.f;
    for (int g = g0; g < g1; ++g) {
        const int i = g * C + c;
        double a = 0.0, b = 0.0, s = 0.0;
        for (int k = lane; k < nchunks; k += NF_B * 64) {
            float v[NF_B][3];
#pragma unroll
            for (int u = 0; u < NF_B; ++u) {
                const int kk = k + u * 64;
                const size_t o = (((size_t)g * nchunks + (kk < nchunks ? kk : nchunks - 1)) * C + c) * 3;
                v[u][0] = part[o];
                v[u][1] = part[o + 1];
                v[u][2] = dslope_gc ? part[o + 2] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < NF_B; ++u)
                if (k + u * 64 < nchunks) {
                    a += (double)v[u][0];
                    b += (double)v[u][1];
                    if (dslope_gc) s += (double)v[u][2];
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_xor(a, off);
            b += __shfl_xor(b, off);
            if (dslope_gc) s += __shfl_xor(s, off);
        }
        if (lane != 0) continue;
        if (dslope_gc) dslope_gc[i] = (float)s;  // per (g, c) PReLU-slope partials, summed by sum_small_kernel
        sums[(size_t)i * 2] = (float)a;
        sums[(size_t)i * 2 + 1] = (float)b;
        ta += (float)a;  // fp32 sum over the groups = what accumulating G separate backward calls into .grad does
        tb += (float)b;
    }
    if (lane == 0 && (G == 1 || chain)) {
        if (dbeta) dbeta[c] = accum ? dbeta[c] + ta : ta;
        if (dgamma) dgamma[c] = accum ? dgamma[c] + tb : tb;
    }
}

// out[0] (+)= sum of n floats, one block, fixed order (the PReLU slope gradient from its per-(g,c) partials)
__global__ __launch_bounds__(256) void sum_small_kernel(const float* __restrict__ v, int n, float* __restrict__ out, int accum) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += (double)v[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s_ = 128; s_ > 0; s_ >>= 1) {
        if ((int)threadIdx.x < s_) red[threadIdx.x] += red[threadIdx.x + s_];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = accum ? out[0] + (float)red[0] : (float)red[0];
}

// apply: y = act((x-mean)*invstd*gamma+beta) [+ res]
// Thread layout as in the statistics pass: tx owns VW fixed channels (its scale/shift live in registers, so the
// loop body is load - fma - act - store with no per-element parameter gathers or index divisions), ty strides over
// the pixels of the block's chunk.  A wave's accesses are contiguous runs of CTX*VW floats per pixel.
template <int VW>
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ res, int P, int C, int CTX,
                                                         int chunk, int act, float slope,
                                                         const float* __restrict__ slope_ptr, const PShuf ps) {
    if (slope_ptr) {  // fused nn.PReLU(): LeakyReLU with the learnable slope read from the device
        slope = *slope_ptr;
        act = ACT_LRELU;
    }
    const int tid = threadIdx.x;
    const int tx = tid % CTX, ty = tid / CTX, TY = 256 / CTX;
    const int g = blockIdx.z;
    const int c = (blockIdx.x * CTX + tx) * VW;
    if (c >= C) return;
    float sc[VW], sh[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) {
        const float is = invstd[(size_t)g * C + c + v], mu = mean[(size_t)g * C + c + v];
        sc[v] = is * (gamma ? gamma[c + v] : 1.f);
        sh[v] = (beta ? beta[c + v] : 0.f) - mu * sc[v];
    }
    const int p0 = blockIdx.y * chunk;
    int p1 = p0 + chunk;
    if (p1 > P) p1 = P;
    const size_t base = (size_t)g * P * C + c;
#pragma unroll 4
    for (int p = p0 + ty; p < p1; p += TY) {
        const size_t e = base + (size_t)p * C;
        if (VW == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + e);
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (res) r = *reinterpret_cast<const f32x4*>(res + e);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = act_apply(fmaf(v[k], sc[k], sh[k]), act, slope) + r[k];
            if (ps.on) {
                const size_t q = pshuf_base(ps, p, c >> 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) y[q + pshuf_off(ps, k)] = o[k];
            } else {
                *reinterpret_cast<f32x4*>(y + e) = o;
            }
        } else {
            y[e] = act_apply(fmaf(x[e], sc[0], sh[0]), act, slope) + (res ? res[e] : 0.f);
        }
    }
}

// dx = gamma*invstd*(dyz - s0/P - xhat*s1/P); dyz = dy * act'(z)        (same thread layout as norm_apply_kernel)
template <int VW>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ sums, int P, int C, int CTX, int chunk, int act,
    float slope, float invP, float* __restrict__ csum, const float* __restrict__ slope_ptr, const PShuf ps) {
    __shared__ float red[256 * VW];
    if (slope_ptr) {
        slope = *slope_ptr;
        act = ACT_LRELU;
    }
    const int tid = threadIdx.x;
    const int tx = tid % CTX, ty = tid / CTX, TY = 256 / CTX;
    const int g = blockIdx.z;
    const int c = (blockIdx.x * CTX + tx) * VW;
    const bool cok = c < C;
    if (!cok && !csum) return;
    float mu[VW], is[VW], ga[VW], be[VW], k0[VW], k1[VW], cs[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) {
        cs[v] = 0.f;
        const size_t gc = (size_t)g * C + (cok ? c + v : 0);
        mu[v] = mean[gc];
        is[v] = invstd[gc];
        ga[v] = (gamma && cok) ? gamma[c + v] : 1.f;
        be[v] = (beta && cok) ? beta[c + v] : 0.f;
        k0[v] = sums[gc * 2] * invP;
        k1[v] = sums[gc * 2 + 1] * invP;
    }
    const int p0 = blockIdx.y * chunk;
    int p1 = cok ? p0 + chunk : p0;
    if (p1 > P) p1 = P;
    const size_t base = (size_t)g * P * C + c;
#pragma unroll 4
    for (int p = p0 + ty; p < p1; p += TY) {
        const size_t e = base + (size_t)p * C;
        float xv[VW], dv[VW], ov[VW];
        if (VW == 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + e);
            if (ps.on) {
                const size_t q = pshuf_base(ps, p, c >> 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) { xv[k] = a[k]; dv[k] = dy[q + pshuf_off(ps, k)]; }
            } else {
                const f32x4 b = *reinterpret_cast<const f32x4*>(dy + e);
#pragma unroll
                for (int k = 0; k < 4; ++k) { xv[k] = a[k]; dv[k] = b[k]; }
            }
        } else {
            xv[0] = x[e];
            dv[0] = dy[e];
        }
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            const float xh = (xv[k] - mu[k]) * is[k];
            float d = dv[k];
            if (act != ACT_NONE) {
                const float z = xh * ga[k] + be[k];
                if (act == ACT_LRELU) d *= (z > 0.f ? 1.f : slope);
                else if (act == ACT_RELU) d = z > 0.f ? d : 0.f;
            }
            ov[k] = ga[k] * is[k] * (d - k0[k] - xh * k1[k]);
            cs[k] += ov[k];
        }
        if (VW == 4) {
            const f32x4 o = {ov[0], ov[1], ov[2], ov[3]};
            *reinterpret_cast<f32x4*>(dx + e) = o;
        } else {
            dx[e] = ov[0];
        }
    }
    if (csum) colsum_slab_store<VW>(cs, red, csum, tid, tx, CTX, TY, cok, c, C);
}

// dx = dy * [mask[g][c]] * act'(y) (y = the activation output, masked when a Dropout2d mask is given) AND the
// per-block column sums of dx: the backward of `Conv2d -> act [-> Dropout2d]` produces the gradient the conv's dgrad /
// wgrad consume and the slabs its bias gradient is reduced from in ONE pass over dy (a separate two-launch column sum
// re-read the tensor this kernel had just written).  Same thread layout as norm_apply_kernel.
template <int VW>
__global__ __launch_bounds__(256) void act_bwd_colsum_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ mask, float* __restrict__ dx,
                                                             float* __restrict__ csum, int P, int C, int CTX, int chunk,
                                                             int act, float slope) {
    __shared__ float red[256 * VW];
    const int tid = threadIdx.x;
    const int tx = tid % CTX, ty = tid / CTX, TY = 256 / CTX;
    const int g = blockIdx.z;
    const int c = (blockIdx.x * CTX + tx) * VW;
    const bool cok = c < C;
    float mk[VW], cs[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) {
        cs[v] = 0.f;
        mk[v] = (mask && cok) ? mask[(size_t)g * C + c + v] : 1.f;
    }
    const int p0 = blockIdx.y * chunk;
    int p1 = cok ? p0 + chunk : p0;
    if (p1 > P) p1 = P;
    const size_t base = (size_t)g * P * C + c;
#pragma unroll 4
    for (int p = p0 + ty; p < p1; p += TY) {
        const size_t e = base + (size_t)p * C;
        if (VW == 4) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(dy + e);
            f32x4 o;
            if (act != ACT_NONE) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(y + e);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = d[k] * mk[k] * act_grad_from_out(v[k], act, slope);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = d[k] * mk[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) cs[k] += o[k];
            *reinterpret_cast<f32x4*>(dx + e) = o;
        } else {
            const float o = dy[e] * mk[0] * (act != ACT_NONE ? act_grad_from_out(y[e], act, slope) : 1.f);
            cs[0] += o;
            dx[e] = o;
        }
    }
    colsum_slab_store<VW>(cs, red, csum, tid, tx, CTX, TY, cok, c, C);
}

// ---------------------------------------------------------------------------------------------
static void norm_plan(int G, int P, int C, int& VW, int& CTX, int& chunk, int& nchunks, int& gx) {
    VW = (C % 4 == 0) ? 4 : 1;
    int cv = C / VW;
    CTX = 1;
    while (CTX < cv && CTX < 64) CTX <<= 1;
    gx = cdiv(cv, CTX);
    int TY = 256 / CTX;
    // ~512 blocks in total (two per CU), at least 4 pixels per ty lane.  Round 6, with the pipelined pixel walk of norm_partial_kernel
    // (tools/norm_microbench.py, profiles/r06_ab.txt call 36): 1024 -> 512 blocks = statistics 15.6 -> 12.7 us and backward 49.3 -> 45.8 us
    // on SRGAN's trunk tensor, 307 -> 283 us on its 604 MB discriminator tensor, 124.6 -> 121.4 / 64.9 -> 61.8 us on DCGAN's; 256 and 128
    // blocks lose on the backward (the sums pass no longer covers the chip), 2048 / 4096 lose everywhere.
    long blocks_other = (long)gx * G;
    long want = cdiv((long)MIGAN_KNOB("MIGAN_NORM_CHUNKS", 512), blocks_other);
    long maxc = cdiv(P, (long)TY * 4);
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    chunk = cdiv(P, want);
    nchunks = cdiv(P, chunk);
}

MIGAN_API size_t migan_norm_workspace(int G, int P, int C) {
    int VW, CTX, chunk, nchunks, gx;
    norm_plan(G, P, C, VW, CTX, chunk, nchunks, gx);
    return ((size_t)G * nchunks * C * 3 + (size_t)G * C * 2) * sizeof(float);
}

// pixel chunking of the streaming apply kernels: ~4096 blocks, >= 8 pixels per ty lane
static void apply_plan(int G, int P, int C, int& VW, int& CTX, int& chunk, dim3& grid) {
    int nchunks_stats, chunk_stats, gx;
    norm_plan(G, P, C, VW, CTX, chunk_stats, nchunks_stats, gx);
    int TY = 256 / CTX;
    long want = cdiv((long)MIGAN_KNOB("MIGAN_NORM_APPLY_CHUNKS", 4096), (long)gx * G);
    long maxc = cdiv(P, (long)TY * 8);
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    chunk = cdiv(P, want);
    grid = dim3(gx, cdiv(P, chunk), G);
}

// ---------------------------------------------------------------------------------------------
// Small instance-style tensors (no running statistics; P <= 1024 pixels per group): statistics AND normalisation in ONE launch.
// The inner levels of the pix2pix U-Net (pix2pix/models.py:25,41: InstanceNorm2d on 512 x 2x2 .. 256 x 32x32 at batch 1), its
// PatchGAN and CycleGAN at one image per GPU are tensors of 2 K - 256 K elements: the three launches of the streaming path
// (statistics, finalize, apply; three more in backward) are ~5 us of latency each for microseconds of work - 138 of the 337
// launches of a pix2pix step.  Here a workgroup owns 16 channels of one group and ALL its pixels (64 pixel lanes x 4 channel
// quads): pass 1 sums around the group's first pixel (the shifted-data form of the streaming kernels), the 64 lanes are combined
// in a fixed order in double, pass 2 re-reads the (L2-resident) pixels and writes.  No cross-workgroup reduction exists.
// Measured (profiles/r03_abi_check.txt): 6.3-18.6 us per launch, 4e-8 / 5e-8 from a host fp64 evaluation.
// ---------------------------------------------------------------------------------------------
// (Round 6 measured the same kernels with 4 channels per workgroup for 1025 .. 4096 pixels per group - the 64 x 64 maps of CycleGAN's trunk at
// one image per GPU, 19 of the 23 InstanceNorm layers of a generator pass: the recorded step went from 33.0 to 38.0 ms, pix2pix from 2.93 to 3.05
// (profiles/r06_ab.txt calls 25, 26).  A workgroup's lanes then read 16 bytes at a 1 KB stride - an eighth of every cache line - and 64
// workgroups walk 64 KB each: the three streaming launches with channel-contiguous lanes on the whole chip are faster.  Removed.)
#define NS_CH 16
__global__ __launch_bounds__(256) void norm_small_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             float* __restrict__ mean, float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ res, int P, int C, int act, float slope,
                                                             float eps, const float* __restrict__ mask) {
    // mask (optional, same [G][P][C] layout): the nn.Dropout behind the activation (pix2pix/models.py:27,44) - y = act(..) * mask
    __shared__ float red[2][256 * 4];
    __shared__ float st[2][NS_CH];
    const int tid = threadIdx.x, tx = tid & 3, ty = tid >> 2;
    const int g = blockIdx.z, c = blockIdx.x * NS_CH + tx * 4;
    const float* xb = x + (size_t)g * P * C + c;
    const f32x4 shift = *reinterpret_cast<const f32x4*>(xb);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    for (int p = ty; p < P; p += 64) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(xb + (size_t)p * C) - shift;
        s0 += d;
        s1 += d * d;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        red[0][tid * 4 + v] = s0[v];
        red[1][tid * 4 + v] = s1[v];
    }
    __syncthreads();
    if (tid < NS_CH) {  // thread = channel tid of the slab: quad tid >> 2, element tid & 3
        const int q = tid >> 2, v = tid & 3;
        double sd = 0.0, sq = 0.0;
        for (int l = 0; l < 64; ++l) {
            sd += (double)red[0][(l * 4 + q) * 4 + v];
            sq += (double)red[1][(l * 4 + q) * 4 + v];
        }
        const double K = (double)xb[tid - tx * 4];   // the group's first pixel, channel blockIdx.x * 16 + tid
        const double md = sd / P;
        double M2 = sq - sd * md;
        if (M2 < 0.0) M2 = 0.0;
        const float m = (float)(K + md), is = (float)(1.0 / sqrt(M2 / P + (double)eps));
        st[0][tid] = m;
        st[1][tid] = is;
        mean[(size_t)g * C + blockIdx.x * NS_CH + tid] = m;
        invstd[(size_t)g * C + blockIdx.x * NS_CH + tid] = is;
    }
    __syncthreads();
    float sc[4], sh[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        sc[v] = st[1][tx * 4 + v] * (gamma ? gamma[c + v] : 1.f);
        sh[v] = (beta ? beta[c + v] : 0.f) - st[0][tx * 4 + v] * sc[v];
    }
    float* yb = y + (size_t)g * P * C + c;
    const float* rb = res ? res + (size_t)g * P * C + c : nullptr;
    const float* mb = mask ? mask + (size_t)g * P * C + c : nullptr;
    for (int p = ty; p < P; p += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (size_t)p * C);
        f32x4 r = {0.f, 0.f, 0.f, 0.f}, mk = {1.f, 1.f, 1.f, 1.f};
        if (rb) r = *reinterpret_cast<const f32x4*>(rb + (size_t)p * C);
        if (mb) mk = *reinterpret_cast<const f32x4*>(mb + (size_t)p * C);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = act_apply(fmaf(v[k], sc[k], sh[k]), act, slope) * mk[k] + r[k];
        *reinterpret_cast<f32x4*>(yb + (size_t)p * C) = o;
    }
}
// backward: dx = gamma*invstd*(dyz - s0/P - xhat*s1/P), (s0, s1) = sums over the group's pixels of (dyz, dyz*xhat), dyz = dy*act'(z);
// csum (optional): the column-sum slabs of dx the preceding conv's bias gradient is reduced from - row (g, 0) holds the group's
// sums, rows (g, 1 .. rows_per_g - 1) are zeroed (the consumer adds all migan_norm_colsum_slabs() rows)
__global__ __launch_bounds__(256) void norm_small_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ dx, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int P, int C, int act, float slope,
                                                             float* __restrict__ csum, int rows_per_g, const float* __restrict__ mask) {
    __shared__ float red[2][256 * 4];
    __shared__ float st[2][NS_CH];
    const int tid = threadIdx.x, tx = tid & 3, ty = tid >> 2;
    const int g = blockIdx.z, c = blockIdx.x * NS_CH + tx * 4;
    const size_t base = (size_t)g * P * C + c;
    float mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        mu[v] = mean[(size_t)g * C + c + v];
        is[v] = invstd[(size_t)g * C + c + v];
        ga[v] = gamma ? gamma[c + v] : 1.f;
        be[v] = beta ? beta[c + v] : 0.f;
    }
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    for (int p = ty; p < P; p += 64) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + (size_t)p * C);
        f32x4 dv = *reinterpret_cast<const f32x4*>(dy + base + (size_t)p * C);
        if (mask) dv *= *reinterpret_cast<const f32x4*>(mask + base + (size_t)p * C);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - mu[k]) * is[k];
            float d = dv[k];
            if (act != ACT_NONE) {
                const float z = xh * ga[k] + be[k];
                if (act == ACT_LRELU) d *= (z > 0.f ? 1.f : slope);
                else if (act == ACT_RELU) d = z > 0.f ? d : 0.f;
            }
            s0[k] += d;
            s1[k] += d * xh;
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        red[0][tid * 4 + v] = s0[v];
        red[1][tid * 4 + v] = s1[v];
    }
    __syncthreads();
    if (tid < NS_CH) {
        const int q = tid >> 2, v = tid & 3;
        double a = 0.0, b = 0.0;
        for (int l = 0; l < 64; ++l) {
            a += (double)red[0][(l * 4 + q) * 4 + v];
            b += (double)red[1][(l * 4 + q) * 4 + v];
        }
        st[0][tid] = (float)a / (float)P;
        st[1][tid] = (float)b / (float)P;
    }
    __syncthreads();
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    for (int p = ty; p < P; p += 64) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + (size_t)p * C);
        f32x4 dv = *reinterpret_cast<const f32x4*>(dy + base + (size_t)p * C);
        if (mask) dv *= *reinterpret_cast<const f32x4*>(mask + base + (size_t)p * C);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - mu[k]) * is[k];
            float d = dv[k];
            if (act != ACT_NONE) {
                const float z = xh * ga[k] + be[k];
                if (act == ACT_LRELU) d *= (z > 0.f ? 1.f : slope);
                else if (act == ACT_RELU) d = z > 0.f ? d : 0.f;
            }
            o[k] = ga[k] * is[k] * (d - st[0][tx * 4 + k] - xh * st[1][tx * 4 + k]);
        }
        cs += o;
        *reinterpret_cast<f32x4*>(dx + base + (size_t)p * C) = o;
    }
    if (csum) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 4; ++v) red[0][tid * 4 + v] = cs[v];
        __syncthreads();
        if (tid < NS_CH) {
            const int q = tid >> 2, v = tid & 3;
            float a = 0.f;
            for (int l = 0; l < 64; ++l) a += red[0][(l * 4 + q) * 4 + v];
            const int ch = blockIdx.x * NS_CH + tid;
            csum[((size_t)g * rows_per_g) * C + ch] = a;
            for (int r = 1; r < rows_per_g; ++r) csum[((size_t)g * rows_per_g + r) * C + ch] = 0.f;
        }
    }
}
static bool norm_small_ok(int G, int P, int C) {
    return C % NS_CH == 0 && P >= 2 && P <= 1024 && (long)G * (C / NS_CH) >= 8 && (long)G * P * C <= (1L << 20) && G <= 65535;
}
// 1: migan_norm_fwd_small takes the shape (instance-style statistics: no running statistics, no cross-replica exchange)
MIGAN_API int migan_norm_small_ok(int G, int P, int C) { return norm_small_ok(G, P, C) ? 1 : 0; }
// Statistics + normalisation (+ affine, activation, residual) in one launch; mean / invstd [G][C] are written for the backward.
MIGAN_API int migan_norm_fwd_small(const float* x, float* y, float* mean, float* invstd, const float* gamma, const float* beta,
                                   const float* res, const float* mask, int G, int P, int C, int act, float slope, float eps,
                                   void* stream) {
    if (!norm_small_ok(G, P, C)) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(norm_small_fwd_kernel, dim3(C / NS_CH, 1, G), dim3(256), 0, (hipStream_t)stream, x, y, mean, invstd, gamma,
                       beta, res, P, C, act, slope, eps, mask);
    HIP_LAUNCH_CHECK();
    return 0;
}
// Backward of migan_norm_fwd_small with a dropout mask (dz = dy * mask * act'(z)); csum as for migan_norm_bwd (may be NULL)
MIGAN_API int migan_norm_bwd_small(const float* x, const float* dy, const float* mask, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, float* dx, int G, int P, int C, int act, float slope,
                                   float* csum, void* stream);

static int norm_stats_impl(const float* x, float* mean, float* invstd, float* var_out, float* running_mean,
                           float* running_var, long long* num_batches_tracked, float momentum, float eps, int G, int P,
                           int C, float* ws, size_t ws_bytes, hipStream_t st) {
    int VW, CTX, chunk, nchunks, gx;
    norm_plan(G, P, C, VW, CTX, chunk, nchunks, gx);
    if (ws_bytes < migan_norm_workspace(G, P, C)) return (int)hipErrorInvalidValue;
    dim3 grid(gx, nchunks, G);
    if (VW == 4)
        MIGAN_LAUNCH((norm_partial_kernel<4, false>), grid, dim3(256), 0, st, x, nullptr, nullptr,
                           nullptr, nullptr, nullptr, ws, P, C, CTX, chunk, nchunks, 0, 0.f, nullptr, PShuf{});
    else
        MIGAN_LAUNCH((norm_partial_kernel<1, false>), grid, dim3(256), 0, st, x, nullptr, nullptr,
                           nullptr, nullptr, nullptr, ws, P, C, CTX, chunk, nchunks, 0, 0.f, nullptr, PShuf{});
    HIP_LAUNCH_CHECK();
    const int chain = (G > 1 && running_mean != nullptr) ? 1 : 0;  // BatchNorm over G sub-batches (InstanceNorm has no running stats)
    MIGAN_LAUNCH(norm_finalize_fwd_kernel, dim3(cdiv((long)(chain ? 1 : G) * C * 64, 256)), dim3(256), 0, st, ws, mean,
                       invstd, var_out, running_mean, running_var, num_batches_tracked, G, P, C, nchunks, chunk, eps,
                       momentum, chain);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Training-mode statistics: mean/invstd [G][C] (+ running stat update when G==1 and pointers given).
MIGAN_API int migan_norm_stats(const float* x, float* mean, float* invstd, float* running_mean,
                               float* running_var, long long* num_batches_tracked, float momentum, float eps,
                               int G, int P, int C, float* ws, size_t ws_bytes, void* stream) {
    return norm_stats_impl(x, mean, invstd, nullptr, running_mean, running_var, num_batches_tracked, momentum, eps, G, P,
                           C, ws, ws_bytes, (hipStream_t)stream);
}

// Local moments only (mean and BIASED variance of this rank's shard): first half of cross-replica BatchNorm.
MIGAN_API int migan_norm_moments(const float* x, float* mean, float* var, int G, int P, int C, float* ws,
                                 size_t ws_bytes, void* stream) {
    return norm_stats_impl(x, mean, nullptr, var, nullptr, nullptr, nullptr, 0.f, 0.f, G, P, C, ws, ws_bytes,
                           (hipStream_t)stream);
}

// Second half: `gathered` = [world][2][C] (mean, biased variance) of `world` EQUAL shards of P_local pixels each
// (all_gather of the migan_norm_moments outputs).  Chan's parallel combination in double -> global-batch mean / invstd,
// running statistics updated with the global unbiased variance: exactly what the single-process reference computes on
// the whole batch (dcgan.py:53-60, srgan/models.py:23-26, wgan_gp.py:49).
__global__ void norm_sync_finalize_kernel(const float* __restrict__ gathered, int world, double P_local,
                                          float* __restrict__ mean, float* __restrict__ invstd, float* running_mean,
                                          float* running_var, long long* nbt, float momentum, float eps, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) nbt[0] += 1;
    if (c >= C) return;
    double m = 0.0;
    for (int r = 0; r < world; ++r) m += (double)gathered[((size_t)r * 2) * C + c];
    m /= world;
    double M2 = 0.0;
    for (int r = 0; r < world; ++r) {
        const double mr = (double)gathered[((size_t)r * 2) * C + c], vr = (double)gathered[((size_t)r * 2 + 1) * C + c];
        M2 += P_local * (vr + (mr - m) * (mr - m));
    }
    const double Pt = P_local * world;
    const double var = M2 / Pt;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = Pt > 1.0 ? M2 / (Pt - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}
MIGAN_API int migan_norm_sync_finalize(const float* gathered, int world, long long P_local, float* mean, float* invstd,
                                       float* running_mean, float* running_var, long long* num_batches_tracked,
                                       float momentum, float eps, int C, void* stream) {
    if (world < 1 || C < 1) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(norm_sync_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gathered, world,
                       (double)P_local, mean, invstd, running_mean, running_var, num_batches_tracked, momentum, eps, C);
    HIP_LAUNCH_CHECK();
    return 0;
}

// y = act(norm(x)*gamma+beta) [+ res] with given statistics (train or eval).
static int norm_apply_impl(const float* x, float* y, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, const float* res, int G, int P, int C, int act, float slope,
                           const float* slope_ptr, hipStream_t st, const PShuf ps = PShuf{}) {
    if ((size_t)G * P * C == 0) return 0;
    int VW, CTX, chunk;
    dim3 grid;
    apply_plan(G, P, C, VW, CTX, chunk, grid);
    if (ps.on && (VW != 4 || G != 1 || res)) return (int)hipErrorInvalidValue;
    if (VW == 4)
        MIGAN_LAUNCH((norm_apply_kernel<4>), grid, dim3(256), 0, st, x, y, mean, invstd, gamma, beta, res, P, C,
                           CTX, chunk, act, slope, slope_ptr, ps);
    else
        MIGAN_LAUNCH((norm_apply_kernel<1>), grid, dim3(256), 0, st, x, y, mean, invstd, gamma, beta, res, P, C,
                           CTX, chunk, act, slope, slope_ptr, ps);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_norm_apply(const float* x, float* y, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, const float* res, int G, int P, int C,
                               int act, float slope, void* stream) {
    return norm_apply_impl(x, y, mean, invstd, gamma, beta, res, G, P, C, act, slope, nullptr, (hipStream_t)stream);
}
// y = PReLU(norm(x)*gamma+beta) [+ res]: nn.BatchNorm2d -> nn.PReLU() of srgan/models.py:23-24,55-57 in the apply launch;
// prelu_weight is the layer's single learnable slope on the device (num_parameters = 1)
// shuffle_H, shuffle_W > 0: additionally nn.PixelShuffle(2) (srgan/models.py:56) - x is [N][H][W][C] with P = N*H*W, G = 1,
// C % 4 == 0, no residual, and y is written as [N][2H][2W][C/4] (see PShuf); prelu_weight may then be NULL (shuffle only)
MIGAN_API int migan_norm_apply_prelu(const float* x, float* y, const float* mean, const float* invstd, const float* gamma,
                                     const float* beta, const float* res, const float* prelu_weight, int G, int P, int C,
                                     int shuffle_H, int shuffle_W, void* stream) {
    const bool shuf = shuffle_H > 0 && shuffle_W > 0;
    if ((!prelu_weight && !shuf) || (shuf && (P % (shuffle_H * shuffle_W) != 0))) return (int)hipErrorInvalidValue;
    return norm_apply_impl(x, y, mean, invstd, gamma, beta, res, G, P, C, prelu_weight ? ACT_LRELU : ACT_NONE, 0.f, prelu_weight,
                           (hipStream_t)stream, shuf ? pshuf_make(shuffle_H, shuffle_W, C) : PShuf{});
}

// Number of [C]-slabs of per-block column sums the streaming backward kernels (migan_norm_bwd / migan_norm_bwd_apply /
// migan_act_bwd_colsum) write for a [G][P][C] view.
MIGAN_API int migan_norm_colsum_slabs(int G, int P, int C) {
    if ((size_t)G * P * C == 0) return 0;
    int VW, CTX, chunk;
    dim3 grid;
    apply_plan(G, P, C, VW, CTX, chunk, grid);
    return (int)(grid.y * grid.z);
}

// Backward, first half: sums[G][C][2] = (sum dyz, sum dyz*xhat) over this rank's pixels (dyz = dy * act'(z));
// dgamma/dbeta [C] written (or accumulated) when G == 1.
static int norm_bwd_sums_impl(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, float* sums, float* dgamma, float* dbeta, int G, int P, int C, int act,
                              float slope, float* ws, size_t ws_bytes, int accumulate, const float* slope_ptr,
                              float* dslope_gc, hipStream_t st, const PShuf ps = PShuf{}) {
    int VW, CTX, chunk, nchunks, gx;
    norm_plan(G, P, C, VW, CTX, chunk, nchunks, gx);
    if (ws_bytes < (size_t)G * nchunks * C * 3 * sizeof(float)) return (int)hipErrorInvalidValue;
    if (ps.on && (VW != 4 || G != 1)) return (int)hipErrorInvalidValue;
    dim3 grid(gx, nchunks, G);
    if (VW == 4)
        MIGAN_LAUNCH((norm_partial_kernel<4, true>), grid, dim3(256), 0, st, x, dy, mean, invstd, gamma,
                           beta, ws, P, C, CTX, chunk, nchunks, act, slope, slope_ptr, ps);
    else
        MIGAN_LAUNCH((norm_partial_kernel<1, true>), grid, dim3(256), 0, st, x, dy, mean, invstd, gamma,
                           beta, ws, P, C, CTX, chunk, nchunks, act, slope, slope_ptr, ps);
    HIP_LAUNCH_CHECK();
    const int chain = (G > 1 && (dgamma != nullptr || dbeta != nullptr)) ? 1 : 0;  // BatchNorm over G sub-batches
    MIGAN_LAUNCH(norm_finalize_bwd_kernel, dim3(cdiv((long)(chain ? 1 : G) * C * 64, 256)), dim3(256), 0, st, ws, sums,
                       dgamma, dbeta, G, C, nchunks, accumulate, dslope_gc, chain);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_norm_bwd_sums(const float* x, const float* dy, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, float* sums, float* dgamma, float* dbeta, int G,
                                  int P, int C, int act, float slope, float* ws, size_t ws_bytes, int accumulate,
                                  void* stream) {
    return norm_bwd_sums_impl(x, dy, mean, invstd, gamma, beta, sums, dgamma, dbeta, G, P, C, act, slope, ws, ws_bytes,
                              accumulate, nullptr, nullptr, (hipStream_t)stream);
}

// Backward, second half: dx = gamma*invstd*(dyz - sums0/P_total - xhat*sums1/P_total).  P_total = the number of pixels
// the sums cover: P, or world*P after the sums were all-reduced for cross-replica BatchNorm.  csum (optional):
// migan_norm_colsum_slabs() x [C] per-block column sums of dx, from which the preceding conv's bias gradient is
// reduced inside its wgrad launch (migan_conv2d_wgrad db_slabs).
static int norm_bwd_apply_impl(const float* x, const float* dy, float* dx, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, const float* sums, int G, int P, int C, int act,
                               float slope, long long P_total, float* csum, const float* slope_ptr, hipStream_t st,
                               const PShuf ps = PShuf{}) {
    if ((size_t)G * P * C == 0) return 0;
    int VW, CTX, chunk;
    dim3 grid;
    apply_plan(G, P, C, VW, CTX, chunk, grid);
    if (ps.on && (VW != 4 || G != 1)) return (int)hipErrorInvalidValue;
    const float invP = (float)(1.0 / (double)(P_total > 0 ? P_total : P));
    if (VW == 4)
        MIGAN_LAUNCH((norm_bwd_apply_kernel<4>), grid, dim3(256), 0, st, x, dy, dx, mean, invstd, gamma, beta,
                           sums, P, C, CTX, chunk, act, slope, invP, csum, slope_ptr, ps);
    else
        MIGAN_LAUNCH((norm_bwd_apply_kernel<1>), grid, dim3(256), 0, st, x, dy, dx, mean, invstd, gamma, beta,
                           sums, P, C, CTX, chunk, act, slope, invP, csum, slope_ptr, ps);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_norm_bwd_apply(const float* x, const float* dy, float* dx, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, const float* sums, int G, int P, int C,
                                   int act, float slope, long long P_total, float* csum, void* stream) {
    return norm_bwd_apply_impl(x, dy, dx, mean, invstd, gamma, beta, sums, G, P, C, act, slope, P_total, csum, nullptr,
                               (hipStream_t)stream);
}

// Backward of y = act(norm(x)*gamma+beta) through the batch statistics (both halves; ws as migan_norm_workspace()).
MIGAN_API int migan_norm_bwd(const float* x, const float* dy, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta,
                             int G, int P, int C, int act, float slope, float* ws, size_t ws_bytes,
                             int accumulate, float* csum, void* stream) {
    if (ws_bytes < migan_norm_workspace(G, P, C)) return (int)hipErrorInvalidValue;
    if (!dgamma && !dbeta && norm_small_ok(G, P, C) && (act == ACT_NONE || act == ACT_LRELU || act == ACT_RELU)) {
        // small instance-style tensor: both halves in one launch (see norm_small_fwd_kernel)
        const int rows_per_g = csum ? migan_norm_colsum_slabs(G, P, C) / G : 0;
        MIGAN_LAUNCH(norm_small_bwd_kernel, dim3(C / NS_CH, 1, G), dim3(256), 0, (hipStream_t)stream, x, dy, dx, mean, invstd,
                           gamma, beta, P, C, act, slope, csum, rows_per_g, (const float*)nullptr);
        HIP_LAUNCH_CHECK();
        return 0;
    }
    int VW, CTX, chunk, nchunks, gx;
    norm_plan(G, P, C, VW, CTX, chunk, nchunks, gx);
    float* sums = ws + (size_t)G * nchunks * C * 3;
    int rc = migan_norm_bwd_sums(x, dy, mean, invstd, gamma, beta, sums, dgamma, dbeta, G, P, C, act, slope, ws, ws_bytes,
                                 accumulate, stream);
    if (rc) return rc;
    return migan_norm_bwd_apply(x, dy, dx, mean, invstd, gamma, beta, sums, G, P, C, act, slope, P, csum, stream);
}

MIGAN_API int migan_norm_bwd_small(const float* x, const float* dy, const float* mask, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, float* dx, int G, int P, int C, int act, float slope,
                                   float* csum, void* stream) {
    if (!norm_small_ok(G, P, C) || !(act == ACT_NONE || act == ACT_LRELU || act == ACT_RELU)) return (int)hipErrorInvalidValue;
    const int rows_per_g = csum ? migan_norm_colsum_slabs(G, P, C) / G : 0;
    MIGAN_LAUNCH(norm_small_bwd_kernel, dim3(C / NS_CH, 1, G), dim3(256), 0, (hipStream_t)stream, x, dy, dx, mean, invstd,
                       gamma, beta, P, C, act, slope, csum, rows_per_g, mask);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Backward of y = PReLU(norm(x)*gamma+beta): as migan_norm_bwd with the slope read from the device, plus the slope's own
// gradient dprelu[0] (+)= sum dy * min(z, 0) - a third sum of the statistics pass, so the PReLU layer costs no pass of its
// own over the tensor (srgan/models.py:23-24: 16 residual blocks + 2 up-sampling stages per generator step).
// ws: migan_norm_workspace_prelu() bytes.
MIGAN_API size_t migan_norm_workspace_prelu(int G, int P, int C) {
    return migan_norm_workspace(G, P, C) + (size_t)G * C * sizeof(float);
}
MIGAN_API int migan_norm_bwd_prelu(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, const float* prelu_weight, float* dx, float* dgamma, float* dbeta,
                                   float* dprelu, int G, int P, int C, float* ws, size_t ws_bytes, int accumulate,
                                   int dprelu_accumulate, float* csum, int shuffle_H, int shuffle_W, void* stream) {
    const bool shuf = shuffle_H > 0 && shuffle_W > 0;
    if ((!prelu_weight && !shuf) || ws_bytes < migan_norm_workspace_prelu(G, P, C)) return (int)hipErrorInvalidValue;
    const PShuf ps = shuf ? pshuf_make(shuffle_H, shuffle_W, C) : PShuf{};
    const int act_ = prelu_weight ? ACT_LRELU : ACT_NONE;
    hipStream_t st = (hipStream_t)stream;
    int VW, CTX, chunk, nchunks, gx;
    norm_plan(G, P, C, VW, CTX, chunk, nchunks, gx);
    float* sums = ws + (size_t)G * nchunks * C * 3;
    float* dsl = ws + migan_norm_workspace(G, P, C) / sizeof(float);
    int rc = norm_bwd_sums_impl(x, dy, mean, invstd, gamma, beta, sums, dgamma, dbeta, G, P, C, act_, 0.f, ws,
                                migan_norm_workspace(G, P, C), accumulate, prelu_weight,
                                (dprelu && prelu_weight) ? dsl : nullptr, st, ps);
    if (rc) return rc;
    if (dprelu && prelu_weight) {
        MIGAN_LAUNCH(sum_small_kernel, dim3(1), dim3(256), 0, st, dsl, G * C, dprelu, dprelu_accumulate);
        HIP_LAUNCH_CHECK();
    }
    return norm_bwd_apply_impl(x, dy, dx, mean, invstd, gamma, beta, sums, G, P, C, act_, 0.f, P, csum, prelu_weight, st, ps);
}

// The two halves of migan_norm_bwd_prelu, for cross-replica BatchNorm (data parallel; srgan/models.py:23-24,55-57 at
// 2 images per rank): sums [G][C][2] over this rank's pixels are all-reduced (SUM) between them and P_total = world * P.  dprelu
// (+)= this rank's part of the slope gradient (the ranks' parts are summed with the rest of the gradient bucket).
// ws: migan_norm_workspace_prelu() bytes (the first half only).
MIGAN_API int migan_norm_bwd_sums_prelu(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                                        const float* beta, const float* prelu_weight, float* sums, float* dgamma, float* dbeta,
                                        float* dprelu, int G, int P, int C, float* ws, size_t ws_bytes, int accumulate,
                                        int dprelu_accumulate, int shuffle_H, int shuffle_W, void* stream) {
    const bool shuf = shuffle_H > 0 && shuffle_W > 0;
    if ((!prelu_weight && !shuf) || !sums || ws_bytes < migan_norm_workspace_prelu(G, P, C)) return (int)hipErrorInvalidValue;
    const PShuf ps = shuf ? pshuf_make(shuffle_H, shuffle_W, C) : PShuf{};
    hipStream_t st = (hipStream_t)stream;
    float* dsl = ws + migan_norm_workspace(G, P, C) / sizeof(float);
    int rc = norm_bwd_sums_impl(x, dy, mean, invstd, gamma, beta, sums, dgamma, dbeta, G, P, C, prelu_weight ? ACT_LRELU : ACT_NONE, 0.f,
                                ws, migan_norm_workspace(G, P, C), accumulate, prelu_weight, (dprelu && prelu_weight) ? dsl : nullptr,
                                st, ps);
    if (rc) return rc;
    if (dprelu && prelu_weight) {
        MIGAN_LAUNCH(sum_small_kernel, dim3(1), dim3(256), 0, st, dsl, G * C, dprelu, dprelu_accumulate);
        HIP_LAUNCH_CHECK();
    }
    return 0;
}
MIGAN_API int migan_norm_bwd_apply_prelu(const float* x, const float* dy, float* dx, const float* mean, const float* invstd,
                                         const float* gamma, const float* beta, const float* prelu_weight, const float* sums, int G,
                                         int P, int C, long long P_total, float* csum, int shuffle_H, int shuffle_W, void* stream) {
    const bool shuf = shuffle_H > 0 && shuffle_W > 0;
    if (!prelu_weight && !shuf) return (int)hipErrorInvalidValue;
    const PShuf ps = shuf ? pshuf_make(shuffle_H, shuffle_W, C) : PShuf{};
    return norm_bwd_apply_impl(x, dy, dx, mean, invstd, gamma, beta, sums, G, P, C, prelu_weight ? ACT_LRELU : ACT_NONE, 0.f, P_total,
                               csum, prelu_weight, (hipStream_t)stream, ps);
}

// Backward of `act [-> Dropout2d]` behind a conv, viewed [G = N][P = H*W][C]: dx = dy * mask[g][c] * act'(y) (mask may
// be NULL, act may be 0) plus migan_norm_colsum_slabs(G, P, C) x [C] per-block column sums of dx (see above).
MIGAN_API int migan_act_bwd_colsum(const float* dy, const float* y, const float* mask_gc, float* dx, float* csum, int G,
                                   int P, int C, int act, float slope, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if ((size_t)G * P * C == 0) return 0;
    if (!csum) return (int)hipErrorInvalidValue;
    int VW, CTX, chunk;
    dim3 grid;
    apply_plan(G, P, C, VW, CTX, chunk, grid);
    if (VW == 4)
        MIGAN_LAUNCH((act_bwd_colsum_kernel<4>), grid, dim3(256), 0, st, dy, y, mask_gc, dx, csum, P, C, CTX, chunk,
                           act, slope);
    else
        MIGAN_LAUNCH((act_bwd_colsum_kernel<1>), grid, dim3(256), 0, st, dy, y, mask_gc, dx, csum, P, C, CTX, chunk,
                           act, slope);
    HIP_LAUNCH_CHECK();
    return 0;
}

// eval-mode BatchNorm: invstd[c] = 1/sqrt(running_var[c] + eps)  (cyclegan/pix2pix sample paths call .eval() models)
__global__ void rsqrt_eps_kernel(const float* __restrict__ var, float* __restrict__ invstd, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) invstd[c] = (float)(1.0 / sqrt((double)var[c] + (double)eps));
}
MIGAN_API int migan_rsqrt_eps(const float* var, float* invstd, int C, float eps, void* stream) {
    if (C <= 0) return 0;
    MIGAN_LAUNCH(rsqrt_eps_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, var, invstd, C, eps);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Second-order backward of the normalisation (conv-critic gradient penalties: dragan.py:144-167 back-propagates through
// autograd.grad of a BatchNorm discriminator, dualgan.py:116-135 through InstanceNorm).  First-order backward, per
// (g, c) over P pixels, xh = (x - mean) * invstd, a = gamma * invstd:
//     dx = a * (d - m1 - xh * m2),   m1 = mean(d),  m2 = mean(d * xh)          (d = gradient w.r.t. the norm output)
// Given u = gradient w.r.t. dx, with Su = sum u, Sux = sum u*xh, Sud = sum u*d, Q = Sud - m1*Su - m2*Sux:
//     g_d = a * (u - Su/P - xh * Sux/P)                                       (the operator is self-adjoint)
//     g_x = -gamma * invstd^2 * ( xh * Q/P + Sux/P * (d - m1 - xh*m2) + m2 * (u - Su/P - xh*Sux/P) )
//     g_gamma = invstd * Q      (G == 1)
// Pass 1 (norm_bwd2_partial_kernel) = 5 sums per (g, chunk, c): Su, Sux, Sud, sum d, sum d*xh; pass 2 finalises them in
// double (one wave per (g,c)); pass 3 streams x, d, u once and writes g_d and g_x.
// ---------------------------------------------------------------------------------------------
template <int VW>
__global__ __launch_bounds__(256) void norm_bwd2_partial_kernel(const float* __restrict__ x, const float* __restrict__ d,
                                                                const float* __restrict__ u,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float* __restrict__ part,
                                                                int P, int C, int CTX, int chunk, int nchunks) {
    __shared__ float red[5][256 * VW];
    const int tid = threadIdx.x;
    const int tx = tid % CTX, ty = tid / CTX, TY = 256 / CTX;
    const int g = blockIdx.z, ck = blockIdx.y;
    const int c = (blockIdx.x * CTX + tx) * VW;
    const bool cok = c < C;
    float s[5][VW], mu[VW], is[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) {
#pragma unroll
        for (int q = 0; q < 5; ++q) s[q][v] = 0.f;
        mu[v] = cok ? mean[(size_t)g * C + c + v] : 0.f;
        is[v] = cok ? invstd[(size_t)g * C + c + v] : 0.f;
    }
    int p0 = ck * chunk, p1 = p0 + chunk;
    if (p1 > P) p1 = P;
    const size_t base = (size_t)g * P * C + c;
    if (cok) {
        for (int p = p0 + ty; p < p1; p += TY) {
            const size_t e = base + (size_t)p * C;
            float xv[VW], dv[VW], uv[VW];
            if (VW == 4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(x + e), b = *reinterpret_cast<const f32x4*>(d + e),
                            cc = *reinterpret_cast<const f32x4*>(u + e);
#pragma unroll
                for (int k = 0; k < 4; ++k) { xv[k] = a[k]; dv[k] = b[k]; uv[k] = cc[k]; }
            } else {
                xv[0] = x[e]; dv[0] = d[e]; uv[0] = u[e];
            }
#pragma unroll
            for (int v = 0; v < VW; ++v) {
                const float xh = (xv[v] - mu[v]) * is[v];
                s[0][v] += uv[v];
                s[1][v] += uv[v] * xh;
                s[2][v] += uv[v] * dv[v];
                s[3][v] += dv[v];
                s[4][v] += dv[v] * xh;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int v = 0; v < VW; ++v) red[q][tid * VW + v] = s[q][v];
    __syncthreads();
    if (ty == 0 && cok) {
#pragma unroll
        for (int v = 0; v < VW; ++v) {
            float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            for (int y = 0; y < TY; ++y)
#pragma unroll
                for (int q = 0; q < 5; ++q) a[q] += red[q][(y * CTX + tx) * VW + v];
            const size_t o = (((size_t)g * nchunks + ck) * C + c + v) * 5;
#pragma unroll
            for (int q = 0; q < 5; ++q) part[o + q] = a[q];
        }
    }
}
__global__ __launch_bounds__(256) void norm_bwd2_finalize_kernel(const float* __restrict__ part, float* __restrict__ sums,
                                                                 const float* __restrict__ invstd, float* dgamma, int G,
                                                                 int C, int nchunks, int accum) {
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= G * C) return;
    const int g = i / C, c = i - g * C;
    double a[5] = {0, 0, 0, 0, 0};
    for (int k = lane; k < nchunks; k += NF_B * 64) {
        float v[NF_B][5];
#pragma unroll
        for (int u = 0; u < NF_B; ++u) {
            const int kk = k + u * 64;
            const size_t o = (((size_t)g * nchunks + (kk < nchunks ? kk : nchunks - 1)) * C + c) * 5;
#pragma unroll
            for (int q = 0; q < 5; ++q) v[u][q] = part[o + q];
        }
#pragma unroll
        for (int u = 0; u < NF_B; ++u)
            if (k + u * 64 < nchunks) {
#pragma unroll
                for (int q = 0; q < 5; ++q) a[q] += (double)v[u][q];
            }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off);
    if (lane != 0) return;
#pragma unroll
    for (int q = 0; q < 5; ++q) sums[(size_t)i * 5 + q] = (float)a[q];
    (void)invstd; (void)dgamma; (void)accum;  // g_gamma is written by norm_bwd2_apply_kernel (it needs 1/P)
}
template <int VW>
__global__ __launch_bounds__(256) void norm_bwd2_apply_kernel(const float* __restrict__ x, const float* __restrict__ d,
                                                              const float* __restrict__ u, float* __restrict__ gd,
                                                              float* __restrict__ gx, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ sums, float* dgamma, int dg_accum,
                                                              int P, int C, int CTX, int chunk, float invP) {
    const int tid = threadIdx.x;
    const int tx = tid % CTX, ty = tid / CTX, TY = 256 / CTX;
    const int g = blockIdx.z;
    const int c = (blockIdx.x * CTX + tx) * VW;
    if (c >= C) return;
    float mu[VW], is[VW], ga[VW], su[VW], sux[VW], m1[VW], m2[VW], q[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) {
        const size_t gc = (size_t)g * C + c + v;
        mu[v] = mean[gc];
        is[v] = invstd[gc];
        ga[v] = gamma ? gamma[c + v] : 1.f;
        su[v] = sums[gc * 5] * invP;
        sux[v] = sums[gc * 5 + 1] * invP;
        m1[v] = sums[gc * 5 + 3] * invP;
        m2[v] = sums[gc * 5 + 4] * invP;
        const float Q = sums[gc * 5 + 2] - m1[v] * sums[gc * 5] - m2[v] * sums[gc * 5 + 1];
        q[v] = Q * invP;
        if (dgamma && blockIdx.y == 0 && ty == 0) {  // one thread per channel (G == 1 only)
            const float gg = is[v] * Q;
            dgamma[c + v] = dg_accum ? dgamma[c + v] + gg : gg;
        }
    }
    const int p0 = blockIdx.y * chunk;
    int p1 = p0 + chunk;
    if (p1 > P) p1 = P;
    const size_t base = (size_t)g * P * C + c;
#pragma unroll 2
    for (int p = p0 + ty; p < p1; p += TY) {
        const size_t e = base + (size_t)p * C;
        float xv[VW], dv[VW], uv[VW], od[VW], ox[VW];
        if (VW == 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + e), b = *reinterpret_cast<const f32x4*>(d + e),
                        cc = *reinterpret_cast<const f32x4*>(u + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) { xv[k] = a[k]; dv[k] = b[k]; uv[k] = cc[k]; }
        } else {
            xv[0] = x[e]; dv[0] = d[e]; uv[0] = u[e];
        }
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            const float xh = (xv[k] - mu[k]) * is[k];
            const float tu = uv[k] - su[k] - xh * sux[k];
            const float td = dv[k] - m1[k] - xh * m2[k];
            od[k] = ga[k] * is[k] * tu;
            ox[k] = -ga[k] * is[k] * is[k] * (xh * q[k] + sux[k] * td + m2[k] * tu);
        }
        if (VW == 4) {
            if (gd) *reinterpret_cast<f32x4*>(gd + e) = f32x4{od[0], od[1], od[2], od[3]};
            if (gx) *reinterpret_cast<f32x4*>(gx + e) = f32x4{ox[0], ox[1], ox[2], ox[3]};
        } else {
            if (gd) gd[e] = od[0];
            if (gx) gx[e] = ox[0];
        }
    }
}
// ws: migan_norm_workspace2(G,P,C) bytes.  gd / gx / dgamma may be NULL (not needed).
MIGAN_API size_t migan_norm_workspace2(int G, int P, int C) {
    int VW, CTX, chunk, nchunks, gx;
    norm_plan(G, P, C, VW, CTX, chunk, nchunks, gx);
    return ((size_t)G * nchunks * C * 5 + (size_t)G * C * 5) * sizeof(float);
}
MIGAN_API int migan_norm_bwd2(const float* x, const float* d, const float* u, const float* mean, const float* invstd,
                              const float* gamma, float* gd, float* gx, float* dgamma, int dgamma_accumulate, int G, int P,
                              int C, float* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if ((size_t)G * P * C == 0) return 0;
    if (ws_bytes < migan_norm_workspace2(G, P, C)) return (int)hipErrorInvalidValue;
    int VW, CTX, chunk, nchunks, gxb;
    norm_plan(G, P, C, VW, CTX, chunk, nchunks, gxb);
    float* sums = ws + (size_t)G * nchunks * C * 5;
    dim3 grid(gxb, nchunks, G);
    if (VW == 4)
        MIGAN_LAUNCH((norm_bwd2_partial_kernel<4>), grid, dim3(256), 0, st, x, d, u, mean, invstd, ws, P, C, CTX, chunk, nchunks);
    else
        MIGAN_LAUNCH((norm_bwd2_partial_kernel<1>), grid, dim3(256), 0, st, x, d, u, mean, invstd, ws, P, C, CTX, chunk, nchunks);
    HIP_LAUNCH_CHECK();
    MIGAN_LAUNCH(norm_bwd2_finalize_kernel, dim3(cdiv((long)G * C * 64, 256)), dim3(256), 0, st, ws, sums, invstd,
                       nullptr, G, C, nchunks, 0);
    HIP_LAUNCH_CHECK();
    int VW2, CTX2, chunk2;
    dim3 grid2;
    apply_plan(G, P, C, VW2, CTX2, chunk2, grid2);
    float* dgm = (G == 1) ? dgamma : nullptr;
    const float invP = (float)(1.0 / (double)P);
    if (VW == 4)
        MIGAN_LAUNCH((norm_bwd2_apply_kernel<4>), grid2, dim3(256), 0, st, x, d, u, gd, gx, mean, invstd, gamma, sums,
                           dgm, dgamma_accumulate, P, C, CTX2, chunk2, invP);
    else
        MIGAN_LAUNCH((norm_bwd2_apply_kernel<1>), grid2, dim3(256), 0, st, x, d, u, gd, gx, mean, invstd, gamma, sums,
                           dgm, dgamma_accumulate, P, C, CTX2, chunk2, invP);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Statistics from the per-tile (mean, M2, count) triples a conv kernel left in its epilogue (migan_conv2d_fwd_stats):
// Chan's parallel combination in double, ONE WAVE per (g, c): pass 1 total count and mean, pass 2
// M2 = sum_t (M2_t + n_t * (mean_t - mean)^2).  Replaces the statistics pass over the conv output
// (norm_partial_kernel<.., false> + norm_finalize_fwd_kernel) by this finalize alone.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void norm_finalize_chan_kernel(const float* __restrict__ part, float* __restrict__ mean,
                                                                 float* __restrict__ invstd, float* running_mean,
                                                                 float* running_var, long long* nbt, int G, int C,
                                                                 int nchunks, float eps, float momentum) {
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= G * C) return;
    const int g = i / C, c = i - g * C;
    double n = 0.0, sm = 0.0;
    for (int k = lane; k < nchunks; k += NF_B * 64) {
        float v[NF_B][2];
#pragma unroll
        for (int u = 0; u < NF_B; ++u) {
            const int kk = k + u * 64;
            const size_t o = (((size_t)g * nchunks + (kk < nchunks ? kk : nchunks - 1)) * C + c) * 3;
            v[u][0] = part[o];
            v[u][1] = part[o + 2];
        }
#pragma unroll
        for (int u = 0; u < NF_B; ++u)
            if (k + u * 64 < nchunks) {
                const double nt = (double)v[u][1];
                n += nt;
                sm += nt * (double)v[u][0];
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_xor(n, off);
        sm += __shfl_xor(sm, off);
    }
    const double m = n > 0.0 ? sm / n : 0.0;
    double M2 = 0.0;
    for (int k = lane; k < nchunks; k += NF_B * 64) {
        float v[NF_B][3];
#pragma unroll
        for (int u = 0; u < NF_B; ++u) {
            const int kk = k + u * 64;
            const size_t o = (((size_t)g * nchunks + (kk < nchunks ? kk : nchunks - 1)) * C + c) * 3;
            v[u][0] = part[o];
            v[u][1] = part[o + 1];
            v[u][2] = part[o + 2];
        }
#pragma unroll
        for (int u = 0; u < NF_B; ++u)
            if (k + u * 64 < nchunks) {
                const double nt = (double)v[u][2], d = (double)v[u][0] - m;
                M2 += (double)v[u][1] + nt * d * d;
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) M2 += __shfl_xor(M2, off);
    if (lane != 0) return;
    if (i == 0 && nbt) nbt[0] += 1;
    const double var = n > 0.0 ? M2 / n : 0.0;
    mean[i] = (float)m;
    invstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean && G == 1) {
        const double unb = n > 1.0 ? M2 / (n - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}
MIGAN_API int migan_norm_stats_from_conv(const float* part, int nchunks, float* mean, float* invstd, float* running_mean,
                                         float* running_var, long long* num_batches_tracked, float momentum, float eps,
                                         int G, int C, void* stream) {
    if (G < 1 || C < 1 || nchunks < 1) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(norm_finalize_chan_kernel, dim3(cdiv((long)G * C * 64, 256)), dim3(256), 0, (hipStream_t)stream, part,
                       mean, invstd, running_mean, running_var, num_batches_tracked, G, C, nchunks, eps, momentum);
    HIP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Backward of  BatchNorm2d(C) -> LeakyReLU / ReLU -> Conv2d(C, 1, 3, 1, 1)  (the generator's last block, dcgan.py:60-62) with the conv's
// input gradient never stored.  g = conv_transpose(dz, w) at a pixel is nine products of the one-channel dz around it with the pixel's
// own weight column - cheaper to recompute from the 2 MB dz than to write and re-read as a C-channel tensor (134 MB at the headline batch).
//   pass A (bn_conv1_bwd_sums_kernel): ONE walk over x yields the conv's weight-gradient partials (x read through the normalisation and the
//     activation, as the forward read it) AND the two batch sums of the BatchNorm backward; the nine dz taps of a pixel are shared by both.
//   pass B (bn_conv1_bwd_apply_kernel): dx = gamma * invstd * (g * act'(z) - s0/P - xhat * s1/P), g recomputed; column-sum slabs of dx for
//     the bias gradient of the conv in front, as norm_bwd_apply_kernel writes them.
// Before (profiles/r06_dcgan_kernel_stats.txt): input-gradient launch 51 us (writes g) + weight gradient 43 + sums 30 + apply 67 us.
// Thread layout of both: tx = tid % CTX owns four channels, ty = tid / CTX walks the pixels of the block's chunk (thin_wgrad_kernel's walk).
// ---------------------------------------------------------------------------------------------
struct BnConv1Geom {
    int N, H, W, C, CTX, chunk, nchunks, act;
    float slope, invP;
};

// The dz rows a block's pixel chunk [q0, q1) touches, one halo row above and below, one zero halo column left and right, staged in LDS:
// row k of the window is global row (q0 / W) - 1 + k of the [N * H][W] image stack (zero outside the stack).  Rows of a NEIGHBOURING image
// are real data there - a pixel in the first / last row of its image masks the tap row above / below it (mt / mb in the walks).
// (First form of these kernels: nine clamped, predicated global loads per pixel and thread - the address arithmetic of the taps made both
// walks VALU-bound, 69 / 62 us for 134 MB, profiles/r06_ab.txt call 41.)
__device__ __forceinline__ void bn_conv1_stage(const BnConv1Geom& g, const float* __restrict__ dz, float* dzw, int q0, int q1, int tid) {
    const int W2 = g.W + 2, row0 = q0 / g.W - 1, nrows = (q1 - 1) / g.W - row0 + 2, NH = g.N * g.H;
    for (int i = tid; i < nrows * W2; i += 256) {
        const int rr = i / W2, col = i - rr * W2 - 1, R = row0 + rr;
        dzw[i] = (R >= 0 && R < NH && col >= 0 && col < g.W) ? dz[(long)R * g.W + col] : 0.f;
    }
}
// taps of the pixel at window row wr, column iw: dv[r * 3 + s] = dz[ih + 1 - r][iw + 1 - s], rows outside the pixel's image masked
__device__ __forceinline__ void bn_conv1_taps(const float* dzw, int W2, int wr, int iw, float mt, float mb, float (&dv)[9]) {
    const float* p = dzw + wr * W2 + iw + 1;
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
        dv[0 + s_] = p[W2 + 1 - s_] * mb;
        dv[3 + s_] = p[1 - s_];
        dv[6 + s_] = p[-W2 + 1 - s_] * mt;
    }
}

__global__ __launch_bounds__(256) void bn_conv1_bwd_sums_kernel(const BnConv1Geom g, const float* __restrict__ x, const float* __restrict__ dz,
                                                                const float* __restrict__ w, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ wpart,
                                                                float* __restrict__ npart) {
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    extern __shared__ __attribute__((aligned(16))) float dzw[];
    const int tid = threadIdx.x;
    const int tx = tid % g.CTX, ty = tid / g.CTX, TY = 256 / g.CTX;
    const int c = (blockIdx.y * g.CTX + tx) * 4;
    const bool cok = c < g.C;
    f32x4 acc[9], wv[9], s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 mu = s0, is = s0, sc = s0, sh = s0;
    float sdz = 0.f;   // sum of dz over the chunk (lane tx == 0 of channel block 0): the conv's bias gradient
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        wv[t] = cok ? *reinterpret_cast<const f32x4*>(w + (size_t)t * g.C + c) : f32x4{0.f, 0.f, 0.f, 0.f};   // w: [1][3][3][C]
    }
    if (cok) {
        mu = *reinterpret_cast<const f32x4*>(mean + c);
        is = *reinterpret_cast<const f32x4*>(invstd + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // the forward's arithmetic (norm_apply_kernel / thin_conv_kernel INMAP)
            sc[k] = is[k] * (gamma ? gamma[c + k] : 1.f);
            sh[k] = (beta ? beta[c + k] : 0.f) - mu[k] * sc[k];
        }
    }
    const float nslope = g.act == ACT_NONE ? 1.f : (g.act == ACT_LRELU ? g.slope : 0.f);   // act'(z) for z <= 0
    const int P = g.N * g.H * g.W, W2 = g.W + 2;
    int q0 = blockIdx.x * g.chunk, q1 = q0 + g.chunk;
    if (q1 > P) q1 = P;
    bn_conv1_stage(g, dz, dzw, q0, q1, tid);
    __syncthreads();
    if (cok && q0 + ty < q1) {
        int q = q0 + ty;
        const int row0 = q0 / g.W - 1;
        int R = q / g.W, iw = q - R * g.W;
        int ih = R % g.H, wr = R - row0;
        const bool lead = tx == 0 && blockIdx.y == 0;
        // the x loads run BN_C1_U pixels ahead of the arithmetic (clamped index; the loop body is unrolled over the BN_C1_U slots): with one
        // 16-byte load per thread and iteration in flight the walk was bound by the round trip, 51 us for 134 MB (profiles/r06_ab.txt call 50)
        constexpr int BN_C1_U = 4;
        f32x4 xq[BN_C1_U];
#pragma unroll
        for (int u = 0; u < BN_C1_U; ++u) {
            const int qq = q + u * TY;
            xq[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(qq < q1 ? qq : q1 - 1) * g.C + c);
        }
        while (q < q1) {
#pragma unroll
            for (int u = 0; u < BN_C1_U; ++u) {
                if (q >= q1) break;
                while (iw >= g.W) {
                    iw -= g.W;
                    ++wr;
                    if (++ih >= g.H) ih = 0;
                }
                const f32x4 xv = xq[u];
                {
                    const int qn = q + BN_C1_U * TY;
                    xq[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(qn < q1 ? qn : q1 - 1) * g.C + c);
                }
                float dv[9];
                bn_conv1_taps(dzw, W2, wr, iw, ih > 0 ? 1.f : 0.f, ih < g.H - 1 ? 1.f : 0.f, dv);
                if (lead) sdz += dv[4];
                f32x4 a, xh, da;   // activated input of the conv, normalised value, act'(z)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    xh[k] = (xv[k] - mu[k]) * is[k];
                    const float z = fmaf(xv[k], sc[k], sh[k]);
                    da[k] = z > 0.f ? 1.f : nslope;
                    a[k] = z * da[k];
                }
                f32x4 gq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    acc[t] += a * dv[t];
                    gq += wv[t] * dv[t];
                }
                const f32x4 d = gq * da;
                s0 += d;
                s1 += d * xh;
                q += TY;
                iw += TY;
            }
        }
    }
    // reduce over the TY pixel lanes through LDS, one slab at a time (fixed order)
    float* wout = wpart + (size_t)blockIdx.x * 9 * g.C;
    for (int a_ = 0; a_ < 12; ++a_) {
        f32x4 v = acc[0];
#pragma unroll
        for (int i = 1; i < 9; ++i)
            if (i == a_) v = acc[i];
        if (a_ == 9) v = s0;
        if (a_ == 10) v = s1;
        if (a_ == 11) v = f32x4{sdz, 0.f, 0.f, 0.f};
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + tid * 4) = v;
        __syncthreads();
        if (ty == 0 && cok) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int y = 0; y < TY; ++y) s += *reinterpret_cast<const f32x4*>(red + (y * g.CTX + tx) * 4);
            if (a_ < 9) {
                *reinterpret_cast<f32x4*>(wout + (size_t)a_ * g.C + c) = s;
            } else if (a_ < 11) {
#pragma unroll
                for (int k = 0; k < 4; ++k) npart[((size_t)blockIdx.x * g.C + c + k) * 3 + (a_ - 9)] = s[k];   // norm_finalize_bwd_kernel's records
            } else {
                // third field of the records: the chunk's sum of dz in channel 0's record, zero elsewhere - the finalize kernel's
                // per-channel third sum (the PReLU-slope slot) then holds the conv's bias gradient in element 0
#pragma unroll
                for (int k = 0; k < 4; ++k) npart[((size_t)blockIdx.x * g.C + c + k) * 3 + 2] = (c + k == 0) ? s[0] : 0.f;
            }
        }
    }
}

__global__ __launch_bounds__(256) void bn_conv1_bwd_apply_kernel(const BnConv1Geom g, const float* __restrict__ x, const float* __restrict__ dz,
                                                                 const float* __restrict__ w, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ sums,
                                                                 float* __restrict__ dx, float* __restrict__ csum) {
    __shared__ float red[256 * 4];
    extern __shared__ __attribute__((aligned(16))) float dzw[];
    const int tid = threadIdx.x;
    const int tx = tid % g.CTX, ty = tid / g.CTX, TY = 256 / g.CTX;
    const int c = (blockIdx.x * g.CTX + tx) * 4;
    const bool cok = c < g.C;
    f32x4 wv[9];
    float mu[4], is[4], sc[4], sh[4], gi[4], k0[4], k1[4], cs[4];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = cok ? *reinterpret_cast<const f32x4*>(w + (size_t)t * g.C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cs[k] = 0.f;
        const int cc = cok ? c + k : 0;
        mu[k] = mean[cc];
        is[k] = invstd[cc];
        const float ga = (gamma && cok) ? gamma[cc] : 1.f;
        sc[k] = is[k] * ga;
        sh[k] = ((beta && cok) ? beta[cc] : 0.f) - mu[k] * sc[k];
        gi[k] = ga * is[k];
        k0[k] = sums[(size_t)cc * 2] * g.invP;
        k1[k] = sums[(size_t)cc * 2 + 1] * g.invP;
    }
    const float nslope = g.act == ACT_NONE ? 1.f : (g.act == ACT_LRELU ? g.slope : 0.f);
    const int P = g.N * g.H * g.W, W2 = g.W + 2;
    int q0 = blockIdx.y * g.chunk, q1 = q0 + g.chunk;
    if (q1 > P) q1 = P;
    if (q0 < q1) bn_conv1_stage(g, dz, dzw, q0, q1, tid);
    __syncthreads();
    if (cok && q0 + ty < q1) {
        int q = q0 + ty;
        const int row0 = q0 / g.W - 1;
        int R = q / g.W, iw = q - R * g.W;
        int ih = R % g.H, wr = R - row0;
        constexpr int BN_C1_U = 4;   // x loads four pixels ahead, as in the first walk
        f32x4 xq[BN_C1_U];
#pragma unroll
        for (int u = 0; u < BN_C1_U; ++u) {
            const int qq = q + u * TY;
            xq[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(qq < q1 ? qq : q1 - 1) * g.C + c);
        }
        while (q < q1) {
#pragma unroll
            for (int u = 0; u < BN_C1_U; ++u) {
                if (q >= q1) break;
                while (iw >= g.W) {
                    iw -= g.W;
                    ++wr;
                    if (++ih >= g.H) ih = 0;
                }
                const f32x4 xv = xq[u];
                {
                    const int qn = q + BN_C1_U * TY;
                    xq[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(qn < q1 ? qn : q1 - 1) * g.C + c);
                }
                float dv[9];
                bn_conv1_taps(dzw, W2, wr, iw, ih > 0 ? 1.f : 0.f, ih < g.H - 1 ? 1.f : 0.f, dv);
                f32x4 gq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 9; ++t) gq += wv[t] * dv[t];
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (xv[k] - mu[k]) * is[k];
                    const float z = fmaf(xv[k], sc[k], sh[k]);
                    const float da = z > 0.f ? 1.f : nslope;
                    o[k] = gi[k] * (gq[k] * da - k0[k] - xh * k1[k]);
                    cs[k] += o[k];
                }
                *reinterpret_cast<f32x4*>(dx + (size_t)q * g.C + c) = o;
                q += TY;
                iw += TY;
            }
        }
    }
    if (csum) colsum_slab_store<4>(cs, red, csum, tid, tx, g.CTX, TY, cok, c, g.C);
}

int wgrad_reduce_slabs(const float* ws, float* dw, int nslabs, int Co, int T, int Ci, int accum, const float* db_slabs, float* db, int db_nslab,
                       int db_accum, hipStream_t st);   // conv_igemm.hip

static size_t bn_conv1_lds(int chunk, int W) { return (size_t)(chunk / W + 4) * (W + 2) * sizeof(float); }   // window rows of a chunk
static bool bn_conv1_ok(int N, int H, int W, int C) {
    if (!(N >= 1 && H >= 1 && W >= 1 && C % 4 == 0 && C >= 16 && C <= 256 && (long)N * H * W * C < (1L << 31) && (long)N * H * W >= 2))
        return false;
    int VW, CTX, chunk, nchunks, gx;
    norm_plan(1, N * H * W, C, VW, CTX, chunk, nchunks, gx);
    dim3 gridB;
    int chunkB;
    apply_plan(1, N * H * W, C, VW, CTX, chunkB, gridB);
    return bn_conv1_lds(chunk, W) <= 48 * 1024 && bn_conv1_lds(chunkB, W) <= 48 * 1024;
}
MIGAN_API int migan_bn_conv1_bwd_ok(int N, int H, int W, int C) { return bn_conv1_ok(N, H, W, C) ? 1 : 0; }
// ws: weight-gradient partials [nchunks][9][C] + sums records [nchunks][C][3] + sums [C][2] + third sums [C]
MIGAN_API size_t migan_bn_conv1_bwd_workspace(int N, int H, int W, int C) {
    if (!bn_conv1_ok(N, H, W, C)) return 0;
    int VW, CTX, chunk, nchunks, gx;
    norm_plan(1, N * H * W, C, VW, CTX, chunk, nchunks, gx);
    return ((size_t)nchunks * C * 12 + (size_t)C * 3) * sizeof(float);
}
// x [N][H][W][C] (the BatchNorm input), dz [N][H][W] (gradient at the conv's pre-activation output), w_ohwi [1][3][3][C], mean / invstd [C]
// (migan_norm_stats), gamma / beta [C] or NULL; act: ACT_NONE / ACT_LRELU / ACT_RELU between the two.
// Writes dx [N][H][W][C], dw_oihw [1][C][3][3] (dw_accumulate: +=), db [1] = sum of dz (optional; db_accumulate: +=), dgamma / dbeta [C]
// (optional; affine_accumulate: +=), and - csum != NULL - migan_norm_colsum_slabs(1, N*H*W, C) x [C] column-sum slabs of dx.
MIGAN_API int migan_bn_conv1_bwd(const float* x, const float* dz, const float* w_ohwi, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int act, float slope, float* dx, float* dw_oihw, int dw_accumulate,
                                 float* db, int db_accumulate, float* dgamma, float* dbeta, int affine_accumulate, float* csum, float* ws,
                                 size_t ws_bytes, int N, int H, int W, int C, void* stream) {
    if (!bn_conv1_ok(N, H, W, C) || (act != ACT_NONE && act != ACT_LRELU && act != ACT_RELU)) return (int)hipErrorInvalidValue;
    if (ws_bytes < migan_bn_conv1_bwd_workspace(N, H, W, C)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int P = N * H * W;
    int VW, CTX, chunk, nchunks, gx;
    norm_plan(1, P, C, VW, CTX, chunk, nchunks, gx);
    BnConv1Geom g = {N, H, W, C, CTX, chunk, nchunks, act, slope, 1.f / (float)P};
    float* wpart = ws;
    float* npart = ws + (size_t)nchunks * C * 9;
    float* sums = npart + (size_t)nchunks * C * 3;
    float* third = sums + (size_t)C * 2;
    MIGAN_LAUNCH(bn_conv1_bwd_sums_kernel, dim3(nchunks, gx), dim3(256), bn_conv1_lds(chunk, W), st, g, x, dz, w_ohwi, mean, invstd, gamma,
                 beta, wpart, npart);
    HIP_LAUNCH_CHECK();
    MIGAN_LAUNCH(norm_finalize_bwd_kernel, dim3(cdiv((long)C * 64, 256)), dim3(256), 0, st, npart, sums, dgamma, dbeta, 1, C, nchunks,
                 affine_accumulate, db ? third : (float*)nullptr, 0);
    HIP_LAUNCH_CHECK();
    if (db) {   // third[0] = sum over the chunks of their dz sums
        MIGAN_LAUNCH(sum_small_kernel, dim3(1), dim3(256), 0, st, third, 1, db, db_accumulate);
        HIP_LAUNCH_CHECK();
    }
    if (int rc = wgrad_reduce_slabs(wpart, dw_oihw, nchunks, 1, 9, C, dw_accumulate, nullptr, nullptr, 0, 0, st)) return rc;
    int chunkB;
    dim3 gridB;
    apply_plan(1, P, C, VW, CTX, chunkB, gridB);
    g.chunk = chunkB;
    MIGAN_LAUNCH(bn_conv1_bwd_apply_kernel, gridB, dim3(256), bn_conv1_lds(chunkB, W), st, g, x, dz, w_ohwi, mean, invstd, gamma, beta, sums,
                 dx, csum);
    HIP_LAUNCH_CHECK();
    return 0;
}
