// K7: one WGAN-GP critic iteration of the MLP critic (wgan_gp.py:68-83, 119-138, 160-176) as SIX dependent launches:
//     real_v = D(real), fake_v = D(fake), gp = compute_gradient_penalty(D, real, fake), d_loss = -mean(real_v) + mean(fake_v) + 10 gp,
//     d_loss.backward()                                  ->  d_loss, gp and the gradient of every critic parameter.
// D = Linear(Din,H1) LeakyReLU Linear(H1,H2) LeakyReLU Linear(H2,1) at B <= 64 rows: 1.4 GFLOP, weights 2.6 MB (L2 / MALL
// resident) - the op-by-op path spends 0.41 ms in ~75 launches of 3-6 us on it.  The chain of seven GEMMs is inherently serial
// (every stage needs whole rows of the previous one), so the unit of cost is a dependent launch: 2.35 us for an empty one on the
// MI355X, 4.6-6 us for a stage whose waves issue ONE batch of loads (profiles/r04_abi_check_and_two_rank.txt).  Round 3's kernel
// ran the same stages as seven launches of 4.6-34.5 us (116.7 us back to back): its weight-gradient stage walked K = 3B rows as
// twelve dependent load rounds per wave (34.5 us), the x^ stages took two rounds.  Here every wave issues all the loads of its
// K slice at once (<= 128-256 deep: one round trip to the L2 / Infinity Cache), the weight gradients ride in the launches of the
// stages they can overlap (dW1 beside du1, dW2 / dw3 / the losses beside du2), and the gradients are written, not accumulated
// (the caller skips the zero fill of the bucket).  A grid-wide barrier instead of a launch boundary was measured and rejected
// (4.8-7.2 us per barrier at best, MI355X_MICROARCH.md; 152 us for the single persistent launch of round 3).
//
// With x^ = a x_r + (1-a) x_f,  a1 = X W1^T + b1, h1 = lrelu(a1), m1 = lrelu'(a1), a2 = h1 W2^T + b2, h2, m2, o = h2 w3^T + b3:
//   gp path     u2 = m2 (.) w3,  v1 = u2 W2,  u1 = m1 (.) v1,  g = u1 W1,  n_i = |g_i|,  gp = mean (n_i - 1)^2            (rows of x^)
//   its grads   c_i = 2 lambda (n_i - 1) / (B n_i) g_i,  du1 = c W1^T,  dv1 = m1 (.) du1,  du2 = dv1 W2^T  (lrelu'' = 0)
//   plain part  do = -1/B (real rows), +1/B (fake rows):  da2 = do w3 (.) m2,  dh1 = da2 W2,  da1 = dh1 (.) m1
//   dW1 = [u1; da1_r; da1_f]^T [c; x_r; x_f]      dW2 = [u2; da2_r; da2_f]^T [dv1; h1_r; h1_f]      (one K = 3B GEMM each)
//   dw3 = sum_{r,f} do h2 + sum_{x^} m2 (.) du2,  db1 = colsum [da1_r; da1_f],  db2 = colsum [da2_r; da2_f],  db3 = sum do = 0
// x^ is never materialised: a1 is affine in X and the interpolation weights sum to one, so a1(x^) = a a1(x_r) + (1-a) a1(x_f)
// (one third of the largest GEMM less; the rounding differs from interpolate-then-multiply by ~1 ulp of a1).
// Launches: 1 a1 (real, fake -> three row blocks of h1)  2 a2 (+ S2 = [u2; da2], row dots for o)  3 [v1; dh1] = S2 W2 (-> S1 = [u1; da1])
//           4 g = u1 W1 (+ row sums of squares)  5 dv1 = m1 (.) coef (g W1^T) | dW1, db1  6 dw3 (x^ part via du2 = dv1 W2^T) | dW2, db2 | gp, losses.
// Tiles are 16x16 (NT) / 16x32 (NN) / 16x64 (TN) on v_mfma_f32_16x16x4_f32, operands straight from L2 (the loops of skinny_mm.hip),
// K cut over the waves of a workgroup and combined through LDS in a fixed order (deterministic).
#include "common.h"

#define CF_WAVES 8
#define CF_THREADS (64 * CF_WAVES)

struct CriticFused {
    int B, RB, Din, H1, H2;  // rows, rows rounded up to 16, layer widths (all % 128 == 0)
    float slope, lambda;
    int accum;               // 1: gradients are ADDED into gW1..gb3, 0: written
    const float *real, *fake, *alpha;
    const float *W1, *b1, *W2, *b2, *w3, *b3;
    float *gW1, *gb1, *gW2, *gb2, *gw3, *gb3;
    float* out;       // d_loss, gp, mean D(real), mean D(fake)
    float* ws;        // see cf_layout
};

// workspace carve-up (floats); row blocks are [x^ | real | fake], RB rows each
struct CfLayout {
    size_t h1, h2, s2, s1, g, dv1, opart, gsq, total;
};
static __host__ __device__ inline CfLayout cf_layout(int RB, int Din, int H1, int H2) {
    CfLayout L;
    size_t o = 0;
    L.h1 = o; o += (size_t)3 * RB * H1;
    L.h2 = o; o += (size_t)3 * RB * H2;
    L.s2 = o; o += (size_t)3 * RB * H2;
    L.s1 = o; o += (size_t)3 * RB * H1;
    L.g = o; o += (size_t)RB * Din;
    L.dv1 = o; o += (size_t)RB * H1;
    L.opart = o; o += (size_t)3 * RB * (H2 / 16);
    L.gsq = o; o += (size_t)RB * (Din / 32);
    L.total = o;
    return L;
}

__device__ __forceinline__ f32x4 cf_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float cf_lrelu(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ float cf_mask(float h, float s) { return h > 0.f ? 1.f : s; }  // lrelu'(a) from lrelu(a): same sign

// One wave's K slice of a 16x16 tile, C += A[16][klen] W[16][klen]^T; ap / wp point at this lane's row (+ 4 * (lane >> 4)).
// NB float4 per operand are issued before the first MFMA (NB * 16 k values: one round trip for the whole batch).
template <int NB>
__device__ __forceinline__ void cf_nt_batch(const float* __restrict__ ap, const float* __restrict__ wp, f32x4& acc0, f32x4& acc1) {
    f32x4 a[NB], b[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        a[u] = *reinterpret_cast<const f32x4*>(ap + 16 * u);
        b[u] = *reinterpret_cast<const f32x4*>(wp + 16 * u);
    }
    __builtin_amdgcn_sched_barrier(0);   // every load of the batch is issued before its first MFMA (one round trip, not NB)
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (u & 1) acc1 = cf_mfma(a[u][s], b[u][s], acc1);
            else acc0 = cf_mfma(a[u][s], b[u][s], acc0);
        }
}
__device__ __forceinline__ f32x4 cf_nt_partial(const float* __restrict__ ap, const float* __restrict__ wp, int klen) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int k0 = 0;
    for (; k0 + 128 <= klen; k0 += 128) cf_nt_batch<8>(ap + k0, wp + k0, acc0, acc1);   // 128 k values per round of loads (64 registers)
    if (k0 + 64 <= klen) { cf_nt_batch<4>(ap + k0, wp + k0, acc0, acc1); k0 += 64; }
    if (k0 + 32 <= klen) { cf_nt_batch<2>(ap + k0, wp + k0, acc0, acc1); k0 += 32; }
    if (k0 + 16 <= klen) cf_nt_batch<1>(ap + k0, wp + k0, acc0, acc1);
    return acc0 + acc1;
}
// two row sets against the same weight rows (launch 1: real and fake)
template <int NB>
__device__ __forceinline__ void cf_nt_batch2(const float* __restrict__ ap0, const float* __restrict__ ap1, const float* __restrict__ wp,
                                             f32x4& acc0, f32x4& acc1) {
    f32x4 a[NB], c[NB], b[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        a[u] = *reinterpret_cast<const f32x4*>(ap0 + 16 * u);
        c[u] = *reinterpret_cast<const f32x4*>(ap1 + 16 * u);
        b[u] = *reinterpret_cast<const f32x4*>(wp + 16 * u);
    }
    __builtin_amdgcn_sched_barrier(0);   // every load of the batch is issued before its first MFMA (one round trip, not NB)
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc0 = cf_mfma(a[u][s], b[u][s], acc0);
            acc1 = cf_mfma(c[u][s], b[u][s], acc1);
        }
}
__device__ __forceinline__ void cf_nt_partial2(const float* __restrict__ ap0, const float* __restrict__ ap1,
                                               const float* __restrict__ wp, int klen, f32x4& acc0, f32x4& acc1) {
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    int k0 = 0;
    for (; k0 + 128 <= klen; k0 += 128) cf_nt_batch2<8>(ap0 + k0, ap1 + k0, wp + k0, acc0, acc1);
    if (k0 + 64 <= klen) { cf_nt_batch2<4>(ap0 + k0, ap1 + k0, wp + k0, acc0, acc1); k0 += 64; }
    if (k0 + 32 <= klen) { cf_nt_batch2<2>(ap0 + k0, ap1 + k0, wp + k0, acc0, acc1); k0 += 32; }
    if (k0 + 16 <= klen) cf_nt_batch2<1>(ap0 + k0, ap1 + k0, wp + k0, acc0, acc1);
}
// C[16][32] += A[16][rlen] W[rlen][Nc]: tile e of the wave holds columns col0 + 2 * (lane & 15) + e.  ap: lane's row + 4*(lane>>4);
// wp: W + (4 * (lane >> 4)) * Nc + col0 + 2 * (lane & 15)  (both at the start of this wave's slice)
typedef float cf_f32x2 __attribute__((ext_vector_type(2)));
template <int NB>
__device__ __forceinline__ void cf_nn_batch(const float* __restrict__ ap, const float* __restrict__ wp, int Nc, f32x4& acc0, f32x4& acc1) {
    f32x4 a[NB];
    cf_f32x2 b[NB][4];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        a[u] = *reinterpret_cast<const f32x4*>(ap + 16 * u);
#pragma unroll
        for (int s = 0; s < 4; ++s) b[u][s] = *reinterpret_cast<const cf_f32x2*>(wp + (size_t)(16 * u + s) * Nc);
    }
    __builtin_amdgcn_sched_barrier(0);   // every load of the batch is issued before its first MFMA (one round trip, not NB)
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc0 = cf_mfma(a[u][s], b[u][s][0], acc0);
            acc1 = cf_mfma(a[u][s], b[u][s][1], acc1);
        }
}
__device__ __forceinline__ void cf_nn_partial(const float* __restrict__ ap, const float* __restrict__ wp, int Nc, int rlen,
                                              f32x4& acc0, f32x4& acc1) {
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    int n0 = 0;
    for (; n0 + 64 <= rlen; n0 += 64) cf_nn_batch<4>(ap + n0, wp + (size_t)n0 * Nc, Nc, acc0, acc1);
    if (n0 + 32 <= rlen) { cf_nn_batch<2>(ap + n0, wp + (size_t)n0 * Nc, Nc, acc0, acc1); n0 += 32; }
    if (n0 + 16 <= rlen) cf_nn_batch<1>(ap + n0, wp + (size_t)n0 * Nc, Nc, acc0, acc1);
}

// sum over the 16 lanes that share lane >> 4 (the 16 columns of one output row)
__device__ __forceinline__ float cf_rowsum16(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ float cf_wavesum(float v) {
    v = cf_rowsum16(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
// the penalty's coefficient of row i from the per-tile row sums of squares of launch 4; torch's norm() backward: zero subgradient at 0
__device__ __forceinline__ float cf_coef(float n2, float lambda, float invB) {
    const float n = sqrtf(n2);
    return n > 0.f ? 2.f * lambda * (n - 1.f) * invB / n : 0.f;
}

#define CF_COMMON                                                        \
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;            \
    const int rr = lane & 15, kq = lane >> 4;                            \
    const int B = p.B, RB = p.RB, Din = p.Din, H1 = p.H1, H2 = p.H2;     \
    const int RG = RB / 16;                                              \
    const CfLayout L = cf_layout(RB, Din, H1, H2);                       \
    float* const h1b = p.ws + L.h1;                                      \
    float* const h2b = p.ws + L.h2;                                      \
    float* const s2b = p.ws + L.s2;                                      \
    float* const s1b = p.ws + L.s1;                                      \
    float* const gb = p.ws + L.g;                                        \
    float* const dv1b = p.ws + L.dv1;                                    \
    float* const opart = p.ws + L.opart;                                 \
    float* const gsq = p.ws + L.gsq;                                     \
    const float slope = p.slope, invB = 1.f / (float)B;                  \
    (void)lane; (void)ks; (void)rr; (void)kq; (void)RG; (void)h1b; (void)h2b; (void)s2b; (void)s1b; (void)gb; (void)dv1b;  \
    (void)opart; (void)gsq; (void)slope; (void)invB; (void)Din; (void)H1; (void)H2; (void)B

// ---- launch 1: a1 of the real and fake rows, K = Din; three row blocks of h1 out.  One 16x16 tile per workgroup, K over the 8 waves
__global__ __launch_bounds__(CF_THREADS) void critic_fused_p1_kernel(const CriticFused p) {
    __shared__ f32x4 part[2][(CF_WAVES - 1) * 64];
    CF_COMMON;
    const int t = blockIdx.x;
    const int g = t / (H1 / 16), c = t - g * (H1 / 16);
    const int r0 = g * 16, col0 = c * 16, klen = Din / CF_WAVES;
    const int arow = r0 + rr < B ? r0 + rr : B - 1;
    f32x4 ar, af;
    cf_nt_partial2(p.real + (size_t)arow * Din + ks * klen + kq * 4, p.fake + (size_t)arow * Din + ks * klen + kq * 4,
                   p.W1 + (size_t)(col0 + rr) * Din + ks * klen + kq * 4, klen, ar, af);
    if (ks > 0) {
        part[0][(ks - 1) * 64 + lane] = ar;
        part[1][(ks - 1) * 64 + lane] = af;
    }
    __syncthreads();
    if (ks != 0) return;
#pragma unroll
    for (int q = 1; q < CF_WAVES; ++q) {
        ar += part[0][(q - 1) * 64 + lane];
        af += part[1][(q - 1) * 64 + lane];
    }
    const int col = col0 + rr;
    const float bv = p.b1[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = r0 + kq * 4 + r;
        const bool ok = i < B;
        const float al = p.alpha[ok ? i : 0];
        const float vr = ar[r] + bv, vf = af[r] + bv;
        const float vx = al * vr + (1.f - al) * vf;
        h1b[(size_t)(0 * RB + i) * H1 + col] = ok ? cf_lrelu(vx, slope) : 0.f;
        h1b[(size_t)(1 * RB + i) * H1 + col] = ok ? cf_lrelu(vr, slope) : 0.f;
        h1b[(size_t)(2 * RB + i) * H1 + col] = ok ? cf_lrelu(vf, slope) : 0.f;
    }
}

// ---- launch 2: a2 for the 3 row blocks, K = H1; h2, S2 = [u2; da2_r; da2_f], per-tile row dots h2 . w3
__global__ __launch_bounds__(CF_THREADS) void critic_fused_p2_kernel(const CriticFused p) {
    __shared__ f32x4 part[(CF_WAVES - 1) * 64];
    CF_COMMON;
    const int t = blockIdx.x;
    const int g = t / (H2 / 16), c = t - g * (H2 / 16);
    const int blk = g / RG, r0 = g * 16 /* row in the stacked buffer */, col0 = c * 16, klen = H1 / CF_WAVES;
    f32x4 acc = cf_nt_partial(h1b + (size_t)(r0 + rr) * H1 + ks * klen + kq * 4, p.W2 + (size_t)(col0 + rr) * H1 + ks * klen + kq * 4, klen);
    if (ks > 0) part[(ks - 1) * 64 + lane] = acc;
    __syncthreads();
    if (ks != 0) return;
#pragma unroll
    for (int q = 1; q < CF_WAVES; ++q) acc += part[(q - 1) * 64 + lane];
    const int col = col0 + rr;
    const float bv = p.b2[col], w3 = p.w3[col];
    const float dout = blk == 0 ? 1.f : (blk == 1 ? -invB : invB);  // block 0: u2 = m2 (.) w3
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + kq * 4 + r, i = row - blk * RB;
        const bool ok = i < B;
        const float h = ok ? cf_lrelu(acc[r] + bv, slope) : 0.f;
        h2b[(size_t)row * H2 + col] = h;
        s2b[(size_t)row * H2 + col] = ok ? dout * w3 * cf_mask(h, slope) : 0.f;
        const float dot = cf_rowsum16(h * w3);
        if (rr == 0) opart[(size_t)row * (H2 / 16) + c] = dot;
    }
}

// ---- launch 3: T = S2 W2 (NN, K = H2), S1 = m1 (.) T  (u1 for the x^ rows, da1 for real / fake)
__global__ __launch_bounds__(CF_THREADS) void critic_fused_p3_kernel(const CriticFused p) {
    __shared__ f32x4 part[2][(CF_WAVES - 1) * 64];
    CF_COMMON;
    const int t = blockIdx.x;
    const int g = t / (H1 / 32), c = t - g * (H1 / 32);
    const int r0 = g * 16, col0 = c * 32, rlen = H2 / CF_WAVES;
    f32x4 a0, a1;
    cf_nn_partial(s2b + (size_t)(r0 + rr) * H2 + ks * rlen + kq * 4, p.W2 + (size_t)(ks * rlen + kq * 4) * H1 + col0 + 2 * rr, H1, rlen, a0, a1);
    if (ks > 0) {
        part[0][(ks - 1) * 64 + lane] = a0;
        part[1][(ks - 1) * 64 + lane] = a1;
    }
    __syncthreads();
    if (ks != 0) return;
#pragma unroll
    for (int q = 1; q < CF_WAVES; ++q) {
        a0 += part[0][(q - 1) * 64 + lane];
        a1 += part[1][(q - 1) * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t o = (size_t)(r0 + kq * 4 + r) * H1 + col0 + 2 * rr;
        const cf_f32x2 hv = *reinterpret_cast<const cf_f32x2*>(h1b + o);
        cf_f32x2 v = {cf_mask(hv[0], slope) * a0[r], cf_mask(hv[1], slope) * a1[r]};
        *reinterpret_cast<cf_f32x2*>(s1b + o) = v;
    }
}

// ---- launch 4: g = u1 W1 (NN, K = H1) for the x^ rows, per-tile row sums of squares
__global__ __launch_bounds__(CF_THREADS) void critic_fused_p4_kernel(const CriticFused p) {
    __shared__ f32x4 part[2][(CF_WAVES - 1) * 64];
    CF_COMMON;
    const int t = blockIdx.x;
    const int g = t / (Din / 32), c = t - g * (Din / 32);
    const int r0 = g * 16, col0 = c * 32, rlen = H1 / CF_WAVES;
    f32x4 a0, a1;
    cf_nn_partial(s1b + (size_t)(r0 + rr) * H1 + ks * rlen + kq * 4, p.W1 + (size_t)(ks * rlen + kq * 4) * Din + col0 + 2 * rr, Din, rlen, a0, a1);
    if (ks > 0) {
        part[0][(ks - 1) * 64 + lane] = a0;
        part[1][(ks - 1) * 64 + lane] = a1;
    }
    __syncthreads();
    if (ks != 0) return;
#pragma unroll
    for (int q = 1; q < CF_WAVES; ++q) {
        a0 += part[0][(q - 1) * 64 + lane];
        a1 += part[1][(q - 1) * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + kq * 4 + r;
        const size_t o = (size_t)row * Din + col0 + 2 * rr;
        cf_f32x2 v = {a0[r], a1[r]};
        *reinterpret_cast<cf_f32x2*>(gb + o) = v;
        const float sq = cf_rowsum16(a0[r] * a0[r] + a1[r] * a1[r]);
        if (rr == 0) gsq[(size_t)row * (Din / 32) + c] = sq;
    }
}

// The weight gradient dW = S^T R of one layer as 16 x 64 wave tiles (TN, K = the 3 RB stacked rows): a workgroup holds two tiles, each cut
// into four K slices of whole 16-row chunks (one batch of loads per wave), combined through LDS in slice order.  `first`: dW1
// (S = S1, R = [coef (.) g | real | fake]), else dW2 (S = S2, R = [dv1 | h1_real | h1_fake]).  The kt == 0 tiles also produce the bias
// gradient (column sums of the real / fake rows of S).
__device__ __forceinline__ void cf_dw_tiles(const CriticFused& p, const bool first, const int wg, f32x4 (*red)[3][4][64], float (*cred)[3][16],
                                            float* coef_rows) {
    CF_COMMON;
    const int N = first ? H1 : H2, K = first ? Din : H1;
    const int ktiles = K / 64;
    const int tl = ks >> 2, sl = ks & 3;               // tile of the workgroup, K slice of the tile
    const int w = wg * 2 + tl;                         // wave tile; the caller launches (N / 16) * ktiles / 2 workgroups for this role
    const int nt = w / ktiles, kt = w - nt * ktiles;
    const int n0 = nt * 16, k0 = kt * 64;
    const float* S = first ? s1b : s2b;                // A operand [3 RB][N]
    const int chunks = 3 * RG, cps = (chunks + 3) / 4; // 16-row chunks in all, per slice
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    float colsum = 0.f;
    // dW1: the penalty coefficient of every x^ row, once per workgroup: 8 threads per row sum the Din / 32 partial sums of squares of
    // launch 4 (fixed order), issued together with the tile's own loads and combined after them
    float n2p = 0.f;
    if (first) {
        const int row = threadIdx.x >> 3, q8 = threadIdx.x & 7, n = Din / 32;
        if (row < RB)
            for (int cc = q8; cc < n; cc += 8) n2p += gsq[(size_t)row * n + cc];
    }
    // chunk q: row block q / RG, rows (q % RG) * 16 ..  - at most 3 chunks per slice (RG <= 4), all loads before the first MFMA
    float a[3][4];
    f32x4 b[3][4];
    bool plain[3], pen[3];
    int m0s[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = sl * cps + j;
        const bool live = j < cps && q < chunks;
        const int qc = live ? q : 0;
        const int blk = qc / RG, m0 = (qc - blk * RG) * 16;
        plain[j] = blk > 0;
        pen[j] = first && blk == 0;
        m0s[j] = m0;
        const float* Rb = first ? (blk == 0 ? gb : (blk == 1 ? p.real : p.fake)) : (blk == 0 ? dv1b : h1b + (size_t)blk * RB * H1);
        const float* Sq = S + (size_t)(blk * RB + m0 + kq) * N + n0 + rr;   // the workspace holds RB rows per block: no clamp
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = m0 + 4 * s + kq;
            const int mc = m < B ? m : B - 1;                                 // the inputs hold B rows only
            const float av = Sq[(size_t)(4 * s) * N];
            a[j][s] = (live && m < B) ? av : 0.f;
            b[j][s] = *reinterpret_cast<const f32x4*>(Rb + (size_t)mc * K + k0 + 4 * rr);
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // the tile's loads and the coefficient loads are in flight together
    if (first) {   // (block-uniform)
        n2p += __shfl_xor(n2p, 1);
        n2p += __shfl_xor(n2p, 2);
        n2p += __shfl_xor(n2p, 4);
        if ((threadIdx.x & 7) == 0) coef_rows[threadIdx.x >> 3] = cf_coef(n2p, p.lambda, invB);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (pen[j])   // c = coef (.) g
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int m = m0s[j] + 4 * s + kq;
                    b[j][s] *= coef_rows[m < B ? m : B - 1];
                }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (plain[j]) colsum += a[j][s];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = cf_mfma(a[j][s], b[j][s][e], acc[e]);
        }
    colsum += __shfl_xor(colsum, 16);
    colsum += __shfl_xor(colsum, 32);
    if (sl > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[tl][sl - 1][e][lane] = acc[e];
        if (kq == 0) cred[tl][sl - 1][rr] = colsum;
    }
    __syncthreads();
    if (sl != 0) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += red[tl][q][e][lane];
        colsum += cred[tl][q][rr];
    }
    float* dW = first ? p.gW1 : p.gW2;
    float* db = first ? p.gb1 : p.gb2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float* o = dW + (size_t)(n0 + kq * 4 + r) * K + k0 + 4 * rr;
        f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        if (p.accum) v += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = v;
    }
    if (kt == 0 && kq == 0) db[n0 + rr] = p.accum ? db[n0 + rr] + colsum : colsum;
}

// ---- launch 5: workgroups [0, RG * H1/16): du1 = coef (g W1^T) (NT, K = Din), dv1 = m1(x^) (.) du1;  the rest: dW1, db1
// (two workgroups per CU: 384 workgroups in one round - at 150 registers the second 128 waited for the first 256, 12 us)
__global__ __launch_bounds__(CF_THREADS) void critic_fused_p5_kernel(const CriticFused p) {
    __shared__ f32x4 part[2][3][4][64];   // role 1 uses the first 7 * 64 entries
    __shared__ float cred[2][3][16];
    __shared__ float coef_s[16];
    __shared__ float coef_rows[64];
    CF_COMMON;
    const int nrole1 = RG * (H1 / 16);
    if ((int)blockIdx.x >= nrole1) {
        cf_dw_tiles(p, true, blockIdx.x - nrole1, part, cred, coef_rows);
        return;
    }
    f32x4* const part1 = &part[0][0][0][0];
    const int t = blockIdx.x;
    const int g = t / (H1 / 16), c = t - g * (H1 / 16);
    const int r0 = g * 16, col0 = c * 16, klen = Din / CF_WAVES;
    f32x4 acc = cf_nt_partial(gb + (size_t)(r0 + rr) * Din + ks * klen + kq * 4, p.W1 + (size_t)(col0 + rr) * Din + ks * klen + kq * 4, klen);
    if (ks > 0) part1[(ks - 1) * 64 + lane] = acc;
    if (ks == CF_WAVES - 1) {
        // this tile's 16 rows: 4 lanes per row each sum a quarter of the Din / 32 partial sums of squares, in a fixed order
        const int row = lane >> 2, qq = lane & 3, n = Din / 32;
        float n2 = 0.f;
        for (int cc = qq; cc < n; cc += 4) n2 += gsq[(size_t)(r0 + row) * n + cc];
        n2 += __shfl_xor(n2, 1);
        n2 += __shfl_xor(n2, 2);
        if (qq == 0) coef_s[row] = r0 + row < B ? cf_coef(n2, p.lambda, invB) : 0.f;
    }
    __syncthreads();
    if (ks != 0) return;
#pragma unroll
    for (int q = 1; q < CF_WAVES; ++q) acc += part1[(q - 1) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + kq * 4 + r;
        const size_t o = (size_t)row * H1 + col0 + rr;
        dv1b[o] = cf_mask(h1b[o], slope) * coef_s[kq * 4 + r] * acc[r];   // h1b block 0 = the x^ rows
    }
}

// ---- launch 6: workgroups [0, H2/16): du2 = dv1 W2^T (NT, K = H1) for ALL rows of 16 columns, e = m2(x^) (.) du2, dw3 = colsum e + plain part;
//                workgroup H2/16: gp, the losses, db3;  the rest: dW2, db2
__global__ __launch_bounds__(CF_THREADS) void critic_fused_p6_kernel(const CriticFused p) {
    __shared__ f32x4 part[2][3][4][64];
    __shared__ float cred[2][3][16];
    __shared__ float csum[4][16];
    CF_COMMON;
    const int nrole1 = H2 / 16;
    if ((int)blockIdx.x > nrole1) {
        cf_dw_tiles(p, false, blockIdx.x - nrole1 - 1, part, cred, nullptr);
        return;
    }
    if ((int)blockIdx.x == nrole1) {   // one wave: gp from launch 4's sums of squares, the losses from launch 2's row dots
        if (ks != 0) return;
        float pen = 0.f, sr = 0.f, sf = 0.f;
        if (lane < B) {
            float n2 = 0.f;
            for (int c = 0; c < Din / 32; ++c) n2 += gsq[(size_t)lane * (Din / 32) + c];
            const float n = sqrtf(n2);
            pen = (n - 1.f) * (n - 1.f);
            for (int c = 0; c < H2 / 16; ++c) {
                sr += opart[(size_t)(RB + lane) * (H2 / 16) + c];
                sf += opart[(size_t)(2 * RB + lane) * (H2 / 16) + c];
            }
        }
        pen = cf_wavesum(pen);
        sr = cf_wavesum(sr);
        sf = cf_wavesum(sf);
        if (lane == 0) {
            const float b3 = p.b3[0];
            const float mr = sr * invB + b3, mf = sf * invB + b3, gp = pen * invB;
            p.out[1] = gp;
            p.out[2] = mr;
            p.out[3] = mf;
            p.out[0] = -mr + mf + p.lambda * gp;
            if (!p.accum) p.gb3[0] = 0.f;  // sum of do = -1 + 1: the reference's gradient of b3 is exactly zero as well
        }
        return;
    }
    // role 1: 8 waves = (row group) x (K slice)
    f32x4* const part1 = &part[0][0][0][0];
    const int RGW = RG >= 3 ? 4 : RG, KSL = CF_WAVES / RGW;
    const int rg = ks % RGW, sl = ks / RGW;
    const int col0 = blockIdx.x * 16, klen = H1 / KSL;
    const bool rows_live = rg < RG;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (rows_live)
        acc = cf_nt_partial(dv1b + (size_t)(rg * 16 + rr) * H1 + sl * klen + kq * 4, p.W2 + (size_t)(col0 + rr) * H1 + sl * klen + kq * 4, klen);
    if (sl > 0) part1[((sl - 1) * RGW + rg) * 64 + lane] = acc;
    __syncthreads();
    if (sl == 0) {
        for (int q = 1; q < KSL; ++q) acc += part1[((q - 1) * RGW + rg) * 64 + lane];
        float s = 0.f;
        if (rows_live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = rg * 16 + kq * 4 + r;
                const size_t o = (size_t)i * H2 + col0 + rr;
                // plain part of dw3: do h2 of the real (-1/B) and fake (+1/B) rows; x^ part: m2 (.) du2
                if (i < B) s += cf_mask(h2b[o], slope) * acc[r] + invB * (h2b[(size_t)2 * RB * H2 + o] - h2b[(size_t)RB * H2 + o]);
            }
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (kq == 0) csum[rg][rr] = s;
    }
    __syncthreads();
    if (ks == 0 && kq == 0) {
        float s = 0.f;
        for (int q = 0; q < RGW; ++q) s += csum[q][rr];
        p.gw3[col0 + rr] = p.accum ? p.gw3[col0 + rr] + s : s;
    }
}

MIGAN_API int migan_critic_fused_ok(int B, int Din, int H1, int H2) {
    return B >= 1 && B <= 64 && Din % 128 == 0 && H1 % 128 == 0 && H2 % 128 == 0 && Din >= 128 && H1 >= 128 && H2 >= 128;
}
// bytes of `ws` (float scratch, any contents)
MIGAN_API size_t migan_critic_fused_workspace(int B, int Din, int H1, int H2) {
    if (!migan_critic_fused_ok(B, Din, H1, H2)) return 0;
    return cf_layout((B + 15) / 16 * 16, Din, H1, H2).total * sizeof(float);
}
// One critic iteration's forward + backward (see the file header): six launches on `stream`.  Gradients are written into gw1..gb3
// (accumulate != 0: added).  out[4] = d_loss, gp, mean D(real), mean D(fake).  phase: 0 = all six launches; p in 1..6 = launch p alone
// on whatever the workspace holds (timing harness, tools/abi_check.cpp).
MIGAN_API int migan_critic_fused(const float* real, const float* fake, const float* alpha, const float* w1, const float* b1,
                                 const float* w2, const float* b2, const float* w3, const float* b3, float* gw1, float* gb1,
                                 float* gw2, float* gb2, float* gw3, float* gb3, float* out, float* ws, size_t ws_bytes,
                                 int B, int Din, int H1, int H2, float slope, float lambda, int accumulate, int phase, void* stream) {
    if (!migan_critic_fused_ok(B, Din, H1, H2) || ws_bytes < migan_critic_fused_workspace(B, Din, H1, H2)) return (int)hipErrorInvalidValue;
    if (phase < 0 || phase > 6) return (int)hipErrorInvalidValue;
    CriticFused p;
    p.B = B; p.RB = (B + 15) / 16 * 16; p.Din = Din; p.H1 = H1; p.H2 = H2;
    p.slope = slope; p.lambda = lambda; p.accum = accumulate != 0;
    p.real = real; p.fake = fake; p.alpha = alpha;
    p.W1 = w1; p.b1 = b1; p.W2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3;
    p.gW1 = gw1; p.gb1 = gb1; p.gW2 = gw2; p.gb2 = gb2; p.gw3 = gw3; p.gb3 = gb3;
    p.out = out; p.ws = ws;
    hipStream_t st = (hipStream_t)stream;
    const int RG = p.RB / 16;
    const dim3 blk(CF_THREADS);
    if (phase == 0 || phase == 1) { MIGAN_LAUNCH(critic_fused_p1_kernel, dim3(RG * (H1 / 16)), blk, 0, st, p); HIP_LAUNCH_CHECK(); }
    if (phase == 0 || phase == 2) { MIGAN_LAUNCH(critic_fused_p2_kernel, dim3(3 * RG * (H2 / 16)), blk, 0, st, p); HIP_LAUNCH_CHECK(); }
    if (phase == 0 || phase == 3) { MIGAN_LAUNCH(critic_fused_p3_kernel, dim3(3 * RG * (H1 / 32)), blk, 0, st, p); HIP_LAUNCH_CHECK(); }
    if (phase == 0 || phase == 4) { MIGAN_LAUNCH(critic_fused_p4_kernel, dim3(RG * (Din / 32)), blk, 0, st, p); HIP_LAUNCH_CHECK(); }
    if (phase == 0 || phase == 5) {
        MIGAN_LAUNCH(critic_fused_p5_kernel, dim3(RG * (H1 / 16) + (H1 / 16) * (Din / 64) / 2), blk, 0, st, p);
        HIP_LAUNCH_CHECK();
    }
    if (phase == 0 || phase == 6) {
        MIGAN_LAUNCH(critic_fused_p6_kernel, dim3(H2 / 16 + 1 + (H2 / 16) * (H1 / 64) / 2), blk, 0, st, p);
        HIP_LAUNCH_CHECK();
    }
    return 0;
}
