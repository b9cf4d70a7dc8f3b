// Forward of an MLP generator - Linear [-> BatchNorm1d (training mode)] [-> LeakyReLU | Tanh | ...] per layer - at <= 64 rows, one
// launch per layer: wgan_gp.py:42-65 / gan.py:38-61 (100 -> 128 -> 256 -> 512 -> 1024 -> prod(img_shape), BatchNorm1d(out, 0.8)
// on layers 2-4, LeakyReLU(0.2), Tanh), which the critic iterations of wgan_gp.py:163 run under no_grad: 5 Linear + 3 x (statistics,
// finalize, apply) = 14 launches of 3-6 us op by op.  A workgroup owns 16 output columns and ALL rows, so the batch statistics of a
// BatchNorm1d column never leave the workgroup: its 16 waves are (row group) x (K slice), every wave fetches its K slice in batches of
// 128 k values (one round trip to the L2 / Infinity Cache per batch; round 3's kernel took one round per 64 with half the slices: eight
// dependent rounds for K = 1024, 19.6 us for that layer - profiles/r04_abi_check_and_two_rank.txt), the K slices are combined through LDS
// in a fixed order, the column mean and the two-pass variance over the <= 64 rows are taken across the row-group waves through LDS, and
// the running statistics (momentum, unbiased variance) and num_batches_tracked are updated as nn.BatchNorm1d does in training mode.
// (A single persistent launch with grid-wide barriers between the layers was measured slower - 63 us against 61 us for five launches in
// round 3, each barrier writes back and invalidates the L2 - and is gone.)
#include "common.h"

#define MF_WAVES 16
#define MF_THREADS (64 * MF_WAVES)
#define MF_MAX_LAYERS 8
#define MF_TICKETS 1024   // column tiles of a BatchNorm1d layer (N <= 16384); the caller's `tickets` holds this many zeroed words

struct MlpLayer {
    const float *W, *b, *gamma, *beta;
    float *rmean, *rvar;
    long long* nbt;
    int K, N, bn, act;
    float slope, eps, momentum;
};
struct MlpFused {
    int B, RB, nlayers, maxN;
    const float* x;
    float* y;
    float* ws;        // two [RB][maxN] activation buffers (forward without a graph), or the save buffer (see MlpSave)
    int saving;       // 1: every layer's output (and normalised pre-affine value + invstd of the BatchNorm layers) is kept for mlp_fused_bwd
    int layer;        // the layer this launch computes
    int fuse0;        // 1: the launch of layer 1 computes layer 0 on the fly (see mlp_fused_fwd_kernel) and layer 0 has no launch
    unsigned* tickets;  // one per column tile, zero at rest (BatchNorm1d layers: the last row-group workgroup normalises the column block)
    MlpLayer L[MF_MAX_LAYERS];
};
// Save buffer of a forward that will be differentiated: for layer l < last h_l [RB][N_l]; for BatchNorm layers xhat_l [RB][N_l]
// and invstd_l [N_l] (the last layer's output is y itself).  Offsets in floats.
struct MlpSave {
    size_t h, xhat, invstd;   // offsets of layer l's pieces
};
// offsets of layer l (l == nlayers: .h = total size).  A loop over <= 8 layers instead of a table: a table indexed with a run-time
// layer number would live in scratch memory inside the kernels.
static __host__ __device__ inline MlpSave mf_save_at(int RB, int nlayers, const MlpLayer* L, int l) {
    MlpSave S;
    size_t o = 0;
    S.h = S.xhat = S.invstd = 0;
    for (int q = 0; q < nlayers; ++q) {
        const size_t n = (size_t)RB * L[q].N;
        const size_t h = o;
        if (q + 1 < nlayers) o += n;
        const size_t xh = o, is = o + (L[q].bn ? n : 0);
        if (L[q].bn) o += n + (size_t)(L[q].N + 15) / 16 * 16;
        if (q == l) {
            S.h = h;
            S.xhat = xh;
            S.invstd = is;
        }
    }
    if (l >= nlayers) S.h = o;
    return S;
}

__device__ __forceinline__ f32x4 mf_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// K slice [kbeg, kend) of a 16x16 NT tile; ap / wp point at this lane's row at k = 0; K % 4 == 0, so a lane's float4 is inside or
// outside the slice.  NB x 16 k values per batch: all loads of a batch are issued before its first MFMA.
template <int NB>
__device__ __forceinline__ void mf_nt_batch(const float* __restrict__ ap, const float* __restrict__ wp, int k0, int kbeg, int kend, int kq,
                                            f32x4& acc0, f32x4& acc1) {
    f32x4 a[NB], b[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int k = k0 + 16 * u + 4 * kq;
        const bool in = k < kend;
        const int kc = in ? k : kbeg;  // any valid address
        a[u] = *reinterpret_cast<const f32x4*>(ap + kc);
        b[u] = *reinterpret_cast<const f32x4*>(wp + kc);
        if (!in) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);   // every load of the batch is issued before its first MFMA (one round trip, not NB)
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (u & 1) acc1 = mf_mfma(a[u][s], b[u][s], acc1);
            else acc0 = mf_mfma(a[u][s], b[u][s], acc0);
        }
}
__device__ __forceinline__ f32x4 mf_nt_partial(const float* __restrict__ ap, const float* __restrict__ wp, int kbeg, int kend, int kq) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int k0 = kbeg;
    for (; k0 + 64 < kend; k0 += 128) mf_nt_batch<8>(ap, wp, k0, kbeg, kend, kq, acc0, acc1);   // > 64 left: a batch of 128
    if (k0 + 32 < kend) mf_nt_batch<4>(ap, wp, k0, kbeg, kend, kq, acc0, acc1);
    else if (k0 + 16 < kend) mf_nt_batch<2>(ap, wp, k0, kbeg, kend, kq, acc0, acc1);
    else if (k0 < kend) mf_nt_batch<1>(ap, wp, k0, kbeg, kend, kq, acc0, acc1);
    return acc0 + acc1;
}

// One workgroup = one 16 x 16 output tile (column tile x row group), K cut over the 16 waves: (N / 16) * (rows / 16) small workgroups per
// layer instead of N / 16 fat ones - a workgroup that pulls 320 KB through one CU's L1 ran at 20-40 GB/s (20 us for the 1024 -> 1024
// layer, profiles/r04_abi_check_and_two_rank.txt), the same bytes spread over four times the CUs do not.  A BatchNorm1d layer still needs
// whole columns: every row-group workgroup leaves its tile (Linear output + bias) in the layer's output buffer (write-through stores) and
// takes a ticket of its column tile; the last of the rows / 16 arrivers re-reads the column block (sc1 loads) - one row group per wave, the lane
// layout of the MFMA tile - and runs the SAME two-pass statistics, running-statistics update, affine and activation as before
// (bit-identical results, whoever arrives last).  The tickets are back at zero when the launch ends.
__global__ __launch_bounds__(MF_THREADS) void mlp_fused_fwd_kernel(const MlpFused p) {
    __shared__ f32x4 part[(MF_WAVES - 1) * 64];   // partial tiles of the K-slice waves 1 .. 15
    __shared__ float red[4][16];
    __shared__ unsigned last_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rr = lane & 15, kq = lane >> 4;
    const int B = p.B, RB = p.RB;
    const int RG = RB / 16;
    float* const buf0 = p.ws;
    float* const buf1 = p.ws + (size_t)RB * p.maxN;
    const int l = p.layer;
    const MlpLayer& Ly = p.L[l];
    const int K = Ly.K, N = Ly.N;
    const MlpSave S = mf_save_at(RB, p.nlayers, p.L, l), Sp = mf_save_at(RB, p.nlayers, p.L, l > 0 ? l - 1 : 0);
    const float* in = l == 0 ? p.x : (p.saving ? p.ws + Sp.h : ((l - 1) & 1 ? buf1 : buf0));
    float* out = l == p.nlayers - 1 ? p.y : (p.saving ? p.ws + S.h : (l & 1 ? buf1 : buf0));
    const int t = blockIdx.x / RG, rg0 = blockIdx.x - t * RG;   // column tile, row group
    const int col0 = t * 16, colr = col0 + rr;
    const bool cok = colr < N;             // only the last layer may have N % 16 != 0 (a critic's single output)
    const int col = cok ? colr : N - 1;
    const bool last_layer = l == p.nlayers - 1;
    {
        const int klen = ((K + MF_WAVES - 1) / MF_WAVES + 15) / 16 * 16;
        const int kbeg = wave * klen < K ? wave * klen : K, kend = (wave + 1) * klen < K ? (wave + 1) * klen : K;
        const int arow = rg0 * 16 + rr < B ? rg0 * 16 + rr : B - 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (p.fuse0 && l == 1) {
            // Layer 0 (Linear + activation, no BatchNorm, <= 128 inputs, <= 256 outputs: 100 -> 128 of wgan_gp.py:55) inside the launch of
            // layer 1: K = N_0 <= 256 gives every K-slice wave exactly 16 k values = 16 layer-0 columns, whose 16 x 16 tile the wave
            // computes itself with the operands SWAPPED - C' = W_0 tile . X^T leaves lane (rr, kq) with h_0[row rr][column kbeg + 4 kq + r],
            // which is the A fragment of the product with W_1 (no LDS, no store / reload of h_0).  W_1's fragment is fetched in the same
            // round of loads as X and W_0, so the fused launch costs what layer 1 alone did and layer 0's launch (5.8 us) is gone.
            if (kbeg < kend) {
                const MlpLayer& L0 = p.L[0];
                const f32x4 wv = *reinterpret_cast<const f32x4*>(Ly.W + (size_t)col * K + kbeg + 4 * kq);
                const f32x4 bv = L0.b ? *reinterpret_cast<const f32x4*>(L0.b + kbeg + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 h = mf_nt_partial(L0.W + (size_t)(kbeg + rr) * L0.K, p.x + (size_t)arow * L0.K, 0, L0.K, kq);
                f32x4 a;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a[r] = act_apply(h[r] + bv[r], L0.act, L0.slope);
                    acc = mf_mfma(a[r], wv[r], acc);
                }
                if (p.saving && t == 0) {   // the backward reads h_0: the column-tile-0 workgroup of each row group keeps it
                    const int row = rg0 * 16 + rr;
                    *reinterpret_cast<f32x4*>(p.ws + Sp.h + (size_t)row * K + kbeg + 4 * kq) = row < B ? a : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        } else if (kbeg < kend) {
            acc = mf_nt_partial(in + (size_t)arow * K, Ly.W + (size_t)col * K, kbeg, kend, kq);
        }
        if (wave > 0) part[(wave - 1) * 64 + lane] = acc;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int q = 1; q < MF_WAVES; ++q) acc += part[(q - 1) * 64 + lane];
            const float bv = Ly.b ? Ly.b[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rg0 * 16 + kq * 4 + r;
                const bool ok = row < B;
                const float v = acc[r] + bv;
                if (Ly.bn) {
                    // the tile's Linear output, normalised by the column's last arriver: written THROUGH to the device-coherent level
                    // (sc1), no cache-wide release - see the hand-off below
                    if (cok) __hip_atomic_store(out + (size_t)row * N + col, ok ? v : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (last_layer) {
                    if (ok && cok) out[(size_t)row * N + col] = act_apply(v, Ly.act, Ly.slope);
                } else {
                    out[(size_t)row * N + col] = ok ? act_apply(v, Ly.act, Ly.slope) : 0.f;
                }
            }
        }
    }
    if (!Ly.bn) return;   // (layer-uniform)
    // ---- BatchNorm1d: the last row-group workgroup of this column tile to arrive normalises the whole column block
    // Hand-off without agent-scope fences (a release / acquire pair is buffer_wbl2 / buffer_inv of the whole L2: with one per workgroup
    // the 512 -> 1024 layer took 39.7 us instead of 15, profiles/r04_abi_check_and_two_rank.txt): the tile was stored write-through
    // (sc1), this wave waits until its stores are acknowledged (vmcnt(0)), then takes the ticket; the last arriver reads the column
    // block with sc1 loads, which do not hit a stale L2 line (MI355X_MICROARCH.md, "handoff-flag" / "publish-large").
    if (wave == 0) {
        unsigned tk = 0;
        if (RG > 1) {
            __builtin_amdgcn_wave_barrier();   // every lane's stores are issued before the wave waits for them (no code on the hardware)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) tk = __hip_atomic_fetch_add(p.tickets + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) last_s = (RG <= 1 || tk == (unsigned)(RG - 1)) ? 1u : 0u;
    }
    __syncthreads();
    if (last_s == 0u) return;
    if (RG > 1 && threadIdx.x == 0) __hip_atomic_store(p.tickets + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // at rest again
    const int RGW = RG >= 3 ? 4 : RG;      // waves 0 .. RGW-1: one row group each (the lane layout of the MFMA tile)
    const int rg = wave;
    const bool act_wave = wave < RGW, rows_live = rg < RG;
    float v[4], d[4];
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = rg * 16 + kq * 4 + r;
        ok[r] = act_wave && rows_live && row < B;
        v[r] = (act_wave && rows_live && cok) ? __hip_atomic_load(out + (size_t)row * N + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    }
    float mean = 0.f, var = 0.f;
    if (act_wave) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) s += ok[r] ? v[r] : 0.f;
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (kq == 0) red[rg][rr] = s;
    }
    __syncthreads();
    if (act_wave) {
        float s = 0.f;
        for (int q = 0; q < RGW; ++q) s += red[q][rr];
        mean = s / (float)B;
    }
    __syncthreads();
    if (act_wave) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d[r] = v[r] - mean;
            s += ok[r] ? d[r] * d[r] : 0.f;
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (kq == 0) red[rg][rr] = s;
    }
    __syncthreads();
    if (!act_wave) return;
    {
        float s = 0.f;
        for (int q = 0; q < RGW; ++q) s += red[q][rr];
        var = s / (float)B;
    }
    const float invstd = 1.f / sqrtf(var + Ly.eps);
    const float gm = Ly.gamma ? Ly.gamma[col] : 1.f, bt = Ly.beta ? Ly.beta[col] : 0.f;
    const float scale = invstd * gm, shift = bt;
    if (p.saving && rg == 0 && kq == 0 && cok) p.ws[S.invstd + col] = invstd;
    if (rg == 0 && kq == 0 && cok && Ly.rmean) {  // nn.BatchNorm1d, training: momentum update, unbiased variance
        const float unb = B > 1 ? var * (float)B / (float)(B - 1) : var;
        Ly.rmean[col] = (1.f - Ly.momentum) * Ly.rmean[col] + Ly.momentum * mean;
        Ly.rvar[col] = (1.f - Ly.momentum) * Ly.rvar[col] + Ly.momentum * unb;
    }
    if (rows_live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rg * 16 + kq * 4 + r;
            const float pre = d[r] * scale + shift;
            if (p.saving && cok) p.ws[S.xhat + (size_t)row * N + col] = ok[r] ? d[r] * invstd : 0.f;
            if (last_layer) {
                if (ok[r] && cok) out[(size_t)row * N + col] = act_apply(pre, Ly.act, Ly.slope);
            } else if (cok) {
                out[(size_t)row * N + col] = ok[r] ? act_apply(pre, Ly.act, Ly.slope) : 0.f;
            }
        }
    }
    if (Ly.nbt && t == 0 && threadIdx.x == 0) *Ly.nbt += 1;
}

// ------------------------------------------------------------------------------------------------------------------------
// Backward of the same MLP, one launch per phase (the generator iteration of wgan_gp.py:179-193: g_loss = -mean(D(G(z))),
// g_loss.backward()): given dy = d(loss)/d(output) and the forward's save buffer,
//   top        dz_L = dy (.) act'(y);   dpre_L = BatchNorm-backward(dz_L)                       (column-local: a workgroup owns all rows)
//   l = L..2   dh_{l-1} = dpre_l W_l  (NN, K = N_l);  dz = dh (.) act'(h_{l-1});  dpre_{l-1} = BN-backward(dz)  (+ dgamma, dbeta)
//   [l = 1     dx = dpre_1 W_1                                                                     when the input gradient is wanted]
//   last       dW_l = dpre_l^T h_{l-1}  (TN, K = rows),  db_l = column sums of dpre_l              for every layer that wants them
// BatchNorm1d backward (training mode): dpre = gamma invstd (dz - mean_rows dz - xhat mean_rows (dz xhat)).  Parameter
// gradients are written into the caller's buffers (accum: added).  dpre_l is kept for the last phase with a row stride of N_l rounded up to 16
// (zero padded), so a one-column top layer (the critic's output) is an ordinary K = 16 slice of zeros and one live column.
struct MlpBwd {
    int B, RB, nlayers;
    const float *x, *y, *dy, *save;
    float *ws, *dx;
    int accum;          // 1: parameter gradients are added into gW / gb / ggamma / gbeta, 0: written
    MlpLayer L[MF_MAX_LAYERS];
    float *gW[MF_MAX_LAYERS], *gb[MF_MAX_LAYERS], *ggamma[MF_MAX_LAYERS], *gbeta[MF_MAX_LAYERS];
    int ph_lo, ph_hi;   // phase of this launch (ph_lo == ph_hi): 0 = top, j = 1 .. NL the layer l = NL - j of the chain, NL + 1 = weight / bias gradients
};
static __host__ __device__ inline size_t mf_dpre_off(int RB, int nlayers, const MlpLayer* L, int l) {
    size_t o = 0;
    for (int q = 0; q < l && q < nlayers; ++q) o += (size_t)RB * ((L[q].N + 15) / 16 * 16);
    return o;
}

// C[16][32] += A[16][k range] W[k][Nc] (NN): A rows have stride lda (zero padded to 16), W rows beyond R are not read.  NB x 16 k values
// per batch, all loads of a batch before its first MFMA.
template <int NB>
__device__ __forceinline__ void mf_nn_batch(const float* __restrict__ ap, int R, const float* __restrict__ W, int Nc, int col, int k0,
                                            int kend, int kq, f32x4& acc0, f32x4& acc1) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x4 a[NB];
    f32x2 b[NB][4];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int kk = k0 + 16 * u + 4 * kq;                 // the slice ends on a multiple of 16: whole chunks are in or out
        const bool in = k0 + 16 * u < kend;
        a[u] = in ? *reinterpret_cast<const f32x4*>(ap + kk) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = in ? kk + s : 0;
            b[u][s] = *reinterpret_cast<const f32x2*>(W + (size_t)(k < R ? k : R - 1) * Nc + col);
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // every load of the batch is issued before its first MFMA (one round trip, not NB)
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc0 = mf_mfma(a[u][s], b[u][s][0], acc0);
            acc1 = mf_mfma(a[u][s], b[u][s][1], acc1);
        }
}
__device__ __forceinline__ void mf_nn_partial(const float* __restrict__ ap, int R, const float* __restrict__ W, int Nc, int col,
                                              int kbeg, int kend, int kq, f32x4& acc0, f32x4& acc1) {
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    int k0 = kbeg;
    for (; k0 + 32 < kend; k0 += 64) mf_nn_batch<4>(ap, R, W, Nc, col, k0, kend, kq, acc0, acc1);   // > 32 left: a batch of 64
    if (k0 + 16 < kend) mf_nn_batch<2>(ap, R, W, Nc, col, k0, kend, kq, acc0, acc1);
    else if (k0 < kend) mf_nn_batch<1>(ap, R, W, Nc, col, k0, kend, kq, acc0, acc1);
}

__global__ __launch_bounds__(MF_THREADS) void mlp_fused_bwd_kernel(const MlpBwd p) {
    __shared__ f32x4 part[2][(MF_WAVES - 1) * 64];
    __shared__ float red[2][4][32];
    __shared__ unsigned last_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rr = lane & 15, kq = lane >> 4;
    const int B = p.B, RB = p.RB, NL = p.nlayers;
    const int RG = RB / 16;
    const int RGW = RG >= 3 ? 4 : RG;
    const int KSL = MF_WAVES / RGW;
    const int rg = wave % RGW, ks = wave / RGW;
    const bool rows_live = rg < RG;
    const float invB = 1.f / (float)B;

    // ---- top: dpre of the last layer from dy (16 columns x all rows per workgroup; only the ks == 0 waves work)
    if (p.ph_lo <= 0 && 0 <= p.ph_hi) {
        if (blockIdx.x == 0) {   // the chain phases' column-tile tickets (behind the dpre buffers): zero before the first of them runs
            unsigned* tickets = reinterpret_cast<unsigned*>(p.ws + mf_dpre_off(RB, NL, p.L, NL) + 16);
            for (int i = threadIdx.x; i < MF_TICKETS; i += MF_THREADS) tickets[i] = 0u;
        }
        const int l = NL - 1;
        const MlpLayer& Ly = p.L[l];
        const int N = Ly.N, ld = (N + 15) / 16 * 16;
        float* dpre = p.ws + mf_dpre_off(RB, NL, p.L, l);
        const MlpSave S = mf_save_at(RB, NL, p.L, l);
        for (int t = blockIdx.x; t < ld / 16; t += gridDim.x) {
            const int colr = t * 16 + rr;
            const bool cok = colr < N;
            const int col = cok ? colr : N - 1;
            float dz[4], xh[4];
            bool ok[4];
            if (ks == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rg * 16 + kq * 4 + r;
                    ok[r] = rows_live && row < B && cok;
                    const int rc = row < B ? row : B - 1;
                    const float yv = p.y[(size_t)rc * N + col];
                    dz[r] = ok[r] ? p.dy[(size_t)rc * N + col] * act_grad_from_out(yv, Ly.act, Ly.slope) : 0.f;
                    xh[r] = (Ly.bn && ok[r]) ? p.save[S.xhat + (size_t)row * N + col] : 0.f;
                }
            }
            float s1 = 0.f, s2 = 0.f;
            if (Ly.bn) {
                if (ks == 0) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a += dz[r];
                        b += dz[r] * xh[r];
                    }
                    a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                    b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
                    if (kq == 0) {
                        red[0][rg][rr] = a;
                        red[1][rg][rr] = b;
                    }
                }
                __syncthreads();
                if (ks == 0)
                    for (int q = 0; q < RGW; ++q) {
                        s1 += red[0][q][rr];
                        s2 += red[1][q][rr];
                    }
            }
            if (ks == 0 && rows_live) {
                float g = 1.f;
                if (Ly.bn) {
                    g = (Ly.gamma ? Ly.gamma[col] : 1.f) * p.save[S.invstd + col];
                    if (rg == 0 && kq == 0 && cok) {
                        if (p.gbeta[l]) p.gbeta[l][col] = p.accum ? p.gbeta[l][col] + s1 : s1;
                        if (p.ggamma[l]) p.ggamma[l][col] = p.accum ? p.ggamma[l][col] + s2 : s2;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rg * 16 + kq * 4 + r;
                    const float v = Ly.bn ? g * (dz[r] - s1 * invB - xh[r] * s2 * invB) : dz[r];
                    dpre[(size_t)row * ld + colr] = ok[r] ? v : 0.f;
                }
            }
            __syncthreads();
        }
    }

    // ---- l = L-1 .. 1: dpre_{l-1} from dpre_l;  l = 0: dx (when wanted).  One workgroup = (32-column tile) x (16-row group), the K
    // dimension (R = N_l <= 1024: one batch of <= 64 k values per wave, ONE round trip) cut over the 16 waves - Nc / 32 * rows / 16
    // workgroups instead of Nc / 32 fat ones that walked K in four dependent rounds (21 us for the 1024 -> 1024 layer on 32 CUs,
    // profiles/r04_wgan_gp_graph_kernel_stats.txt).  Where the layer below has a BatchNorm1d its backward needs the column sums over ALL
    // rows: the forward kernel's hand-off - every row-group workgroup leaves dz (write-through), takes the column tile's ticket, the last
    // arriver re-reads the column block (sc1 loads) and runs the same sums in the same order whoever it is.
    for (int l = NL - 1; l >= 0; --l) {
        if (l == 0 && !p.dx) break;
        const int ph = NL - l;
        if (ph < p.ph_lo || ph > p.ph_hi) continue;
        const MlpLayer& Ly = p.L[l];
        const int R = Ly.N, ldR = (R + 15) / 16 * 16, Nc = Ly.K;   // T[rows][Nc] = dpre_l[rows][R] W_l[R][Nc]
        const float* dprel = p.ws + mf_dpre_off(RB, NL, p.L, l);
        const int klen = ((ldR + MF_WAVES - 1) / MF_WAVES + 15) / 16 * 16;
        const int kbeg = wave * klen < ldR ? wave * klen : ldR, kend = (wave + 1) * klen < ldR ? (wave + 1) * klen : ldR;
        const int lo = l > 0 ? l - 1 : 0;                               // the layer whose output this gradient belongs to
        const MlpSave So = mf_save_at(RB, NL, p.L, lo);
        float* dprev = l > 0 ? p.ws + mf_dpre_off(RB, NL, p.L, l - 1) : nullptr;
        const int t = blockIdx.x / RG, rg0 = blockIdx.x - t * RG;       // column tile, row group
        if (t >= Nc / 32) continue;
        const int col = t * 32 + 2 * rr;
        const bool bn = l > 0 && p.L[lo].bn;   // layer-uniform
        unsigned* tickets = reinterpret_cast<unsigned*>(p.ws + mf_dpre_off(RB, NL, p.L, NL) + 16);
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        if (kbeg < kend) mf_nn_partial(dprel + (size_t)(rg0 * 16 + rr) * ldR, R, Ly.W, Nc, col, kbeg, kend, kq, a0, a1);
        if (wave > 0) {
            part[0][(wave - 1) * 64 + lane] = a0;
            part[1][(wave - 1) * 64 + lane] = a1;
        }
        __syncthreads();
        if (wave == 0) {
            for (int q = 1; q < MF_WAVES; ++q) {
                a0 += part[0][(q - 1) * 64 + lane];
                a1 += part[1][(q - 1) * 64 + lane];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rg0 * 16 + kq * 4 + r;
                const bool okr = row < B;
                if (l == 0) {
                    if (okr) {
                        p.dx[(size_t)row * Nc + col] = a0[r];
                        p.dx[(size_t)row * Nc + col + 1] = a1[r];
                    }
                } else {
                    const float* h = p.save + So.h + (size_t)row * Nc + col;
                    const float z0 = okr ? a0[r] * act_grad_from_out(h[0], p.L[lo].act, p.L[lo].slope) : 0.f;
                    const float z1 = okr ? a1[r] * act_grad_from_out(h[1], p.L[lo].act, p.L[lo].slope) : 0.f;
                    float* o = dprev + (size_t)row * Nc + col;   // interior widths are multiples of 32: stride == width
                    if (bn && RG > 1) {   // dz, finished by the column tile's last arriver: written THROUGH (sc1), see mlp_fused_fwd_kernel
                        __hip_atomic_store(o, z0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(o + 1, z1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        o[0] = z0;
                        o[1] = z1;
                    }
                }
            }
        }
        if (!bn) continue;
        // ---- BatchNorm1d backward of layer lo over the whole column block, by the last row-group workgroup to arrive
        if (wave == 0) {
            unsigned tk = 0;
            if (RG > 1) {
                __builtin_amdgcn_wave_barrier();   // every lane's stores are issued before the wave waits for them (no code on the hardware)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) tk = __hip_atomic_fetch_add(tickets + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) last_s = (RG <= 1 || tk == (unsigned)(RG - 1)) ? 1u : 0u;
        }
        __syncthreads();
        if (last_s == 0u) continue;
        if (RG > 1 && threadIdx.x == 0) __hip_atomic_store(tickets + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // at rest again
        const int trg = wave;
        const bool act_wave = wave < RGW, live = act_wave && trg < RG;
        float dz[2][4], xh[2][4];
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = trg * 16 + kq * 4 + r;
            ok[r] = live && row < B;
            const float* o = dprev + (size_t)(live ? row : 0) * Nc + col;
            const float* xp = p.save + So.xhat + (size_t)(live ? row : 0) * Nc + col;
            if (RG > 1) {
                dz[0][r] = live ? __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                dz[1][r] = live ? __hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            } else {
                dz[0][r] = live ? o[0] : 0.f;
                dz[1][r] = live ? o[1] : 0.f;
            }
            xh[0][r] = ok[r] ? xp[0] : 0.f;
            xh[1][r] = ok[r] ? xp[1] : 0.f;
        }
        if (act_wave) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a += dz[e][r];
                    b += dz[e][r] * xh[e][r];
                }
                a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
                if (kq == 0) {
                    red[0][trg][2 * rr + e] = a;
                    red[1][trg][2 * rr + e] = b;
                }
            }
        }
        __syncthreads();
        if (live) {
            float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, g[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                for (int q = 0; q < RGW; ++q) {
                    s1[e] += red[0][q][2 * rr + e];
                    s2[e] += red[1][q][2 * rr + e];
                }
                g[e] = (p.L[lo].gamma ? p.L[lo].gamma[col + e] : 1.f) * p.save[So.invstd + col + e];
                if (trg == 0 && kq == 0) {
                    if (p.gbeta[l - 1]) p.gbeta[l - 1][col + e] = p.accum ? p.gbeta[l - 1][col + e] + s1[e] : s1[e];
                    if (p.ggamma[l - 1]) p.ggamma[l - 1][col + e] = p.accum ? p.ggamma[l - 1][col + e] + s2[e] : s2[e];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = trg * 16 + kq * 4 + r;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float v = g[e] * (dz[e][r] - s1[e] * invB - xh[e][r] * s2[e] * invB);
                    dprev[(size_t)row * Nc + col + e] = ok[r] ? v : 0.f;
                }
            }
        }
    }

    // ---- last: weight / bias gradients, one 16 (n) x 64 (k) tile per wave
    if (p.ph_lo <= NL + 1 && NL + 1 <= p.ph_hi) {
        int base = 0;
        for (int l = 0; l < NL; ++l) {
            const MlpLayer& Ly = p.L[l];
            if (!p.gW[l] && !p.gb[l]) continue;
            const int N = Ly.N, K = Ly.K, ld = (N + 15) / 16 * 16;
            const int ktiles = (K + 63) / 64, ntl = (ld / 16) * ktiles;
            const float* dprel = p.ws + mf_dpre_off(RB, NL, p.L, l);
            const float* hin = l == 0 ? p.x : p.save + mf_save_at(RB, NL, p.L, l - 1).h;      // [B or RB][K]
            // rotate the starting wave from layer to layer so the short lists of the small layers do not all land on wave 0
            const int nw = (int)blockDim.x >> 6;   // this phase is launched with 4-wave workgroups: one wave tile per wave on every CU
            const int gstride = gridDim.x * nw;
            const int gw = (blockIdx.x * nw + wave + gstride - base % gstride) % gstride;
            for (int wt = gw; wt < ntl; wt += gstride) {
                const int nt = wt / ktiles, kt = wt - nt * ktiles;
                const int n0 = nt * 16, k0 = kt * 64;
                const int kc = k0 + 4 * rr;
                const bool kok = kc < K;                      // K % 4 == 0: the lane's four columns are all in or all out
                f32x4 acc[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                float colsum = 0.f;
                for (int c0 = 0; c0 * 16 < RB; c0 += 2) {   // two 16-row chunks per batch of loads (<= 2 rounds; 128 registers per lane at 16 waves)
                    float a[2][4];
                    f32x4 b[2][4];
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int m = (c0 + c) * 16 + 4 * s + kq;
                            const int mc = m < B ? m : B - 1;
                            const bool live = (c0 + c) * 16 < RB;
                            a[c][s] = (live && m < B) ? dprel[(size_t)mc * ld + n0 + rr] : 0.f;
                            b[c][s] = (live && kok) ? *reinterpret_cast<const f32x4*>(hin + (size_t)mc * K + kc) : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            colsum += a[c][s];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[e] = mf_mfma(a[c][s], b[c][s][e], acc[e]);
                        }
                }
                if (p.gW[l] && kok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n0 + kq * 4 + r;
                        if (n < N) {
                            float* o = p.gW[l] + (size_t)n * K + kc;
                            f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                            if (p.accum) v += *reinterpret_cast<const f32x4*>(o);
                            *reinterpret_cast<f32x4*>(o) = v;
                        }
                    }
                }
                if (kt == 0 && p.gb[l]) {  // wave-uniform
                    colsum += __shfl_xor(colsum, 16);
                    colsum += __shfl_xor(colsum, 32);
                    if (kq == 0 && n0 + rr < N) p.gb[l][n0 + rr] = p.accum ? p.gb[l][n0 + rr] + colsum : colsum;
                }
            }
            base += ntl;
        }
    }
}

static bool mf_fill_layers(MlpLayer* L, int nlayers, const int* dims, const float* fpar, void* const* ptrs) {
    for (int l = 0; l < nlayers; ++l) {
        MlpLayer& Y = L[l];
        Y.K = dims[4 * l]; Y.N = dims[4 * l + 1]; Y.bn = dims[4 * l + 2]; Y.act = dims[4 * l + 3];
        Y.slope = fpar[3 * l]; Y.eps = fpar[3 * l + 1]; Y.momentum = fpar[3 * l + 2];
        Y.W = (const float*)ptrs[7 * l]; Y.b = (const float*)ptrs[7 * l + 1];
        Y.gamma = (const float*)ptrs[7 * l + 2]; Y.beta = (const float*)ptrs[7 * l + 3];
        Y.rmean = (float*)ptrs[7 * l + 4]; Y.rvar = (float*)ptrs[7 * l + 5]; Y.nbt = (long long*)ptrs[7 * l + 6];
        if (!Y.W || (Y.rmean == nullptr) != (Y.rvar == nullptr)) return false;
    }
    return true;
}
// dims: K, N, has_bn, act per layer;  fpar: slope, eps, momentum per layer;  ptrs: W, b, gamma, beta, running_mean, running_var,
// num_batches_tracked per layer (device pointers in a HOST array; NULL where absent).  Returns 1 when the kernels take the shape:
// B <= 64, <= 8 layers, K % 4 == 0; N % 32 == 0 for every layer but the last (whose N is free: a critic's single output).
MIGAN_API int migan_mlp_fused_ok(int B, int nlayers, const int* dims) {
    if (B < 1 || B > 64 || nlayers < 1 || nlayers > MF_MAX_LAYERS) return 0;
    for (int l = 0; l < nlayers; ++l) {
        const int K = dims[4 * l], N = dims[4 * l + 1];
        if (K < 4 || K % 4 != 0 || N < 1) return 0;
        if (l + 1 < nlayers && N % 32 != 0) return 0;
        if (l > 0 && K != dims[4 * (l - 1) + 1]) return 0;
        if (dims[4 * l + 2] && B < 2) return 0;   // BatchNorm1d in training mode needs more than one row
    }
    return 1;
}
static void mf_dims_to_layers(MlpLayer* L, int nlayers, const int* dims) {
    for (int l = 0; l < nlayers; ++l) {
        L[l].K = dims[4 * l]; L[l].N = dims[4 * l + 1]; L[l].bn = dims[4 * l + 2];
    }
}
// bytes of the forward's `ws`: save == 0 two ping-pong activation buffers; save != 0 the save buffer migan_mlp_fused_bwd reads
MIGAN_API size_t migan_mlp_fused_workspace(int B, int nlayers, const int* dims, int save) {
    if (!migan_mlp_fused_ok(B, nlayers, dims)) return 0;
    const int RB = (B + 15) / 16 * 16;
    if (save) {
        MlpLayer L[MF_MAX_LAYERS];
        mf_dims_to_layers(L, nlayers, dims);
        return (mf_save_at(RB, nlayers, L, nlayers).h + 16) * sizeof(float);
    }
    int maxN = 0;
    for (int l = 0; l < nlayers; ++l) maxN = dims[4 * l + 1] > maxN ? dims[4 * l + 1] : maxN;
    return (size_t)2 * RB * maxN * sizeof(float);
}
MIGAN_API size_t migan_mlp_fused_bwd_workspace(int B, int nlayers, const int* dims) {
    if (!migan_mlp_fused_ok(B, nlayers, dims)) return 0;
    MlpLayer L[MF_MAX_LAYERS];
    mf_dims_to_layers(L, nlayers, dims);
    return (mf_dpre_off((B + 15) / 16 * 16, nlayers, L, nlayers) + 16 + MF_TICKETS) * sizeof(float);   // dpre buffers | pad | tickets
}
// y[B][N_last] = MLP(x[B][K_0]), BatchNorm1d layers in training mode (batch statistics; running statistics and counters updated): one
// launch per layer on `stream`.  ws: migan_mlp_fused_workspace(.., save) bytes.  tickets: 1024 unsigned ints, zeroed ONCE by the caller (the
// kernels leave them zero; needed when a layer has BatchNorm1d).  only: 0 = every layer; 1 + l = layer l alone on whatever
// the workspace holds (timing harness, tools/abi_check.cpp).
MIGAN_API int migan_mlp_fused_fwd(const float* x, float* y, int B, int nlayers, const int* dims, const float* fpar,
                                  void* const* ptrs, float* ws, size_t ws_bytes, int save, unsigned* tickets, int only, void* stream) {
    if (!migan_mlp_fused_ok(B, nlayers, dims) || ws_bytes < migan_mlp_fused_workspace(B, nlayers, dims, save)) return (int)hipErrorInvalidValue;
    if (only < 0 || only > nlayers) return (int)hipErrorInvalidValue;
    MlpFused p;
    p.B = B; p.RB = (B + 15) / 16 * 16; p.nlayers = nlayers; p.maxN = 0; p.saving = save != 0;
    p.x = x; p.y = y; p.ws = ws; p.tickets = tickets;
    if (!mf_fill_layers(p.L, nlayers, dims, fpar, ptrs)) return (int)hipErrorInvalidValue;
    for (int l = 0; l < nlayers; ++l) {
        p.maxN = p.L[l].N > p.maxN ? p.L[l].N : p.maxN;
        if (p.L[l].bn && (tickets == nullptr || (p.L[l].N + 15) / 16 > MF_TICKETS)) return (int)hipErrorInvalidValue;
    }
    // layer 0 inside the launch of layer 1 (see the kernel): Linear + activation without BatchNorm, <= 128 inputs, 16 .. 256 outputs
    p.fuse0 = (nlayers >= 2 && !p.L[0].bn && p.L[0].N % 16 == 0 && p.L[0].N <= 256 && p.L[0].K <= 128 && p.L[1].K == p.L[0].N) ? 1 : 0;
    for (int l = 0; l < nlayers; ++l) {
        if (only && l != only - 1) continue;
        if (!only && l == 0 && p.fuse0) continue;
        p.layer = l;
        MIGAN_LAUNCH(mlp_fused_fwd_kernel, dim3((p.L[l].N + 15) / 16 * (p.RB / 16)), dim3(MF_THREADS), 0, (hipStream_t)stream, p);
        HIP_LAUNCH_CHECK();
    }
    return 0;
}
// Backward of migan_mlp_fused_fwd(.., save = 1): dy [B][N_last] -> parameter gradients written into (accumulate != 0: added to)
// gptrs[4*l] = {dW [N][K], db [N], dgamma, dbeta} (device pointers in a host array, NULL = not wanted) and, when dx != NULL, dx [B][K_0]
// (needs K_0 % 32 == 0).  save: the forward's ws; y: the forward's output; ws: migan_mlp_fused_bwd_workspace() bytes.  One launch per
// phase: top, the chain l = NL-1 .. 1 (.. 0 when dx is wanted), the gradients.  only: 0 = every phase; 1 + ph = phase ph alone.
MIGAN_API int migan_mlp_fused_bwd(const float* x, const float* y, const float* dy, const float* save, float* dx, int B, int nlayers,
                                  const int* dims, const float* fpar, void* const* ptrs, void* const* gptrs, float* ws,
                                  size_t ws_bytes, int accumulate, int only, void* stream) {
    if (!migan_mlp_fused_ok(B, nlayers, dims) || ws_bytes < migan_mlp_fused_bwd_workspace(B, nlayers, dims)) return (int)hipErrorInvalidValue;
    if (dx && dims[0] % 32 != 0) return (int)hipErrorInvalidValue;
    if (only < 0 || only > nlayers + 2) return (int)hipErrorInvalidValue;
    MlpBwd p;
    p.B = B; p.RB = (B + 15) / 16 * 16; p.nlayers = nlayers; p.accum = accumulate != 0;
    p.x = x; p.y = y; p.dy = dy; p.save = save; p.ws = ws; p.dx = dx;
    if (!mf_fill_layers(p.L, nlayers, dims, fpar, ptrs)) return (int)hipErrorInvalidValue;
    int wave_tiles = 0;   // 16 x 64 wave tiles of the gradient phase
    bool any_grad = false;
    for (int l = 0; l < MF_MAX_LAYERS; ++l) {
        p.gW[l] = p.gb[l] = p.ggamma[l] = p.gbeta[l] = nullptr;
        if (l < nlayers) {
            p.gW[l] = (float*)gptrs[4 * l]; p.gb[l] = (float*)gptrs[4 * l + 1];
            p.ggamma[l] = (float*)gptrs[4 * l + 2]; p.gbeta[l] = (float*)gptrs[4 * l + 3];
            if (p.gW[l] || p.gb[l]) {
                any_grad = true;
                wave_tiles += (p.L[l].N + 15) / 16 * ((p.L[l].K + 63) / 64);
            }
        }
    }
    for (int ph = 0; ph <= nlayers + 1; ++ph) {
        if (ph == nlayers && !dx) continue;          // l = 0 computes only dx
        if (ph == nlayers + 1 && !any_grad) continue;
        if (only && ph != only - 1) continue;
        p.ph_lo = p.ph_hi = ph;
        int grid;
        if (ph == 0) grid = (p.L[nlayers - 1].N + 15) / 16;
        else if (ph <= nlayers) grid = p.L[nlayers - ph].K / 32 * (p.RB / 16);
        else grid = (wave_tiles + 3) / 4;   // one wave tile per wave, four waves per workgroup (1712 tiles for the generator: every CU takes part)
        MIGAN_LAUNCH(mlp_fused_bwd_kernel, dim3(grid < 1 ? 1 : grid), dim3(ph == nlayers + 1 ? 256 : MF_THREADS), 0, (hipStream_t)stream, p);
        HIP_LAUNCH_CHECK();
    }
    return 0;
}
