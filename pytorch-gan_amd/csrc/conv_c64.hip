// Weight-stationary 3x3 / stride 1 / pad 1 convolution for 64 -> 64 channels on gfx950 (v_mfma_f32_32x32x2_f32): the residual trunk
// of the SRGAN generator (srgan/models.py:22-30,47: 33 such convs per generator pass at 96 x 96, and as many input gradients - an
// input gradient of this geometry IS this convolution with the taps reversed and the channel roles swapped).
//
// Why not igemm_dma_kernel: with 64 output channels its tile is 64 x 64 (128 x 64 leaves the last round of workgroups half empty), and
// a 64 x 64 x 16 K-tile moves 8 KB of operands through the CU's one vector-memory path for 512 MFMA cycles per wave - 16 B per
// clock and CU, which that path does not sustain next to the loop's barriers (round 5: MFMA-busy 0.65, K loop at 0.77 of the MFMA rate).
// Of those 8 KB, 4 KB are the SAME 147 KB of weights fetched again by every tile, and the other 4 KB are pixels fetched 9 times, once
// per tap.
//
// Here the weights never move: a workgroup is four waves = (32-channel output block nb) x (32-channel input half kh); a wave keeps its
// 32 x 288 slice of the weight matrix in 144 registers for the whole launch.  The workgroup walks DOWN a strip of 32 output columns:
// one step = one output row of the strip (32 pixels x 64 channels), the input rows live in a four-slot LDS ring (34 pixels x 64
// channels each, one new row per step), and the nine taps read the ring at shifted pixel offsets - every input element enters the CU
// once (34/32 with the halo) instead of 9 x (Co / BN) times: 1.1 B per clock and CU.  A step is 144 MFMAs per wave behind ONE barrier.
// The two input halves of an output block are two waves' partial sums: they meet through a 4 KB LDS buffer, and the two waves
// take turns in finishing the tile (bias, activation, store), so both carry the same load.
//
// Because every input element passes through registers exactly once on its way into the ring, a per-channel affine map + LeakyReLU /
// PReLU can be applied there for one FMA and one select per element: the BatchNorm2d -> PReLU between the two convs of a residual
// block (srgan/models.py:23-24) needs no pass of its own over the tensor (`in_scale` / `in_shift` / `in_slope`); the zero padding is
// applied AFTER the map (a padded element is 0, not act(shift)).
#include "common.h"

#define C64_C 64                       // channels in and out
#define C64_TW 32                      // output columns of a strip
#define C64_HW (C64_TW + 2)            // halo row: pixels
#define C64_PS 68                      // floats per pixel in the ring: 64 channels + one 16-B pad, so that the 16-B slot of chunk c of pixel p is (p + c) mod 16 -
                                       // a ds_read_b128 of one chunk by consecutive pixels is conflict-free, and (tap column, q) are IMMEDIATE offsets
#define C64_ROWF (C64_HW * C64_PS)     // floats of one ring slot
#define C64_RING (4 * C64_ROWF)
#define C64_XBUF (2 * 2 * 16 * 64)     // [step parity][output block][accumulator register][lane]

struct C64Geom {
    int N, H, W, strips;               // strips = W / 32
    int steps, spw;                    // steps = N * strips * H (row-steps, the row index fastest); steps per workgroup
    unsigned mg_h, mg_s;               // fastdiv magics: step / H, (step / H) / strips
    int sh_h, sh_s;
    int act;                           // output activation: ACT_NONE / ACT_LRELU / ACT_RELU
    float slope;
    int accum;                         // y += instead of y =
    int in_on, in_act;                 // input map on / its activation (ACT_NONE or ACT_LRELU)
    float in_slope;                    // ... slope when in_slope_ptr == NULL
};

// INMAP: 0 = x as it is, 1 = affine map, 2 = affine map + LeakyReLU / PReLU (compile-time: the staging code has no branches)
template <int INMAP>
// (x is NOT __restrict__: loads from a restrict-qualified read-only pointer may be moved across anything - the compiler sank the next row's
// loads from the head of a step down to their use 64 MFMAs later and waited for them there; as possibly-aliasing loads they stay in front of
// the memory clobber that follows them)
__global__ __launch_bounds__(256, 2) void c64_conv_kernel(const C64Geom g, const float* x,
                                                           const f32x4* __restrict__ wp, const float* __restrict__ bias,
                                                           float* __restrict__ y, const float* __restrict__ in_mean,
                                                           const float* __restrict__ in_invstd, const float* __restrict__ in_gamma,
                                                           const float* __restrict__ in_beta,
                                                           const float* __restrict__ in_slope_ptr) {
    __shared__ __attribute__((aligned(16))) float smem[C64_RING + C64_XBUF + C64_C];
    float* ring = smem;
    float* xbuf = smem + C64_RING;
    float* lbias = smem + C64_RING + C64_XBUF;   // the bias vector: read from LDS in the epilogue (a global load there would tie the stores'
                                                 // completion to the next step's first register writes through the in-order vmcnt)
    const int tid = threadIdx.x;
    int L = (int)blockIdx.x * g.spw;
    const int L1 = L + g.spw < g.steps ? L + g.spw : g.steps;
    if (L >= L1) return;
    if (tid < C64_C) lbias[tid] = bias ? bias[tid] : 0.f;   // (visible after the first step's priming barrier)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int nb = wave & 1, kh = wave >> 1;
    const int H = g.H, W = g.W;

    // ---- this wave's slice of the weights: bq[t][q][e] = w[co = nb*32 + l31][tap t][ci = kh*32 + (2q + h)*4 + e]  (packed by c64_pack_kernel)
    f32x4 bq[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[t][q] = wp[(size_t)((wave * 9 + t) * 4 + q) * 64 + lane];
    // the slice STAYS in registers: without this the compiler re-materialises the 36 loads inside the step loop (they are loads from a
    // const __restrict__ pointer) - 147 KB per workgroup and step from L2 instead of once per launch
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm volatile("" : "+v"(bq[t][q]));
        }

    // ---- staging of one halo row (34 pixels x 16 chunks of 16 B = 544 chunks; thread t: chunks t, t + 256, t + 512): pixel (t >> 4) + 16k,
    // chunk c = t & 15 (the SAME four channels for all of a thread's chunks: its scale / shift live in registers)
    const int st_c = tid & 15, st_p = tid >> 4;
    const int st_off = st_p * C64_PS + st_c * 4;
    const int st_n = tid < (C64_HW * 16 - 512) ? 3 : 2;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    float in_slope = g.in_slope;
    if (INMAP) {
        // scale / shift of this thread's four channels from the normalisation's statistics and affine parameters - the arithmetic of
        // norm_apply_kernel (sc = invstd * gamma, sh = beta - mean * sc, then fmaf(x, sc, sh)): the values that enter the ring are bit for
        // bit the ones the separate apply pass would have written
        const f32x4 mu = *reinterpret_cast<const f32x4*>(in_mean + st_c * 4), is = *reinterpret_cast<const f32x4*>(in_invstd + st_c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[e] = is[e] * (in_gamma ? in_gamma[st_c * 4 + e] : 1.f);
            sh[e] = (in_beta ? in_beta[st_c * 4 + e] : 0.f) - mu[e] * sc[e];
        }
        if (in_slope_ptr) in_slope = *in_slope_ptr;
    }

    // ---- fragment reads: tap (r, s), group q reads halo pixel l31 + s of ring row r, chunk kh*8 + 2q + h: one per-lane base, the rest immediates
    const int rd0 = l31 * C64_PS + (kh * 8 + h) * 4;

    // step -> (image, strip, row)
    int q1 = fastdiv(L, g.mg_h, g.sh_h);
    int oi = L - q1 * H;
    int n = fastdiv(q1, g.mg_s, g.sh_s);
    int j0 = (q1 - n * g.strips) * C64_TW;

    // (branch-free: a padded element loads a clamped - valid - address and is zeroed when it is written to the ring.  With the loads under
    // `if (ok)` the compiler merged them through copies placed right behind the first MFMA of the step, with an `s_waitcnt vmcnt(0)` in
    // front: a global round trip at the head of every step.)
    auto load_row = [&](int row, f32x4 (&v)[3], bool (&ok)[3]) {
        const bool rok = (unsigned)row < (unsigned)H;
        const int rc = row < 0 ? 0 : (row >= H ? H - 1 : row);
        const float* src = x + ((size_t)(n * H + rc) * W) * C64_C + st_c * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int col = j0 - 1 + st_p + 16 * k;
            ok[k] = rok && (unsigned)col < (unsigned)W && k < st_n;
            const int cc = col < 0 ? 0 : (col >= W ? W - 1 : col);
            v[k] = *reinterpret_cast<const f32x4*>(src + (size_t)cc * C64_C);
        }
    };
    auto store_row = [&](int row, const f32x4 (&v)[3], const bool (&ok)[3]) {
        float* dst = ring + ((row + 1) & 3) * C64_ROWF + st_off;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            f32x4 o = v[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = o[e];
                if (INMAP) t = fmaf(t, sc[e], sh[e]);
                if (INMAP == 2) t = t > 0.f ? t : t * in_slope;
                o[e] = ok[k] ? t : 0.f;   // padding is zero AFTER the map
            }
            if (k < 2 || st_n == 3) *reinterpret_cast<f32x4*>(dst + k * 16 * C64_PS) = o;
        }
    };

    const bool simple_relu = g.act == ACT_RELU;
    const float ns = g.act == ACT_NONE ? 1.f : (g.act == ACT_LRELU ? g.slope : 0.f);

    // The finished tile leaves from registers of its own: a global store reads its data registers asynchronously, so registers that are
    // written again right away (the accumulators: the first MFMA of the next step) would need an `s_waitcnt vmcnt` there - and with the next
    // row's loads already in flight behind the stores that wait is a full round trip at the head of every step.
    f32x4 outq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) outq[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* optr = y;   // ... and so does their address (the compiler orders a write of a pending store's ADDRESS registers behind the store as well)
    bool prime = true;
    for (; L < L1; ++L) {
        if (prime) {
            // a new strip (or the workgroup's first step): rows oi-1, oi, oi+1 from scratch.  Nobody reads the ring here: the last step
            // of the previous strip ended in a barrier after its fragment reads.
            f32x4 v[3][3];
            bool ok[3][3];
#pragma unroll
            for (int d = 0; d < 3; ++d) load_row(oi + d - 1, v[d], ok[d]);   // nine loads in flight, one round trip
#pragma unroll
            for (int d = 0; d < 3; ++d) store_row(oi + d - 1, v[d], ok[d]);
            __syncthreads();
            prime = false;
        }
        // the row the NEXT step adds (it stays in this strip when oi + 1 < H): fetched now, written to the ring between the two MFMA blocks
        const bool pf = (L + 1 < L1) && (oi + 1 < H);
        f32x4 pv[3];
        bool pok[3];
        load_row(oi + 2, pv, pok);   // (unconditional: a clamped row when there is nothing to prefetch; only the ring write is under `pf`)
        // the loads are ISSUED here: nothing that touches memory may move across the clobber, so the compiler cannot sink them down to their
        // use (the ring write, 64 MFMAs further down - where their `s_waitcnt` belongs)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        // One tap = four fragment reads issued TOGETHER, then sixteen MFMAs back to back; the next tap's fragments are fetched in front of
        // this tap's MFMAs.  (MFMA runs of 4 between other instructions - one read per run, as the compiler schedules it on its own - held this
        // loop at 0.78 of the MFMA rate: the same figure as the 64 x 64 tiles of igemm_dma_kernel, whose runs are 4 long as well; its
        // 128 x 64 / 128 x 128 tiles with runs of 8 / 16 reach 0.89 / 0.93.)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* rb[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) rb[r] = ring + ((oi + r) & 3) * C64_ROWF + rd0;   // input row oi + r - 1 lives in slot (row + 1) & 3
        f32x4 fa[2][4];
#define C64_LOAD(T, BUF)                                                                                              \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                     \
        fa[BUF][q] = *reinterpret_cast<const f32x4*>(rb[(T) / 3] + ((T) % 3) * C64_PS + q * 8);
#define C64_MFMA(T, BUF)                                                                                              \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                     \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                 \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[T][q][e], fa[BUF][q][e], acc, 0, 0, 0);
#define C64_TAP(T)                                                                                                    \
    if ((T) + 1 < 9) { C64_LOAD((T) + 1, ((T) + 1) & 1) }                                                             \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                                \
    /* ONE wait per tap: the four fragments pass through an empty asm - the compiler waits for all of them in front of it and for   \
       none behind it (its own per-use waits put an s_waitcnt between every four MFMAs) */                                          \
    asm volatile("" : "+v"(fa[(T) & 1][0]), "+v"(fa[(T) & 1][1]), "+v"(fa[(T) & 1][2]), "+v"(fa[(T) & 1][3]));        \
    C64_MFMA(T, (T) & 1)                                                                                              \
    __builtin_amdgcn_sched_group_barrier(0x8, 16, 0);
        C64_LOAD(0, 0)
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        C64_TAP(0) C64_TAP(1) C64_TAP(2) C64_TAP(3)
        // (the map + ds_write of the prefetched row stay HERE: hoisted to the top of the step - as the scheduler does to interleave the
        // VALU work with the MFMAs - they drag their `s_waitcnt vmcnt(0)` in front of the first MFMAs: a full global round trip per step)
        __builtin_amdgcn_sched_barrier(0);
        if (pf) store_row(oi + 2, pv, pok);
        __builtin_amdgcn_sched_barrier(0);
        C64_TAP(4) C64_TAP(5) C64_TAP(6) C64_TAP(7) C64_TAP(8)
#undef C64_TAP
#undef C64_MFMA
#undef C64_LOAD

        // (outq stays LIVE from one finished tile to the next - a read-only use right in front of its next write: its registers are never
        // handed to anything else, so nothing has to wait for the stores that still read them)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            asm volatile("" ::"v"(outq[k]));
        }
        asm volatile("" ::"v"(optr));
        // ---- the two input halves of an output block meet: the wave whose turn it is not leaves its partial tile in LDS
        const int par = L & 1;
        const bool fin = par == kh;   // wave-uniform
        float* xb = xbuf + ((par * 2 + nb) * 16) * 64 + lane;
        if (!fin) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xb[r * 64] = acc[r];
        }
        __syncthreads();
        if (fin) {
            f32x4 bv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) bv[k] = *reinterpret_cast<const f32x4*>(lbias + nb * 32 + 8 * k + 4 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += xb[r * 64];
            optr = y + ((size_t)(n * H + oi) * W + j0 + l31) * C64_C + nb * 32 + 4 * h;
            float* o = optr;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[4 * k + e] + bv[k][e];
                    outq[k][e] = t > 0.f ? t : (simple_relu ? 0.f : t * ns);
                }
            }
            if (g.accum) {
                f32x4 old[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) old[k] = *reinterpret_cast<const f32x4*>(o + 8 * k);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) outq[k][e] += old[k][e];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(o + 8 * k) = outq[k];
        }
        // next step
        if (++oi == H) {
            oi = 0;
            prime = true;
            j0 += C64_TW;
            if (j0 == W) {
                j0 = 0;
                ++n;
            }
        }
    }
}

// wp[wave][tap][q][lane] (16 B each) from the OIHW weight w[64][64][3][3].  flip != 0: the input-gradient form - output channel = the
// weight's INPUT channel, taps reversed: wp holds w[ci][co][2 - r][2 - s] where the forward form holds w[co][ci][r][s].
__global__ void c64_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int flip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * 9 * 4 * 64 * 4) return;
    const int e = i & 3, lane = (i >> 2) & 63, q = (i >> 8) & 3, t = (i >> 10) % 9, wave = (i >> 10) / 9;
    const int l31 = lane & 31, h = lane >> 5, nb = wave & 1, kh = wave >> 1;
    const int co = nb * 32 + l31, ci = kh * 32 + (2 * q + h) * 4 + e, r = t / 3, s_ = t % 3;
    wp[i] = flip ? w[((size_t)(ci * 64 + co) * 3 + (2 - r)) * 3 + (2 - s_)] : w[((size_t)(co * 64 + ci) * 3 + r) * 3 + s_];
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layer with the ACCUMULATORS stationary: dW[co][t][ci] = sum over pixels of dy[p][co] * T(x)[p + tap t][ci].
// GEMM M = 64 (co), N = 576 (tap, ci), K = pixels.  A workgroup owns ALL of dW (64 x 576 = 36 accumulator tiles of 32 x 32: nine per wave,
// wave = (co block mb) x (ci half kh), 144 registers) for its share of the pixels and walks them exactly as the forward kernel does: a strip
// of 32 columns, one row per step, the input rows in the four-slot LDS ring, the nine taps reading the ring at shifted pixels - and dy's row
// beside it.  Per pixel pair: one fragment of dy (co along the lanes) and nine of x, nine MFMAs.  Every x and dy element enters the CU once;
// igemm-style 64 x 128 tiles re-read x 4.5 times and dy 5 times (tiles_n).  The input map T (BatchNorm -> PReLU of the layer in front) is
// applied where x enters the ring, as in the forward kernel: the normalised tensor never exists in memory (srgan/models.py:22-27).
// Output: one [64][576] slab per workgroup; the fixed-order reduction of the general weight-gradient path adds them (deterministic).
template <int INMAP>
__global__ __launch_bounds__(256, 2) void c64_wgrad_kernel(const C64Geom g, const float* x, const float* dy, float* __restrict__ part,
                                                            const float* __restrict__ in_mean, const float* __restrict__ in_invstd,
                                                            const float* __restrict__ in_gamma, const float* __restrict__ in_beta,
                                                            const float* __restrict__ in_slope_ptr) {
    constexpr int DYF = C64_TW * C64_C;   // floats of one dy row slot
    __shared__ __attribute__((aligned(16))) float smem[C64_RING + 2 * DYF];
    float* ring = smem;
    float* dyl = smem + C64_RING;
    const int tid = threadIdx.x;
    int L = (int)blockIdx.x * g.spw;
    const int L1 = L + g.spw < g.steps ? L + g.spw : g.steps;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int mb = wave & 1, kh = wave >> 1;
    const int H = g.H, W = g.W;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int st_c = tid & 15, st_p = tid >> 4;
    const int st_off = st_p * C64_PS + st_c * 4;
    const int st_n = tid < (C64_HW * 16 - 512) ? 3 : 2;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    float in_slope = g.in_slope;
    if (INMAP) {
        // scale / shift of this thread's four channels from the normalisation's statistics and affine parameters - the arithmetic of
        // norm_apply_kernel (sc = invstd * gamma, sh = beta - mean * sc, then fmaf(x, sc, sh)): the values that enter the ring are bit for
        // bit the ones the separate apply pass would have written
        const f32x4 mu = *reinterpret_cast<const f32x4*>(in_mean + st_c * 4), is = *reinterpret_cast<const f32x4*>(in_invstd + st_c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[e] = is[e] * (in_gamma ? in_gamma[st_c * 4 + e] : 1.f);
            sh[e] = (in_beta ? in_beta[st_c * 4 + e] : 0.f) - mu[e] * sc[e];
        }
        if (in_slope_ptr) in_slope = *in_slope_ptr;
    }
    int q1 = fastdiv(L < g.steps ? L : 0, g.mg_h, g.sh_h);
    int oi = (L < g.steps ? L : 0) - q1 * H;
    int n = fastdiv(q1, g.mg_s, g.sh_s);
    int j0 = (q1 - n * g.strips) * C64_TW;

    auto load_row = [&](int row, f32x4 (&v)[3], bool (&ok)[3]) {
        const bool rok = (unsigned)row < (unsigned)H;
        const int rc = row < 0 ? 0 : (row >= H ? H - 1 : row);
        const float* src = x + ((size_t)(n * H + rc) * W) * C64_C + st_c * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int col = j0 - 1 + st_p + 16 * k;
            ok[k] = rok && (unsigned)col < (unsigned)W && k < st_n;
            const int cc = col < 0 ? 0 : (col >= W ? W - 1 : col);
            v[k] = *reinterpret_cast<const f32x4*>(src + (size_t)cc * C64_C);
        }
    };
    auto store_row = [&](int row, const f32x4 (&v)[3], const bool (&ok)[3]) {
        float* dst = ring + ((row + 1) & 3) * C64_ROWF + st_off;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            f32x4 o = v[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = o[e];
                if (INMAP) t = fmaf(t, sc[e], sh[e]);
                if (INMAP == 2) t = t > 0.f ? t : t * in_slope;
                o[e] = ok[k] ? t : 0.f;
            }
            if (k < 2 || st_n == 3) *reinterpret_cast<f32x4*>(dst + k * 16 * C64_PS) = o;
        }
    };
    // dy row oi of the strip: 32 pixels x 16 chunks = 512 chunks, two per thread (pixel (t >> 4) + 16 k, chunk t & 15)
    auto load_dy = [&](int row, f32x4 (&v)[2]) {
        const float* src = dy + ((size_t)(n * H + row) * W + j0 + st_p) * C64_C + st_c * 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + (size_t)(16 * k) * C64_C);
    };
    auto store_dy = [&](int slot, const f32x4 (&v)[2]) {
        float* dst = dyl + slot * DYF + st_p * C64_C + st_c * 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) *reinterpret_cast<f32x4*>(dst + k * 16 * C64_C) = v[k];
    };

    const int a_rd = h * C64_C + mb * 32 + l31;                 // dy fragment: pixel 2s + h, co mb*32 + l31
    const int b_rd = h * C64_PS + kh * 32 + l31;                // x fragment: halo pixel 2s + h (+ tap column), ci kh*32 + l31
    bool prime = true;
    int dslot = 0;
    for (; L < L1; ++L) {
        if (prime) {
            f32x4 v[3][3], dv[2];
            bool ok[3][3];
#pragma unroll
            for (int d = 0; d < 3; ++d) load_row(oi + d - 1, v[d], ok[d]);
            load_dy(oi, dv);
#pragma unroll
            for (int d = 0; d < 3; ++d) store_row(oi + d - 1, v[d], ok[d]);
            store_dy(dslot, dv);
            __syncthreads();
            prime = false;
        }
        const bool pf = (L + 1 < L1) && (oi + 1 < H);
        f32x4 pv[3], pdv[2];
        bool pok[3];
        load_row(oi + 2, pv, pok);
        load_dy(pf ? oi + 1 : oi, pdv);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        const float* ap = dyl + dslot * DYF + a_rd;
#define C64_WG_STEPS(S0, S1)                                                                                            \
    _Pragma("unroll") for (int s_ = S0; s_ < S1; ++s_) {                                                                \
        const float a = ap[2 * s_ * C64_C];                                                                             \
        _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                                 \
            const float b = ring[((oi + t / 3) & 3) * C64_ROWF + (2 * s_ + t % 3) * C64_PS + b_rd];                     \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);                                       \
        }                                                                                                               \
    }
        C64_WG_STEPS(0, 8)
        __builtin_amdgcn_sched_barrier(0);
        if (pf) {
            store_row(oi + 2, pv, pok);
            store_dy(dslot ^ 1, pdv);
        }
        __builtin_amdgcn_sched_barrier(0);
        C64_WG_STEPS(8, 16)
#undef C64_WG_STEPS
        __syncthreads();
        dslot ^= 1;
        if (++oi == H) {
            oi = 0;
            prime = true;
            j0 += C64_TW;
            if (j0 == W) {
                j0 = 0;
                ++n;
            }
        }
    }
    // slab [workgroup][co][tap * 64 + ci]: register r of tile t = row (r & 3) + 8 (r >> 2) + 4 h of block mb, column kh * 32 + l31
    float* out = part + (size_t)blockIdx.x * (C64_C * 9 * C64_C);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            out[(size_t)co * (9 * C64_C) + t * C64_C + kh * 32 + l31] = acc[t][r];
        }
}

// All the Conv2d(64, 64, 3, 1, 1) layers of a step in ONE launch: tab[i] = {w_oihw, wp_fwd, wp_dgrad} (either pack pointer may be NULL);
// grid (144, n).  A residual trunk (srgan/models.py:42-47) is 33 such layers, each needing both forms every step: 66 launches otherwise.
struct C64PackEntry { const float* w; float* wf; float* wd; };
__global__ void c64_pack_multi_kernel(const C64PackEntry* __restrict__ tab) {
    const C64PackEntry en = tab[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;   // 144 blocks x 256 = 36864 elements
    const int e = i & 3, lane = (i >> 2) & 63, q = (i >> 8) & 3, t = (i >> 10) % 9, wave = (i >> 10) / 9;
    const int l31 = lane & 31, h = lane >> 5, nb = wave & 1, kh = wave >> 1;
    const int co = nb * 32 + l31, ci = kh * 32 + (2 * q + h) * 4 + e, r = t / 3, s_ = t % 3;
    if (en.wf) en.wf[i] = en.w[((size_t)(co * 64 + ci) * 3 + r) * 3 + s_];
    if (en.wd) en.wd[i] = en.w[((size_t)(ci * 64 + co) * 3 + (2 - r)) * 3 + (2 - s_)];
}

// 1 when migan_c64_conv_fwd takes the layer: Conv2d(64, 64, 3, 1, 1) on W % 32 == 0 columns with enough row-steps to fill the chip
static bool c64_geom_ok(int N, int H, int W) {
    return N >= 1 && H >= 1 && W >= 32 && W % 32 == 0 && (size_t)N * H * W * 64 < (1ull << 31);
}
MIGAN_API int migan_c64_conv_ok(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad_t, int pad_l, int pad_b, int pad_r,
                                int gather) {
    if (Ci != 64 || Co != 64 || R != 3 || S != 3 || stride != 1 || pad_t != 1 || pad_l != 1 || pad_b != 1 || pad_r != 1) return 0;
    if (gather != GATHER_ZERO || !c64_geom_ok(N, H, W)) return 0;
    const long steps = (long)N * (W / 32) * H;
    // at least two workgroups per CU with a handful of rows each; smaller layers: the general kernels.  (MIGAN_C64_MIN_STEPS: the parity
    // tests lower the gate so that small shapes exercise this kernel - tests/conftest.py)
    static const long min_steps = getenv("MIGAN_C64_MIN_STEPS") ? atol(getenv("MIGAN_C64_MIN_STEPS")) : 1024;
    return steps >= min_steps ? 1 : 0;
}
MIGAN_API size_t migan_c64_pack_floats(void) { return (size_t)4 * 9 * 4 * 64 * 4; }
MIGAN_API int migan_c64_pack(const float* w_oihw, float* wp, int flip, void* stream) {
    MIGAN_LAUNCH(c64_pack_kernel, dim3(cdiv(4 * 9 * 4 * 64 * 4, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, wp, flip);
    HIP_LAUNCH_CHECK();
    return 0;
}
MIGAN_API int migan_c64_pack_multi(const void* tab, int n, void* stream) {
    if (n <= 0) return 0;
    if (n > 65535) return (int)hipErrorInvalidValue;
    MIGAN_LAUNCH(c64_pack_multi_kernel, dim3(144, n), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const C64PackEntry*>(tab));
    HIP_LAUNCH_CHECK();
    return 0;
}
// y[N][H][W][64] = act(conv3x3(T(x), w) + bias), T(v) = in_act((v - in_mean[c]) * in_invstd[c] * in_gamma[c] + in_beta[c]) when in_mean != NULL
// (the statistics of migan_norm_stats; in_gamma / in_beta may be NULL = 1 / 0; in_act: ACT_NONE or ACT_LRELU with slope *in_slope_ptr, or in_slope
// when in_slope_ptr == NULL), zero padding applied after T.  wp: migan_c64_pack().
// accumulate != 0: y += (the residual sum of srgan/models.py:30 on the input-gradient side).
MIGAN_API int migan_c64_conv_fwd(const float* x, const float* wp, const float* bias, float* y, int N, int H, int W, int act, float slope,
                                 int accumulate, const float* in_mean, const float* in_invstd, const float* in_gamma, const float* in_beta,
                                 int in_act, float in_slope, const float* in_slope_ptr, void* stream) {
    if (!c64_geom_ok(N, H, W)) return (int)hipErrorInvalidValue;   // (any number of row-steps: migan_c64_conv_ok() is the host's size gate)
    if (act != ACT_NONE && act != ACT_LRELU && act != ACT_RELU) return (int)hipErrorInvalidValue;
    if ((in_mean == nullptr) != (in_invstd == nullptr) || (in_act != ACT_NONE && in_act != ACT_LRELU)) return (int)hipErrorInvalidValue;
    C64Geom g = {};
    g.N = N; g.H = H; g.W = W; g.strips = W / 32;
    g.steps = N * g.strips * H;
    // two workgroups per CU; every workgroup the same number of steps where the count divides (SRGAN's trunk: 4608 steps = 512 x 9)
    static const int wgs_env = getenv("MIGAN_C64_WGS") ? atoi(getenv("MIGAN_C64_WGS")) : 512;   // A/B knob
    const int wgs = g.steps < wgs_env ? g.steps : wgs_env;
    g.spw = cdiv(g.steps, wgs);
    fastdiv_magic((unsigned)H, g.mg_h, g.sh_h);
    fastdiv_magic((unsigned)g.strips, g.mg_s, g.sh_s);
    g.act = act; g.slope = slope; g.accum = accumulate;
    g.in_on = in_mean != nullptr; g.in_act = in_act; g.in_slope = in_slope;
    const dim3 grid(cdiv(g.steps, g.spw));
    const f32x4* wq = reinterpret_cast<const f32x4*>(wp);
    hipStream_t st = (hipStream_t)stream;
    if (!in_mean) MIGAN_LAUNCH((c64_conv_kernel<0>), grid, dim3(256), 0, st, g, x, wq, bias, y, in_mean, in_invstd, in_gamma, in_beta, in_slope_ptr);
    else if (in_act == ACT_NONE)
        MIGAN_LAUNCH((c64_conv_kernel<1>), grid, dim3(256), 0, st, g, x, wq, bias, y, in_mean, in_invstd, in_gamma, in_beta, in_slope_ptr);
    else MIGAN_LAUNCH((c64_conv_kernel<2>), grid, dim3(256), 0, st, g, x, wq, bias, y, in_mean, in_invstd, in_gamma, in_beta, in_slope_ptr);
    HIP_LAUNCH_CHECK();
    return 0;
}

int wgrad_reduce_slabs(const float* ws, float* dw, int nslabs, int Co, int T, int Ci, int accum, const float* db_slabs, float* db,
                       int db_nslab, int db_accum, hipStream_t st);   // conv_igemm.hip

static int c64_wgrad_wgs(int N, int H, int W) {
    // one slab (147 KB) per workgroup: on the 96 x 96 trunk (4608 row-steps) 256 workgroups measured 103.4 us against 116.3 us with 512 (the
    // fixed-order reduction reads half the slabs) and 120.3 us for the general kernel; on 384 x 384 maps (73 728 steps) the reduction is noise and
    // two workgroups per CU win: 1284 vs 1328 us (general kernel 1505) - profiles/r06_ab.txt calls 30, 31
    const long steps = (long)N * (W / 32) * H;
    const int knob = MIGAN_KNOB("MIGAN_C64_WGRAD_WGS", 0);
    const int want = knob > 0 ? knob : (steps >= 16384 ? 512 : 256);
    return (int)(steps < want ? steps : want);
}
MIGAN_API size_t migan_c64_wgrad_workspace(int N, int H, int W) {
    if (!c64_geom_ok(N, H, W)) return 0;
    return (size_t)c64_wgrad_wgs(N, H, W) * 64 * 576 * sizeof(float);
}
// dw_oihw [64][64][3][3] (accumulate: +=) = weight gradient of y = conv3x3(T(x), w) from dy [N][H][W][64]; T as in migan_c64_conv_fwd.  db
// (optional) from the caller's column-sum slabs of dy (db_slabs [db_nslab][64], as migan_conv2d_wgrad).  ws: migan_c64_wgrad_workspace().
MIGAN_API int migan_c64_conv_wgrad(const float* x, const float* dy, float* dw_oihw, float* ws, size_t ws_bytes, int N, int H, int W,
                                   int accumulate, float* db, int db_accumulate, const float* db_slabs, int db_nslab,
                                   const float* in_mean, const float* in_invstd, const float* in_gamma, const float* in_beta, int in_act,
                                   float in_slope, const float* in_slope_ptr, void* stream) {
    if (!c64_geom_ok(N, H, W) || ws_bytes < migan_c64_wgrad_workspace(N, H, W)) return (int)hipErrorInvalidValue;
    if ((in_mean == nullptr) != (in_invstd == nullptr) || (in_act != ACT_NONE && in_act != ACT_LRELU)) return (int)hipErrorInvalidValue;
    if (db && !db_slabs) return (int)hipErrorInvalidValue;
    C64Geom g = {};
    g.N = N; g.H = H; g.W = W; g.strips = W / 32;
    g.steps = N * g.strips * H;
    const int wgs = c64_wgrad_wgs(N, H, W);
    g.spw = cdiv(g.steps, wgs);
    fastdiv_magic((unsigned)H, g.mg_h, g.sh_h);
    fastdiv_magic((unsigned)g.strips, g.mg_s, g.sh_s);
    g.in_on = in_mean != nullptr; g.in_act = in_act; g.in_slope = in_slope;
    const int nwg = cdiv(g.steps, g.spw);   // every one of them writes its slab (possibly all zeros)
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(nwg);
    if (!in_mean) MIGAN_LAUNCH((c64_wgrad_kernel<0>), grid, dim3(256), 0, st, g, x, dy, ws, in_mean, in_invstd, in_gamma, in_beta, in_slope_ptr);
    else if (in_act == ACT_NONE)
        MIGAN_LAUNCH((c64_wgrad_kernel<1>), grid, dim3(256), 0, st, g, x, dy, ws, in_mean, in_invstd, in_gamma, in_beta, in_slope_ptr);
    else MIGAN_LAUNCH((c64_wgrad_kernel<2>), grid, dim3(256), 0, st, g, x, dy, ws, in_mean, in_invstd, in_gamma, in_beta, in_slope_ptr);
    HIP_LAUNCH_CHECK();
    return wgrad_reduce_slabs(ws, dw_oihw, nwg, 64, 9, 64, accumulate, db_slabs, db, db_nslab, db_accumulate, st);
}
