// LDS-DMA implicit-GEMM convolution for gfx950 (v_mfma_f32_32x32x2_f32): the main loop of the forward / input-gradient
// conv kernels rebuilt on `buffer_load_dwordx4 ... lds` (16 B per lane straight from HBM/L2 into LDS, no staging
// registers, no ds_write pass), two LDS stages, ONE barrier per K-tile, MFMA fragments by ds_read_b128 on an
// XOR-swizzled, k-permuted image.  Same ConvGeom tap lists / parity classes / coordinate maps and the same epilogue as
// igemm_pipe_kernel (conv_igemm.hip), which stays the path for shapes this kernel does not take.
//
// Reference call sites served: every Conv2d / ConvTranspose2d / conv input gradient with source channels % 4 == 0 and
// >= 32 (dcgan.py:55,59; cyclegan/models.py:28-33,60,75,106-118; pix2pix/models.py:23,39; srgan/models.py:22,38,54,85).
//
// LDS image of one operand tile ([rows][BK = 32 floats], 8 chunks of 16 B per row): chunk (row, kc) lives at chunk index
// row * 8 + (kc ^ ((row >> 1) & 7)).  The LDS-DMA writes lane l of instruction X at byte X * 1024 + l * 16 (wave-uniform
// base + lane * 16 is the only destination form the hardware has), so lane l fetches row X * 8 + (l >> 3), chunk
// kc = (l & 7) ^ f(row): the swizzle is applied on the per-lane SOURCE address and again on the read
// (cdna_hip_programming.md rule 21).  A ds_read_b128 of one chunk column by 32 consecutive rows is then conflict-free
// (its 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} touch 16 distinct 16-B slots of the 256-B bank row).
// k-permutation: lanes 0-31 read chunk 2q, lanes 32-63 chunk 2q+1; MFMA step e of group q multiplies k = 8q + e (lower
// half-wave) and k = 8q + 4 + e (upper) - any bijection of k is a valid order for a sum, and A and B use the same one.
//
// Zero padding, rows beyond M, columns beyond Co and the channel tail of a tap are all "offset outside the buffer":
// the buffer descriptor's range check returns 0 for them, so the gather needs no select and no zero page.
#include "conv_geom.h"
#include <type_traits>
#include <stdlib.h>
#include <math.h>

#define DMA_SENT 0x80000000u  // voffset beyond any buffer this kernel accepts (tensors < 2 GiB)

typedef __attribute__((address_space(3))) void lds_void_t;

// GEMM row m of class cls -> (image, class-local output row, column); see ConvGeom::m2d
__device__ __forceinline__ void dma_decode_m(const ConvGeom& g, int cls, int m, int Ho, int Wo, int& n, int& oi, int& oj) {
    n = fastdiv(m, g.mg_hw[cls], g.sh_hw[cls]);   // 128-pixel blocks do not straddle images (Ho * Wo % 128 == 0)
    const int rem = m - n * Ho * Wo;
    if (g.m2d) {
        const int blk = rem >> 7, i = rem & 127;
        const int bi = fastdiv(blk << 4, g.mg_w[cls], g.sh_w[cls]);   // blk / (Wo / 16)
        const int bj = blk - bi * (Wo >> 4);
        oi = bi * 8 + (i >> 4);
        oj = bj * 16 + (i & 15);
    } else {
        oi = fastdiv(rem, g.mg_w[cls], g.sh_w[cls]);
        oj = rem - oi * Wo;
    }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const void* p, unsigned bytes) {
    // wave-uniform by construction (kernel arguments): no waterfall loop around the buffer ops
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// TAPS_IN: 1 = tap-outer K order (tap, then 32-channel chunks), 4 = channel-chunk outer / tap inner for classes of
// exactly 4 taps (collapsed up-conv forward, 4x4 stride-2 dgrad): the 4 taps' A tiles of one channel chunk overlap by all
// but one pixel row/column and are fetched back to back (L2 hits) - see igemm_pipe_kernel.
// 9 = the same order for 3x3 kernels (one class, nine taps).  Tap-outer, a tile's nine A tiles are the same pixels +-1 re-read 9 x (Ci / BK)
// K-tiles apart - with 160 workgroups per XCD walking 131 KB each per tap (256 channels) nothing is left in the 4 MB L2: SRGAN's
// dgrad 64 -> 256 @192 moved 5.93 GB for a 0.755 GB problem (7.9 x, profiles/r05_pmc_kernels.json) and ran at HBM speed, not at the MFMA
// rate.  Channel-chunk outer, the nine fetches of one 64-byte channel slice of the tile's pixel neighbourhood follow each other.
// NS: LDS stages.  2 = tile kt+1 is fetched while tile kt is multiplied, `vmcnt(0)` + barrier per K-tile (what the
// compiler emits for __syncthreads() with an LDS-DMA in flight) - right whenever several workgroups share a CU.  NS > 2:
// NS-1 tiles in flight, COUNTED `s_waitcnt vmcnt((NS-2) * loads per tile)` + raw s_barrier, so the DMA queue is never
// drained - for launches with one workgroup per CU (few tiles, long K), where the fetch latency of every K-tile is
// otherwise exposed (cdna_hip_programming.md, "Pipelining across barriers").
// SPLITK: blockIdx.z = class * splits + slice; a slice multiplies the K-tiles [slice * KT / splits, (slice+1) * KT / splits)
// and leaves its raw accumulators in a slab of the caller's workspace; the last slice to arrive at the tile's ticket
// (agent-scope release / acquire, cdna_hip_programming.md Guideline 16) adds the slabs in slice order - a fixed order, so
// the result is deterministic - and runs the epilogue.  The ticket resets itself; the workspace needs zeroing once.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int TAPS_IN, bool KTAIL, int OCC, int NS = 2, bool SPLITK = false>
__global__ __launch_bounds__(256, OCC) void igemm_dma_kernel(const ConvGeom g, const float* __restrict__ A,
                                                             const float* __restrict__ Bw,
                                                             const float* __restrict__ bias, float* __restrict__ C,
                                                             unsigned a_bytes, unsigned b_bytes, int splits = 1,
                                                             float* __restrict__ sk_ws = nullptr) {
    constexpr int CPR = BK / 4;                 // 16-B chunks per row
    constexpr int RPI = 64 / CPR;               // rows per DMA instruction (1 KiB)
    constexpr int SH = CPR == 8 ? 1 : 2;        // swizzle: f(row) = (row >> SH) & (CPR - 1)
    constexpr int NQ = BK / 8;                  // fragment groups (8 k values: 4 MFMA steps) per K-tile
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    constexpr int IA = BM / RPI / 4, IB = BN / RPI / 4;  // DMA instructions per wave per K-tile
    static_assert(BK == 32 || BK == 16, "BK");
    static_assert(IA >= 1 && IB >= 1, "tile too small for the DMA mapping");
    constexpr int A_FL = BM * BK, B_FL = BN * BK, ST_FL = A_FL + B_FL;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "tile shape");
    static_assert(NS >= 2 && NS <= 4 && (!SPLITK || TAPS_IN == 1), "pipeline depth / split-K");
    __shared__ __attribute__((aligned(16))) float smem[NS * ST_FL];

    const int tid = threadIdx.x;
    const int cls = SPLITK ? (int)blockIdx.z / splits : (int)blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
    const int slice = SPLITK ? (int)blockIdx.z - cls * splits : 0;
    const int Ho = g.Ho[cls], Wo = g.Wo[cls];
    const int M = g.N * Ho * Wo;
    const int m0 = bx * BM, n0 = by * BN;
    if (m0 >= M) return;
    const int ntap = g.ntap[cls], tapbeg = g.tapbeg[cls];
    const int Ci = g.Ci, Hi = g.Hi, Wi = g.Wi, mode = g.gather;

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const __amdgpu_buffer_rsrc_t rA = dma_rsrc(A, a_bytes), rB = dma_rsrc(Bw, b_bytes);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int tpt = KTAIL ? (Ci + BK - 1) / BK : Ci / BK;  // K-tiles per tap
    const int KT_all = ntap * tpt;
    const int kt_begin = SPLITK ? (int)((long)slice * KT_all / splits) : 0;
    const int KT = (SPLITK ? (int)((long)(slice + 1) * KT_all / splits) : KT_all) - kt_begin;  // K-tiles of this workgroup

    // ---- DMA lane mapping: instruction X = wave * I + i covers rows X*RPI .. X*RPI+RPI-1; this lane: row X*RPI + lane / CPR,
    // chunk kc = (lane % CPR) ^ f(row)
    const int lrow = lane / CPR;
    int a_base[IA], a_pos[IA], a_kc[IA];
    unsigned rowok = 0;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int X = wave * IA + i;
        const int m = m0 + X * RPI + lrow;
        a_kc[i] = ((lane % CPR) ^ (((X * RPI + lrow) >> SH) & (CPR - 1))) * 4;
        a_base[i] = 0;
        a_pos[i] = 0;
        if (m < M) {
            int n, oi, oj;
            dma_decode_m(g, cls, m, Ho, Wo, n, oi, oj);
            a_base[i] = n * Hi * Wi;
            a_pos[i] = ((oi * g.istride) << 16) | (oj * g.istride);
            rowok |= 1u << i;
        }
    }
    unsigned b_off[IB];
    int b_kc[IB];
#pragma unroll
    for (int j = 0; j < IB; ++j) {
        const int X = wave * IB + j;
        const int n = n0 + X * RPI + lrow;
        const int kc = ((lane % CPR) ^ (((X * RPI + lrow) >> SH) & (CPR - 1))) * 4;
        b_kc[j] = kc;
        b_off[j] = n < g.Co ? (unsigned)(n * g.ldw + kc) * 4u : DMA_SENT;
    }
    unsigned a_off[TAPS_IN][IA];  // byte offset of this lane's chunk of row i at channel 0 of the slot's tap (or DMA_SENT)
    int f_wo[TAPS_IN];
    auto setup_tap = [&](int t, unsigned (&aoff)[IA], int& wo) {
        // wave-uniform index into the kernel-argument tables: scalar loads, and a provably uniform soffset for the DMA
        const int dhw = g.dhw[tapbeg + t];
        const int dh = dhw >> 16, dw = (int)(short)(dhw & 0xffff);
        wo = g.wofs[tapbeg + t];
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            int ihs, iws;
            bool ok = (rowok >> i) & 1u;
            ok &= map_bf((a_pos[i] >> 16) + dh, g.HiL, Hi, mode, ihs);
            ok &= map_bf((a_pos[i] & 0xffff) + dw, g.WiL, Wi, mode, iws);
            aoff[i] = ok ? (unsigned)((a_base[i] + ihs * Wi + iws) * Ci + a_kc[i]) * 4u : DMA_SENT;
        }
    };
    int f_t = SPLITK ? kt_begin / tpt : 0, f_c0 = SPLITK ? (kt_begin - f_t * tpt) * BK : 0;  // fetch position: tap, channel chunk
    // issue the DMA of the K-tile at the fetch position (tap slot `slot`) into LDS stage `st`; live = false (NS > 2, past
    // the last tile): the same instruction count with every lane out of range, so the counted waits stay uniform
    auto issue = [&](int st, bool live, auto slot_c) {
        constexpr int SL = decltype(slot_c)::value;
        const unsigned soA = (unsigned)f_c0 * 4u, soB = (unsigned)(f_wo[SL] + f_c0) * 4u;
        float* base = smem + st * ST_FL;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            unsigned vo = a_off[SL][i];
            if (KTAIL) vo = (f_c0 + a_kc[i] < Ci) ? vo : DMA_SENT;
            if (NS > 2) vo = live ? vo : DMA_SENT;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_t*)(base + (wave * IA + i) * 256), 16, (int)vo, (int)soA,
                                                     0, 0);
        }
#pragma unroll
        for (int j = 0; j < IB; ++j) {
            unsigned vo = b_off[j];
            // channel tail: the A chunk is zero there, but the SGPR offset is outside the descriptor's range check, so the weight
            // chunk would be fetched from the next tap - and, for the last row of the last tap, from BEHIND the weight tensor,
            // where a stale NaN times that zero would poison the tile (found by the host execution model, tests/hipemu)
            if (KTAIL) vo = (f_c0 + b_kc[j] < Ci) ? vo : DMA_SENT;
            if (NS > 2) vo = live ? vo : DMA_SENT;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_t*)(base + A_FL + (wave * IB + j) * 256), 16, (int)vo, (int)soB,
                                                     0, 0);
        }
    };

    // ---- fragment read addresses: row = w * T * 32 + i * 32 + l31, chunk (2q + h) ^ f(l31)
    const int fx = (l31 >> SH) & (CPR - 1);
    int a_rd[NQ], b_rd[NQ];  // float offsets of the chunk groups q within a stage (tile i adds i * 32 * BK)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = ((2 * q + h) ^ fx) * 4;
        a_rd[q] = (wm * (TM * 32) + l31) * BK + c;
        b_rd[q] = A_FL + (wn * (TN * 32) + l31) * BK + c;
    }

    auto advance = [&](auto nxt_c) {  // move the fetch position one K-tile on; NXT = tap slot of the new position
        constexpr int NXT = decltype(nxt_c)::value;
        if (TAPS_IN > 1) {
            if (NXT == 0) f_c0 += BK;
        } else {
            f_c0 += BK;
            if (KTAIL ? f_c0 >= Ci : f_c0 == Ci) {
                f_c0 = 0;
                ++f_t;
                if (NS == 2 || f_t < ntap) setup_tap(f_t, a_off[0], f_wo[0]);
            }
        }
    };
    auto multiply = [&](const float* sb) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(sb + a_rd[q] + i * 32 * BK);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(sb + b_rd[q] + j * 32 * BK);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        // weights as the row operand: accumulator tile (i, j) holds CHANNEL (r & 3) + 8 * (r >> 2) + 4 * h of block j in
                        // register r, for PIXEL l31 of block i - four consecutive channels of one pixel per register quad (see the epilogue)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][e], a[i][e], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (NS == 2) {
        if (KT > 0) {
#pragma unroll
            for (int s_ = 0; s_ < TAPS_IN; ++s_) setup_tap(SPLITK ? f_t : s_, a_off[s_], f_wo[s_]);   // (a K slice starts at its own tap)
            issue(0, true, std::integral_constant<int, 0>{});
        }
        auto k_tile = [&](int kt, auto nxt_c) {
            constexpr int NXT = decltype(nxt_c)::value;
            const int cur = kt & 1;
            // tile kt is in LDS stage cur (DMA drained + barrier at the end of the previous iteration / prologue)
            if (kt + 1 < KT) {
                advance(nxt_c);
                issue(cur ^ 1, true, std::integral_constant<int, NXT>{});
            }
            multiply(smem + cur * ST_FL);
            __syncthreads();  // (the compiler drains the LDS-DMA queue, vmcnt(0), in front of the barrier)
        };
        if (KT > 0) __syncthreads();
        if (TAPS_IN == 9) {
            for (int kt = 0; kt < KT; kt += 9) {
                k_tile(kt, std::integral_constant<int, 1 % TAPS_IN>{});
                k_tile(kt + 1, std::integral_constant<int, 2 % TAPS_IN>{});
                k_tile(kt + 2, std::integral_constant<int, 3 % TAPS_IN>{});
                k_tile(kt + 3, std::integral_constant<int, 4 % TAPS_IN>{});
                k_tile(kt + 4, std::integral_constant<int, 5 % TAPS_IN>{});
                k_tile(kt + 5, std::integral_constant<int, 6 % TAPS_IN>{});
                k_tile(kt + 6, std::integral_constant<int, 7 % TAPS_IN>{});
                k_tile(kt + 7, std::integral_constant<int, 8 % TAPS_IN>{});
                k_tile(kt + 8, std::integral_constant<int, 0>{});
            }
        } else if (TAPS_IN > 1) {
            for (int kt = 0; kt < KT; kt += 4) {
                k_tile(kt, std::integral_constant<int, 1 % TAPS_IN>{});
                k_tile(kt + 1, std::integral_constant<int, 2 % TAPS_IN>{});
                k_tile(kt + 2, std::integral_constant<int, 3 % TAPS_IN>{});
                k_tile(kt + 3, std::integral_constant<int, 0>{});
            }
        } else {
            for (int kt = 0; kt < KT; ++kt) k_tile(kt, std::integral_constant<int, 0>{});
        }
    } else {
        static_assert(NS == 2 || TAPS_IN == 1, "deep pipeline: tap-outer order only");
        constexpr int IPT = IA + IB;  // DMA instructions per K-tile and wave
        if (KT > 0) {
            setup_tap(f_t, a_off[0], f_wo[0]);
#pragma unroll
            for (int t = 0; t < NS - 1; ++t) {  // tiles 0 .. NS-2 into stages 0 .. NS-2
                if (t > 0) advance(std::integral_constant<int, 0>{});
                issue(t, t < KT, std::integral_constant<int, 0>{});
            }
        }
        int st_c = 0, st_f = NS - 1;  // stage multiplied this iteration / stage refilled this iteration
        for (int kt = 0; kt < KT; ++kt) {
            // this wave's part of tile kt has landed once at most the NS-2 younger tiles' loads are outstanding; the barrier
            // makes every wave's part visible and says that everybody is done reading the stage refilled below
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NS - 2) * IPT) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            advance(std::integral_constant<int, 0>{});
            issue(st_f, kt + NS - 1 < KT, std::integral_constant<int, 0>{});
            multiply(smem + st_c * ST_FL);
            st_c = st_c + 1 == NS ? 0 : st_c + 1;
            st_f = st_f + 1 == NS ? 0 : st_f + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dead tail fetches (all lanes out of range) before the LDS is reused
    }

    if constexpr (SPLITK) {
        // slab layout: value (i, j, r) of thread tid at ((i * TN + j) * 16 + r) * 256 + tid: every store is one coalesced 1 KiB
        const int tile_id = (cls * (int)gridDim.y + by) * (int)gridDim.x + bx;
        unsigned* tickets = reinterpret_cast<unsigned*>(sk_ws);
        float* slabs = sk_ws + 1024;
        constexpr int SLAB = BM * BN;
        float* mine = slabs + ((size_t)tile_id * splits + slice) * SLAB;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[((i * TN + j) * 16 + r) * 256 + tid] = acc[i][j][r];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(tickets + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            reinterpret_cast<unsigned*>(smem)[0] = t;
        }
        __syncthreads();
        const bool last = reinterpret_cast<unsigned*>(smem)[0] == (unsigned)(splits - 1);
        if (!last) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(tickets + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // at rest again
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const float* sl = slabs + (size_t)tile_id * splits * SLAB;
        for (int s_ = 0; s_ < splits; ++s_, sl += SLAB)  // slice order: a fixed summation order whoever arrives last
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += sl[((i * TN + j) * 16 + r) * 256 + tid];
    }

    // ---- epilogue: bias + activation + optional [N][Co] mask / ReLU mask / accumulate, strided class scatter.
    // A lane owns ONE pixel per 32-row block (the MFMA column l31) and, per 32-channel block, the four channel quads 8k + 4h .. + 3: a
    // 128x128 tile leaves through 16 `global_store_dwordx4` per lane (lanes l31 and l31 + 32 write 32 adjacent bytes of one pixel) instead of
    // 64 `global_store_dword`, and the pixel decode runs once per block, not once per row.  Round 5 measured what the row-per-register
    // form cost: the stores are ISSUE-bound (cdna_hip_programming.md T21) and the vector-memory unit they occupy is the one the LDS-DMA of
    // the CU's other workgroups goes through - 88 us of a 1292 us launch (64 -> 256 @192, srgan/models.py:53), 96 of 1385 (VGG 64 -> 64 @384),
    // 7 of 100 (trunk), added to the K loop's time whatever the phase of the workgroups (profiles/r05_ab.txt calls 26, 27).
    // (Before that, round 5 found `s_waitcnt vmcnt(0)` in front of every store - a per-value `if (bias)` load; the bias quads are now loaded
    // once and USED once right here, so no later join has a load pending.)
    const int Co = g.Co, ostep = g.ostep, HoF = g.HoF, WoF = g.WoF, oh0 = g.oh0[cls], ow0 = g.ow0[cls], m2d = g.m2d, accum = g.accum;
    const unsigned mg_hw = g.mg_hw[cls], mg_w = g.mg_w[cls];
    const int sh_hw = g.sh_hw[cls], sh_w = g.sh_w[cls], act = g.act;
    const float slope = g.slope;
    const float* oscale = g.oscale;
    const float* omask = g.omask;
    const bool vec = (Co & 3) == 0;   // channel quads are whole and 16-byte aligned
    const int cbase = n0 + wn * (TN * 32) + 4 * h;
    f32x4 bq[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = cbase + j * 32 + 8 * k;
            bq[j][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias) {
                if (vec && c < Co) bq[j][k] = *reinterpret_cast<const f32x4*>(bias + c);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) bq[j][k][e] = c + e < Co ? bias[c + e] : 0.f;
            }
        }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            asm volatile("" : "+v"(bq[j][k]));
        }
    const bool linear_out = (ostep == 1 && g.ncls == 1 && !m2d) && !oscale;
    const bool simple = act <= ACT_RELU;
    const float ns = act == ACT_NONE ? 1.f : (act == ACT_LRELU ? slope : 0.f);   // negative-side factor of the simple activations
    const bool relu = act == ACT_RELU;   // its negative side is the constant 0, not v * 0 (-inf * 0 = NaN; act_apply() and torch give 0)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (TM * 32) + i * 32 + l31;
        if (m >= M) continue;
        size_t opix = (size_t)m;
        int n_img = 0;
        if (!linear_out) {
            n_img = fastdiv(m, mg_hw, sh_hw);
            const int rem = m - n_img * Ho * Wo;
            int oi, oj;
            if (m2d) {
                const int blk = rem >> 7, ii = rem & 127;
                const int bi = fastdiv(blk << 4, mg_w, sh_w);
                oi = bi * 8 + (ii >> 4);
                oj = (blk - bi * (Wo >> 4)) * 16 + (ii & 15);
            } else {
                oi = fastdiv(rem, mg_w, sh_w);
                oj = rem - oi * Wo;
            }
            opix = ((size_t)n_img * HoF + (oh0 + oi * ostep)) * WoF + (ow0 + oj * ostep);
        }
        float* crow = C + opix * Co;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = cbase + j * 32 + 8 * k;
                if (c >= Co) continue;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[i][j][4 * k + e] + bq[j][k][e];
                    o[e] = simple ? (v > 0.f ? v : (relu ? 0.f : v * ns)) : act_apply(v, act, slope);
                }
                if (vec) {
                    if (oscale) {
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(oscale + (size_t)n_img * Co + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] *= sc[e];
                    }
                    if (omask) {
                        const f32x4 mk = *reinterpret_cast<const f32x4*>(omask + opix * Co + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = mk[e] > 0.f ? o[e] : 0.f;
                    }
                    if (accum) {
                        const f32x4 old = *reinterpret_cast<const f32x4*>(crow + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += old[e];
                    }
                    // (plain store: `nt`, `sc1` and `nt sc1` buffer stores were measured - every launch slower, the strided ones up to 2x;
                    // profiles/r05_ab.txt call 29)
                    *reinterpret_cast<f32x4*>(crow + c) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (c + e < Co) {
                            float oe = o[e];
                            if (oscale) oe *= oscale[(size_t)n_img * Co + c + e];
                            if (omask) oe = omask[opix * Co + c + e] > 0.f ? oe : 0.f;
                            if (accum) oe += crow[c + e];
                            crow[c + e] = oe;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int BK, int OCC>
static int launch_dma_cfg(const ConvGeom& g, const float* A, const float* Bw, const float* bias, float* C, unsigned a_bytes,
                          unsigned b_bytes, long maxM, hipStream_t st) {
    dim3 grid(cdiv(maxM, BM), cdiv(g.Co, BN), g.ncls);
    // (Several consecutive M-tiles per workgroup - 768 resident workgroups x 3 tiles for the 2304 64x64 tiles of SRGAN's 64 -> 64 trunk
    // at 96x96 instead of 9 tiles per CU on 8 slots - were measured and rejected: 122 vs 114 us forward, 116 vs 110 us input gradient,
    // profiles/r04_ab.txt; the loop also cost every variant 8-14 registers.)
    // (Fewer workgroups per CU than the kernel allows - unused dynamic LDS - so that 9 tiles per CU run as 3 + 3 + 3 or 5 + 4 instead of
    // 8 + 1 / 4 + 4 + 1 with a lone last workgroup: no layer of the three image workloads gains, the trunk loses 3-8 %; profiles/r05_ab.txt
    // call 25.)
    bool tapin = true;
    for (int c = 0; tapin && c < g.ncls; ++c) tapin = g.ntap[c] == 4;
    const bool ktail = g.Ci % BK != 0;
    // nine taps inner: stride-1 3x3 layers on large maps whose source has more channels than one K-tile (MIGAN_DMA_TAPS9=0: tap-outer, A/B knob)
    // (2 = also below 65 536 pixels: the parity tests run small shapes through this order)
    const int taps9_env = MIGAN_KNOB("MIGAN_DMA_TAPS9", 1);
    // Measured per layer (profiles/r06_ab.txt call 14): 64 -> 256 @192 input gradient 1313 -> 1242 us, vgg 64 -> 64 @384 1430 / 1394 -> 1358 / 1338 us;
    // neutral at 256 channels on 64 x 64 / 96 x 96 maps (their pixel neighbourhoods stay in L2 either way); SLOWER for the stride-2 blocks
    // (128 -> 128 @192: 343 -> 368 us) and for 64 x 64 tiles on 96 x 96 maps (106 -> 126 us) - so: stride 1, maps of >= 262 144 pixels.
    const bool tap9 = taps9_env && g.ncls == 1 && g.ntap[0] == 9 && !ktail && g.Ci >= 2 * BK &&
                      ((g.istride == 1 && g.ostep == 1 && maxM >= 262144) || taps9_env == 2);
    // (Three LDS stages with counted waits - two K-tiles in flight, one workgroup of occupancy less - were measured on whole steps for
    // the tap-outer BK = 16 tiles and rejected: DCGAN -0.9 %, CycleGAN -1.6 %, SRGAN -2.8 %, profiles/r04_ab.txt.)
#define DMA_LAUNCH(TI_, KT_)                                                                                         \
    MIGAN_LAUNCH((igemm_dma_kernel<BM, BN, WM, WN, BK, TI_, KT_, OCC>), grid, dim3(256), 0, st, g, A, Bw, bias, C, \
                       a_bytes, b_bytes)
    if (tap9) {
        DMA_LAUNCH(9, false);
    } else if (tapin) {
        if (ktail) DMA_LAUNCH(4, true); else DMA_LAUNCH(4, false);
    } else {
        if (ktail) DMA_LAUNCH(1, true); else DMA_LAUNCH(1, false);
    }
#undef DMA_LAUNCH
    HIP_LAUNCH_CHECK();
    return 0;
}

// Tile choice for the LDS-DMA kernels.  Co-resident workgroups of a CU share its matrix pipes, so what matters is the
// number of tiles a CU has to work through (T / 256, rounded up for the last CU to finish), the padding of the tile grid
// and how dense a wave's MFMA stream is between two barriers (eff: 64 MFMAs per K-tile and wave for 128x128 / 256x64, 32
// for 128x64, 16 for 64x64 and 128x32).  MIGAN_DMA_TILE=BBBNNN forces a tile (A/B knob).
// (192x64 tiles - three 32-row blocks per wave, 48 KB of LDS so that exactly three workgroups fit a CU - for SRGAN's 64 -> 64 trunk at 96x96,
// batch 16 (srgan/models.py:22-27: 2304 64x64 tiles on 2048 slots, or exactly 3 x 256 of these) were measured: 107.0 vs 109.0 us forward, 100.5
// vs 102.9 us input gradient stand-alone, the SRGAN step 198.9 / 198.7 vs 199.4 / 199.7 img/s; profiles/r05_ab.txt call 24.  The second round
// of 256 tiles is not what holds that launch at 0.65 of the MFMA rate; removed.)
struct DmaCand { int bm, bn; double eff; };
static const DmaCand kDmaCands[] = {{128, 128, 1.00}, {128, 64, 0.97}, {64, 64, 0.93}, {128, 32, 0.70}};
static int dma_select(long maxM, int Co, int ncls) {
    // MIGAN_DMA_TILE=KKBBBNNN forces a tile for launches with at least 256 such tiles (A/B knob)
    static const int tile_env = getenv("MIGAN_DMA_TILE") ? atoi(getenv("MIGAN_DMA_TILE")) : 0;
    if (tile_env) {
        const int bm = (tile_env / 1000) % 1000, bn = tile_env % 1000;
        if ((long)cdiv(maxM, bm) * cdiv(Co, bn) * ncls >= 256) return tile_env;
    }
    double best = -1.0, bestT = 0.0;
    int code = 0;
    for (const DmaCand& c : kDmaCands) {
        const long tm = cdiv(maxM, c.bm), tn = cdiv(Co, c.bn);
        const double T = (double)tm * tn * ncls;
        const double util = ((double)maxM / (tm * c.bm)) * ((double)Co / (tn * c.bn));
        const double per_cu = T / 256.0;
        const double balance = per_cu / (double)(long)(per_cu + 0.999999);
        // a workgroup alone on its CU exposes its prologue, epilogue and every barrier (PatchGAN 128->256 @32x32: 84 us with
        // one 128x64 tile per CU, 76 us with two 64x64)
        const double occ = per_cu <= 1.0 ? 0.90 : (per_cu <= 2.0 ? 0.97 : 1.0);
        const double score = c.eff * util * balance * occ;
        if (score > best * 1.0001) {
            best = score;
            bestT = T;
            code = c.bm * 1000 + c.bn;
        }
    }
    // BK = 16 halves the LDS stage: 4-8 instead of 2-5 workgroups per CU cover each other's barriers, prologues and
    // epilogues (profiles/r03_dma_tile_sweep.txt: +4...13 % on every layer with >= 2 tiles per CU); a workgroup alone on
    // its CU wants the longer MFMA run between barriers of BK = 32 (PatchGAN 256->512 @16x16: 87 vs 110 us)
    const int bk = (bestT >= 512.0 && code != 128032) ? 16 : 32;
    return bk * 1000000 + code;
}

// Under-filled launches (at most one 64x64 tile per CU): 64x64x32 tiles, four LDS stages with counted waits (three
// K-tiles in flight per workgroup), and - when the caller supplies a workspace - split-K over up to 512 workgroups with the
// in-kernel ticket reduction.  Serves the DCGAN discriminator convs (dcgan.py:78-80: 64 tiles, K = 576), PatchGAN heads
// (cyclegan/models.py:106-118) and the inner U-Net levels of pix2pix (pix2pix/models.py:62-71: 1-64 pixels, 16.8-33.5 MB of
// weights per layer - a weight-streaming GEMM that needs every CU pulling on HBM).
#define DMA_SK_TICKETS 1024                   // u32 tickets at the head of the workspace (one per output tile)
#define DMA_SK_MAX_SLABS 1024                 // 64x64 fp32 slabs behind them
size_t igemm_dma_splitk_ws_bytes() { return (size_t)DMA_SK_TICKETS * 4 + (size_t)DMA_SK_MAX_SLABS * 64 * 64 * 4; }

static int launch_dma_small(const ConvGeom& g, const float* A, const float* Bw, const float* bias, float* C, unsigned ab,
                            unsigned bb, long maxM, float* ws, size_t ws_bytes, hipStream_t st) {
    const bool ktail = g.Ci % 32 != 0;
    const int tpt = (g.Ci + 31) / 32;
    const long tm = cdiv(maxM, 64), tn = cdiv(g.Co, 64);
    const long T = tm * tn * g.ncls;
    int minKT = 1 << 30;
    for (int c = 0; c < g.ncls; ++c)
        if ((long)g.N * g.Ho[c] * g.Wo[c] > 0 && g.ntap[c] * tpt < minKT) minKT = g.ntap[c] * tpt;
    constexpr int sk_env = 1;
    int S = 1;
    if (ws && sk_env != 0 && T <= DMA_SK_TICKETS && ws_bytes >= igemm_dma_splitk_ws_bytes()) {
        // a slice costs ~0.5 us per K-tile (one 64x64x32 MFMA tile per wave), the last arriver ~0.16 us per 16 KB slab it
        // adds: S ~ sqrt(3 * KT), at most two workgroups per CU and at least 4 K-tiles per slice
        S = (int)(sqrt(3.0 * (double)minKT) + 0.5);
        if (S > 512 / T) S = (int)(512 / T);
        if (S > minKT / 4) S = minKT / 4;
        if (sk_env > 1) S = sk_env;
        if (S > 64) S = 64;
        if ((long)S * T > DMA_SK_MAX_SLABS) S = (int)(DMA_SK_MAX_SLABS / T);
        if (S > minKT) S = minKT;
        if (S < 1) S = 1;
    }
    dim3 grid((unsigned)tm, (unsigned)tn, (unsigned)(g.ncls * S));
#define DMA_SMALL(KT_, SK_)                                                                                             \
    MIGAN_LAUNCH((igemm_dma_kernel<64, 64, 2, 2, 32, 1, KT_, 2, 4, SK_>), grid, dim3(256), 0, st, g, A, Bw, bias, \
                       C, ab, bb, S, ws)
    if (S < 2) return -2;  // not worth cutting: the ordinary tiles
    if (ktail) DMA_SMALL(true, true); else DMA_SMALL(false, true);
#undef DMA_SMALL
    HIP_LAUNCH_CHECK();
    return 0;
}

// Returns -2 when this geometry is not taken by the LDS-DMA kernels (the caller falls through to igemm_pipe_kernel),
// otherwise the launch status.
int launch_igemm_dma(const ConvGeom& g_in, const float* A, const float* Bw, const float* bias, float* C, float* ws,
                     size_t ws_bytes, hipStream_t st) {
    ConvGeom g = g_in;
    {
        // 8 x 16-pixel M blocks (ConvGeom::m2d) for tall column kernels - the R x 1 GEMMs of the width-Toeplitz layers (R = 7 / 9, one
        // column).  (For every stride-1 3x3 layer it was measured 1-3 % slower, profiles/r05_ab.txt call 10: constant 1.)
        constexpr int m2d_env = 1;
        int rows = 0, cols = 0;   // kernel extent from the tap offsets
        if (g.ncls == 1 && g.ntap[0] > 0) {
            int dh0 = 1 << 20, dh1 = -(1 << 20), dw0 = 1 << 20, dw1 = -(1 << 20);
            for (int t = 0; t < g.ntap[0]; ++t) {
                dh0 = g.dh[t] < dh0 ? g.dh[t] : dh0; dh1 = g.dh[t] > dh1 ? g.dh[t] : dh1;
                dw0 = g.dw[t] < dw0 ? g.dw[t] : dw0; dw1 = g.dw[t] > dw1 ? g.dw[t] : dw1;
            }
            rows = dh1 - dh0 + 1; cols = dw1 - dw0 + 1;
        }
        const bool shape_ok = g.ncls == 1 && g.ostep == 1 && g.istride == 1 && !g.accum && g.gather != GATHER_UP2 && g.Ho[0] % 8 == 0 &&
                              g.Wo[0] % 16 == 0 && g.oh0[0] == 0 && g.ow0[0] == 0 && (long)g.N * g.Ho[0] * g.Wo[0] >= 1024;
        g.m2d = shape_ok && ((m2d_env >= 1 && cols == 1 && rows >= 5) || (m2d_env >= 2 && rows >= 3)) ? 1 : 0;
    }
    // (g.stats: the per-tile statistics epilogue for the norm layer behind the conv was ported to this kernel in round 5 and measured
    // against the norm layer's own statistics pass on whole steps: DCGAN 2.530 / 2.550 -> 2.561 / 2.563 ms, SRGAN 82.3 / 81.9 -> 82.4 / 82.1,
    // CycleGAN 139.3 -> 140.3, pix2pix 2.99 -> 3.39 ms (profiles/r05_ab.txt call 21) - the two cross-wave reductions at the end of every
    // tile cost more than one streaming pass saves, as round 2 found on the register-staged kernels.  Removed again; opt-in
    // statistics stay on igemm_pipe_kernel.)
    if (g.stats || g.swz) return -2;
    if (g.Ci % 4 != 0 || g.Ci < 32 || g.ldw % 4 != 0 || g.Co <= 4) return -2;
    for (int c = 0; c < g.ncls; ++c)
        for (int t = 0; t < g.ntap[c]; ++t)
            if (g.wofs[g.tapbeg[c] + t] % 4 != 0) return -2;
    const size_t a_bytes = (size_t)g.N * g.Hi * g.Wi * g.Ci * 4, b_bytes = (size_t)g.Co * g.ldw * 4;
    if (a_bytes >= 0x7ffffff0ull || b_bytes >= 0x7ffffff0ull) return -2;
    long maxM = 0;
    for (int c = 0; c < g.ncls; ++c) {
        const long m = (long)g.N * g.Ho[c] * g.Wo[c];
        if (m > maxM) maxM = m;
    }
    if (maxM == 0) return 0;
    const unsigned ab = (unsigned)a_bytes, bb = (unsigned)b_bytes;
    // split-K pays when the chip is badly under-filled and K is long (profiles/r03_splitk.txt: pix2pix inner levels
    // 164 -> 31 us, DCGAN D.conv4 19 -> 15 us; at 128+ tiles or < 16 K-tiles the slab traffic and the ticket cost more)
    if (ws) {
        const long T64 = (long)cdiv(maxM, 64) * cdiv(g.Co, 64) * g.ncls;
        int minKT = 1 << 30;
        for (int c = 0; c < g.ncls; ++c)
            if ((long)g.N * g.Ho[c] * g.Wo[c] > 0 && g.ntap[c] * ((g.Ci + 31) / 32) < minKT) minKT = g.ntap[c] * ((g.Ci + 31) / 32);
        // (a wider reach - up to 256 / 512 tiles, from 4 / 8 K-tiles - was measured on the DCGAN discriminator convs at batch 128 / 256:
        // 13.6 -> 18.2 us, 14.7 -> 23.7 us, 19.4 -> 21.7 us, the step 2.906 -> 2.931 ms; profiles/r04_ab.txt call 18)
        // (One 64x64 tile per CU with a long reduction - CycleGAN's 256 -> 256 trunk at ONE image, cyclegan.py:28: 256 tiles x 72 K-tiles -
        // cut in two slices per tile was measured: 62.7 vs 60.4 us, profiles/r05_call7_cyclegan_bs1_eager_kernel_stats.txt.  A 64x64 tile
        // fetches 16 KB per 262 kFLOP: at 16 FLOP/B the launch is bound by L2 -> LDS traffic, not by occupancy; removed.)
        if ((T64 <= 64 && minKT >= 16) || (T64 <= 128 && minKT >= 64)) {
            const int rc = launch_dma_small(g, A, Bw, bias, C, ab, bb, maxM, ws, ws_bytes, st);
            if (rc != -2) return rc;
        }
        // (One image per GPU, cyclegan.py:28: the residual trunk's GEMM - M = 4096, N = 256, K = 2304, 256 tiles of 64 x 64, 57 us = half
        // the MFMA rate of the batch-8 launch - was also measured as 64 tiles of 128 x 128 in four K slices, for the large tile's operand
        // reuse: 70 us forward, the step 39.2 vs 35.2 ms, profiles/r05_ab.txt call 14.  Neither occupancy nor L2 -> LDS traffic is what
        // holds this launch back; removed.)
    }
    // (Four LDS stages without a K split for launches of at most one 64 x 64 tile per CU - CycleGAN's trunk at one image: 256 tiles x 72 K-tiles -
    // were measured in round 6: 54.0 vs 54.2 us stand-alone, the recorded step 33.7 vs 32.7 ms: more K-tiles in flight do not help, the launch
    // is bound by the CU's operand path (16 KB per K-tile against 1024 MFMA clocks), not by the latency of one fetch.)
    switch (dma_select(maxM, g.Co, g.ncls)) {
#define DMA_CASE(BK_, BM_, BN_, WM_, WN_, OCC_) \
    case BK_ * 1000000 + BM_ * 1000 + BN_:      \
        return launch_dma_cfg<BM_, BN_, WM_, WN_, BK_, OCC_>(g, A, Bw, bias, C, ab, bb, maxM, st)
        DMA_CASE(32, 128, 128, 2, 2, 2);
        DMA_CASE(32, 128, 64, 2, 2, 3);
        DMA_CASE(32, 64, 64, 2, 2, 5);
        DMA_CASE(32, 128, 32, 4, 1, 4);
        DMA_CASE(16, 128, 128, 2, 2, 4);
        DMA_CASE(16, 128, 64, 2, 2, 5);
        DMA_CASE(16, 64, 64, 2, 2, 8);
#undef DMA_CASE
        default: return -2;
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient on the same LDS-DMA main loop.  GEMM M = Co (rows of dy), N = taps*Ci (columns of the gathered x),
// K = pixels; split-K slabs, XCD-aware block order, bias column sums and the slab layout are wgrad_inc_kernel's
// (conv_igemm.hip), whose fixed-order reduction launch follows unchanged.
//
// LDS image of one K-tile: A [32 pixels][BM] and B [32 pixels][BN], rows contiguous - exactly what a DMA of NHWC data
// produces (a lane fetches 16 B = 4 channels of one pixel; one instruction = 64/(B/4) whole pixel rows), no swizzle: the
// MFMA fragment of k-pair kp is read along the ROW (lanes 0-31: 32 consecutive 4/8-byte items of pixel 2kp, lanes 32-63
// of pixel 2kp+1 - conflict-free).  A ds_read_b64 feeds TWO row blocks: accumulator tile e of a wave holds the rows
// co = base + 2*i + e (i = MFMA row), so the two floats a lane reads are the A operands of two MFMAs; columns alike.
// Each wave owns the 8 consecutive pixels P0 + 8*wave .. +7 of a K-tile; with Wo % 8 == 0 they lie in one image row, so
// the pixel decode (n, oi, oj) is WAVE-UNIFORM (scalar unit) and a lane's gather address is scalar base + lane constant.
// Shapes with Wo % 8 != 0 (a few pixels per image: latency-bound anyway) stay on wgrad_inc_kernel.
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int T> struct FragT;
template <> struct FragT<1> { typedef float type; };
template <> struct FragT<2> { typedef f32x2 type; };
template <> struct FragT<4> { typedef f32x4 type; };
__device__ __forceinline__ float frag_get(float v, int) { return v; }
__device__ __forceinline__ float frag_get(f32x2 v, int e) { return v[e]; }
__device__ __forceinline__ float frag_get(f32x4 v, int e) { return v[e]; }

// (Round 6 also measured three / four LDS stages with counted waits on the 64 x 256 tile - BK 16, four waves, four stages (80 KB, two workgroups
// per CU) and BK 32, eight waves, three stages (120 KB, one workgroup per CU, half the slabs): 282.7 / 294.0 us against 275.1 us, the DCGAN step
// 2.505 / 2.498 against 2.463 ms (profiles/r06_ab.txt call 28).  Under the counters a wave of this kernel is parked 28.5 % of its cycles at the
// tile barrier (igemm_dma_kernel: 8.5-12 %) with an L2 hit rate of 0.667 - but more K-tiles in flight do not shorten that: removed.)
// NW: waves per workgroup (4, or 8 = a 2 x 4 wave grid: twice the waves per SIMD at the same LDS footprint, half the DMA instructions and
// MFMAs per wave and K-tile - for tiles whose LDS stage allows only two workgroups per CU)
template <int BM, int BN, int BK, bool DYS, bool REFL, int OCC, int NW = 4>
__global__ __launch_bounds__(NW * 64, OCC) void wgrad_dma_kernel(const WgradGeom g, const float* __restrict__ X,
                                                             const float* __restrict__ DY, float* __restrict__ part,
                                                             unsigned x_bytes, unsigned dy_bytes) {
    constexpr int WN_ = NW / 2;                    // wave grid 2 x WN_
    constexpr int TM = BM / 2 / 32, TN = BN / WN_ / 32;
    constexpr int CPA = BM / 4, CPB = BN / 4;      // 16-B chunks per pixel row
    constexpr int RPA = 64 / CPA, RPB = 64 / CPB;  // pixel rows per DMA instruction
    constexpr int PPW = BK / NW;                   // consecutive pixel rows of a K-tile owned by one wave (8 or 4)
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    constexpr int IA = PPW / RPA, IB = PPW / RPB;  // DMA instructions per wave per K-tile
    static_assert((BK == 32 || BK == 16) && IA >= 1 && IB >= 1, "BK");
    constexpr int A_FL = BK * BM, B_FL = BK * BN, ST_FL = A_FL + B_FL;
    static_assert(TM >= 1 && TM <= 2 && (TN == 1 || TN == 2 || TN == 4), "tile shape");
    __shared__ __attribute__((aligned(16))) float smem[2 * ST_FL];

    const int tid = threadIdx.x;
    const int T = g.R * g.S;
    const int Ncol = T * g.Ci;
    const int HoWo = g.Ho * g.Wo;
    const int Mpix = g.N * HoWo;
    int split, tile;  // XCD-aware block order, see wgrad_pipe_kernel
    int cls_x = 0;    // DYS with gridDim.y == 1: the phase class rides in blockIdx.x, between split and tile (see launch_wgrad_dma)
    {
        const int tiles = g.tiles_m * g.tiles_n;
        const int ncx = (DYS && gridDim.y == 1) ? 4 : 1;
        const int total = tiles * g.splits * ncx, per = (total + 7) >> 3;
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int lin = xcd * per + k;
        if (k >= per || lin >= total) return;
        split = lin / (tiles * ncx);
        const int rem = lin - split * tiles * ncx;
        cls_x = rem / tiles;
        tile = rem - cls_x * tiles;
    }
    const int co0 = (tile % g.tiles_m) * BM, nc0 = (tile / g.tiles_m) * BN;
    const int p_begin = split * g.pix_per_split < Mpix ? split * g.pix_per_split : Mpix;
    int p_end = p_begin + g.pix_per_split;
    if (p_end > Mpix) p_end = Mpix;
    const int KT = p_end > p_begin ? (p_end - p_begin + BK - 1) / BK : 0;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN_, wn = wave % WN_;
    const int cls = DYS ? (gridDim.y == 1 ? cls_x : (int)blockIdx.y) : 0;
    const int pad_t = DYS ? 1 - (cls >> 1) : g.pad_t, pad_l = DYS ? 1 - (cls & 1) : g.pad_l;
    const int dy_oh0 = cls >> 1, dy_ow0 = cls & 1;
    const __amdgpu_buffer_rsrc_t rX = dma_rsrc(X, x_bytes), rD = dma_rsrc(DY, dy_bytes);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- lane constants of the DMA mapping
    const int cqa = lane % CPA, pra = lane / CPA;
    const int cqb = lane % CPB, prb = lane / CPB;
    const bool a_colok = co0 + cqa * 4 < g.Co;
    int a_lc[IA];  // byte offset of this lane's dy chunk relative to the group's first pixel
#pragma unroll
    for (int j = 0; j < IA; ++j) {
        const int pr = pra + j * RPA;
        a_lc[j] = ((DYS ? 2 * pr : pr) * g.Co + co0 + cqa * 4) * 4;
    }
    int b_dh = 0, b_dw = 0, b_ci = 0;
    bool b_colok = false;
    {
        const int col = nc0 + cqb * 4;
        if (col < Ncol) {
            const int t = col / g.Ci;
            b_ci = col - t * g.Ci;
            const int r = t / g.S, s_ = t - r * g.S;
            b_dh = r - pad_t;
            b_dw = s_ - pad_l;
            b_colok = true;
        }
    }
    int b_lc[IB];   // zero-pad gather: byte offset relative to the group's first source pixel (may be negative)
    int b_dwp[IB];  // dw + pr * stride: column offset of this lane's pixel relative to the group's first source column
#pragma unroll
    for (int j = 0; j < IB; ++j) {
        const int pr = prb + j * RPB;
        b_dwp[j] = b_dw + pr * g.stride;
        b_lc[j] = ((b_dh * g.Wi + b_dwp[j]) * g.Ci + b_ci) * 4;
    }

    // issue the DMA of the K-tile whose first pixel is P0 into LDS stage st
    auto issue = [&](int st, int P0) {
        const int pg = P0 + PPW * wave;  // this wave's PPW pixels (one image row: Wo % 8 == 0), wave-uniform
        const bool gok = pg < p_end;
        const int pc = gok ? pg : 0;
        const int n = fastdiv(pc, g.mg_hw, g.sh_hw);
        const int rem = pc - n * HoWo;
        const int oi = fastdiv(rem, g.mg_w, g.sh_w), oj = rem - oi * g.Wo;
        float* base = smem + st * ST_FL;
        const int ab = DYS ? ((n * g.dy_H + dy_oh0 + 2 * oi) * g.dy_W + dy_ow0 + 2 * oj) * g.Co * 4 : pc * g.Co * 4;
#pragma unroll
        for (int j = 0; j < IA; ++j) {
            const unsigned vo = (gok && a_colok) ? (unsigned)(ab + a_lc[j]) : DMA_SENT;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rD, (lds_void_t*)(base + (wave * IA + j) * 256), 16, (int)vo, 0, 0, 0);
        }
        const int ih = oi * g.stride + b_dh;
        if (REFL) {
            int ihr = ih < 0 ? -ih : ih;
            ihr = ihr >= g.Hi ? 2 * g.Hi - 2 - ihr : ihr;
            const int rowb = (n * g.Hi + ihr) * g.Wi;
#pragma unroll
            for (int j = 0; j < IB; ++j) {
                int iw = oj * g.stride + b_dwp[j];
                iw = iw < 0 ? -iw : iw;
                iw = iw >= g.Wi ? 2 * g.Wi - 2 - iw : iw;
                const unsigned vo = (gok && b_colok) ? (unsigned)(((rowb + iw) * g.Ci + b_ci) * 4) : DMA_SENT;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void_t*)(base + A_FL + (wave * IB + j) * 256), 16, (int)vo,
                                                         0, 0, 0);
            }
        } else {
            const int xb = ((n * g.Hi + oi * g.stride) * g.Wi + oj * g.stride) * g.Ci * 4;
            const bool okh = gok && b_colok && (unsigned)ih < (unsigned)g.Hi;
#pragma unroll
            for (int j = 0; j < IB; ++j) {
                const bool ok = okh && (unsigned)(oj * g.stride + b_dwp[j]) < (unsigned)g.Wi;
                const unsigned vo = ok ? (unsigned)(xb + b_lc[j]) : DMA_SENT;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void_t*)(base + A_FL + (wave * IB + j) * 256), 16, (int)vo,
                                                         0, 0, 0);
            }
        }
    };

    const bool bias_blk = g.bpart != nullptr && nc0 == 0;  // block-uniform
    float bsum = 0.f;
    const int a_rd = h * BM + wm * (TM * 32) + TM * l31;
    const int b_rd = A_FL + h * BN + wn * (TN * 32) + TN * l31;
    typedef typename FragT<TM>::type fa_t;
    typedef typename FragT<TN>::type fb_t;
    if (KT > 0) {
        issue(0, p_begin);
        __syncthreads();
    }
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) issue(cur ^ 1, p_begin + (kt + 1) * BK);
        const float* sb = smem + cur * ST_FL;
        if (bias_blk && tid < BM) {
            float s_ = 0.f;
#pragma unroll
            for (int k = 0; k < BK; ++k) s_ += sb[k * BM + tid];
            bsum += s_;
        }
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const fa_t a = *reinterpret_cast<const fa_t*>(sb + a_rd + kp * 2 * BM);
            const fb_t b = *reinterpret_cast<const fb_t*>(sb + b_rd + kp * 2 * BN);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag_get(a, i), frag_get(b, j), acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if (bias_blk && tid < BM && co0 + tid < g.Co)
        g.bpart[((size_t)cls * g.splits + split) * g.Co + co0 + tid] = bsum;
    // slab [cls][split][co][col]; accumulator tile (e, e') of a wave: co = base + TM*i + e, col = base + TN*l31 + e'
    float* out = part + ((size_t)cls * g.splits + split) * g.Co * Ncol;
#pragma unroll
    for (int e = 0; e < TM; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int co = co0 + wm * (TM * 32) + TM * i + e;
            if (co >= g.Co) continue;
            const int col = nc0 + wn * (TN * 32) + TN * l31;
            float* o = out + (size_t)co * Ncol + col;
            if (TN == 4) {
                if (col + 3 < Ncol) {
                    f32x4 v = {acc[e][0][r], acc[e][1 % TN][r], acc[e][2 % TN][r], acc[e][3 % TN][r]};
                    *reinterpret_cast<f32x4*>(o) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < TN; ++q)
                        if (col + q < Ncol) o[q] = acc[e][q][r];
                }
            } else if (TN == 2) {
                if (col + 1 < Ncol) {
                    f32x2 v = {acc[e][0][r], acc[e][TN - 1][r]};
                    *reinterpret_cast<f32x2*>(o) = v;
                } else if (col < Ncol) {
                    o[0] = acc[e][0][r];
                }
            } else if (col < Ncol) {
                o[0] = acc[e][0][r];
            }
        }
}

// Returns -2 when the LDS-DMA weight-gradient kernel does not take this geometry (caller: wgrad_inc_kernel).
// bm / bn = the tile the caller's plan chose (128x128, 64x128, 64x64); grid_y = 4 for the phase-collapsed up-conv.
int launch_wgrad_dma(const WgradGeom& g, int bm, int bn, bool dys, const float* x, const float* dy, float* ws,
                     hipStream_t st) {
    static const int env = getenv("MIGAN_DMA_WGRAD") ? atoi(getenv("MIGAN_DMA_WGRAD")) : 1;
    if (env == 0) return -2;
    if (g.Wo % 8 != 0 || g.Ci % 4 != 0 || g.Co % 4 != 0 || g.gather == GATHER_UP2 || g.pix_per_split % 32 != 0) return -2;
    const size_t xb = (size_t)g.N * g.Hi * g.Wi * g.Ci * 4;
    const size_t db = dys ? (size_t)g.N * g.dy_H * g.dy_W * g.Co * 4 : (size_t)g.N * g.Ho * g.Wo * g.Co * 4;
    if (xb >= 0x7ffffff0ull || db >= 0x7ffffff0ull) return -2;
    const bool refl = g.gather == GATHER_REFLECT;
    if (dys && refl) return -2;
    const int Ncol = g.R * g.S * g.Ci;
    WgradGeom gg = g;
    gg.tiles_m = cdiv(g.Co, bm);
    gg.tiles_n = cdiv(Ncol, bn);
    // Up-conv (dys): the four phase classes of one pixel range read the SAME x pixels (their 2x2 taps differ) - with the class in grid.y
    // the four launches-worth of workgroups are a whole grid apart and x comes back from the Infinity Cache four times (328.6 MB
    // HBM-side against 201 MB of operands, profiles/r06_pmc_kernels.json).  MIGAN_WGRAD_CLSX=1: the class rides in blockIdx.x between split
    // and tile, so the 4 x tiles_n workgroups of a pixel range are neighbours on ONE XCD and share that XCD's L2.
    static const int clsx_env = getenv("MIGAN_WGRAD_CLSX") ? atoi(getenv("MIGAN_WGRAD_CLSX")) : 1;
    const bool clsx = dys && clsx_env != 0;
    dim3 grid(cdiv(gg.tiles_m * gg.tiles_n * gg.splits * (clsx ? 4 : 1), 8) * 8, dys && !clsx ? 4 : 1);
#define WGD(BM_, BN_, BK_, OCC_)                                                                                             \
    do {                                                                                                                 \
        if (dys) MIGAN_LAUNCH((wgrad_dma_kernel<BM_, BN_, BK_, true, false, OCC_>), grid, dim3(256), 0, st, gg, x, dy, ws, \
                                    (unsigned)xb, (unsigned)db);                                                         \
        else if (refl) MIGAN_LAUNCH((wgrad_dma_kernel<BM_, BN_, BK_, false, true, OCC_>), grid, dim3(256), 0, st, gg, x,   \
                                          dy, ws, (unsigned)xb, (unsigned)db);                                           \
        else MIGAN_LAUNCH((wgrad_dma_kernel<BM_, BN_, BK_, false, false, OCC_>), grid, dim3(256), 0, st, gg, x, dy, ws, \
                                (unsigned)xb, (unsigned)db);                                                             \
    } while (0)
    // 128x128: BK = 16 (32 KB of LDS, 4-5 workgroups per CU); the narrower tiles: BK = 32 (3 / 5 per CU) - measured
    // per layer in profiles/r03_wgrad_dma.txt (differences <= 3 %).
    // wgrad_occ() in conv_igemm.hip (the split planner's slot count) follows this choice.
    const int bk = bm == 128 ? 16 : 32;
    // 64 x 256 (80 KB of LDS: two workgroups per CU): eight waves per workgroup - four per SIMD instead of two cover each other's DMA issue,
    // fragment reads and barriers (MIGAN_WGRAD_NW=4: the four-wave form, A/B knob)
    static const int nw_env = getenv("MIGAN_WGRAD_NW") ? atoi(getenv("MIGAN_WGRAD_NW")) : 8;
    if (bk == 32 && bm == 64 && bn == 256 && nw_env == 8) {
        if (dys) MIGAN_LAUNCH((wgrad_dma_kernel<64, 256, 32, true, false, 4, 8>), grid, dim3(512), 0, st, gg, x, dy, ws, (unsigned)xb,
                              (unsigned)db);
        else if (refl) MIGAN_LAUNCH((wgrad_dma_kernel<64, 256, 32, false, true, 4, 8>), grid, dim3(512), 0, st, gg, x, dy, ws, (unsigned)xb,
                                    (unsigned)db);
        else MIGAN_LAUNCH((wgrad_dma_kernel<64, 256, 32, false, false, 4, 8>), grid, dim3(512), 0, st, gg, x, dy, ws, (unsigned)xb,
                          (unsigned)db);
        HIP_LAUNCH_CHECK();
        return 0;
    }
    if (bk == 32) {
        if (bm == 128 && bn == 128) WGD(128, 128, 32, 2);
        else if (bm == 64 && bn == 256) WGD(64, 256, 32, 2);
        else if (bm == 64 && bn == 128) WGD(64, 128, 32, 3);
        else if (bm == 64 && bn == 64) WGD(64, 64, 32, 5);
        else return -2;
    } else {
        if (bm == 128 && bn == 128) WGD(128, 128, 16, 4);
        else if (bm == 64 && bn == 128) WGD(64, 128, 16, 6);
        else if (bm == 64 && bn == 64) WGD(64, 64, 16, 8);
        else return -2;
    }
#undef WGD
    HIP_LAUNCH_CHECK();
    return 0;
}
