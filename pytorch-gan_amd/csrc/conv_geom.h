// Geometry of one implicit-GEMM convolution launch (tap lists, parity classes, coordinate maps), shared by the
// register-staged kernels (conv_igemm.hip) and the LDS-DMA kernels (conv_dma.hip).
#pragma once
#include "common.h"

#define MAX_TAPS 96
#define MAX_CLS 16

struct ConvGeom {
    int N, Hi, Wi, Ci;   // physical source tensor (NHWC)
    int HiL, WiL;        // logical gather extent (2*Hi for GATHER_UP2)
    int Co;              // GEMM N
    int HoF, WoF;        // full output extent
    int ostep, istride;  // output sub-grid step (parity classes), source step per output index
    int gather, ldw, ncls;
    int accum;           // epilogue adds into the output instead of storing (border-correction launch of the reflection dgrad)
    // LDS-DMA kernels only, one class, Ho % 8 == 0, Wo % 16 == 0: GEMM row m counts pixels in 128-pixel blocks of 8 rows x 16 columns
    // (m = block * 128 + local_row * 16 + local_col, blocks row-major over an image) instead of row-major over the image.  A 128-row
    // M-tile then touches 8 + R - 1 source rows of 16 columns for its R vertical taps instead of R x 128 distinct pixels: the tall
    // R x 1 GEMM of the width-Toeplitz forward (thin_toeplitz.hip) fetched every image row 9 times (6.0 GB against 0.63 GB of operands,
    // profiles/r04_pmc_kernels.json), here the re-reads of a tile are 8 of its own 16 rows.  Set by launch_igemm.
    int m2d;
    // XCD-aware tile order of igemm_pipe_kernel (filled by launch_pipe behind the constant xcd_env, measured without gain in profiles/r02_ab.txt): swz != 0 -> 1-D grid of
    // 8 * per * ntn * ncls workgroups; workgroup L runs on XCD L % 8 and takes M-tile (L % 8) * per + k of that XCD's
    // CONTIGUOUS eighth of the image, with (N-tile, class) fastest: every consumer of one pixel neighbourhood - the 9 taps of
    // adjacent rows, the N-tiles, the 4 phase classes of an up-conv - runs back to back on ONE XCD and finds it in that
    // XCD's L2.  Measured (profiles/r02_ab.txt, r02_conv_microbench.txt): no layer gains more than 2 %, the stride-2
    // parity-class dgrads lose 40 % (classes with 4/2/2/1 taps interleaved on one XCD), whole steps lose 2-3 % - the L2
    // misses of these kernels are served by the MALL and are not what limits them.  Default off.
    int swz, mtiles, ntn, per;
    int prio;            // launch_pipe's constant prio_env (measured without gain): s_setprio 1 while a wave is in its MFMA stream, 0 around the LDS fill
    int act;
    float slope;
    const float* oscale;  // optional [N][Co] multiplier applied after the activation (fused nn.Dropout2d mask)
    // optional ReLU-backward mask, same shape as the output: out = omask > 0 ? out : 0.  The input gradient of a conv whose INPUT is the
    // output of a fused conv+ReLU (the frozen VGG19 of srgan.py:61,112-113: conv, ReLU, conv ...) leaves through the ReLU's derivative
    // here, so the producing layer's backward has no separate act' pass.  LDS-DMA kernels only (launch_igemm refuses otherwise).
    const float* omask;
    // optional per-tile output statistics for the normalisation layer behind the conv (BatchNorm / InstanceNorm):
    // stats[((group * stats_chunks + chunk) * Co + col) * 3 + {0,1,2}] = (mean, M2, count) of this tile's rows of column col,
    // combined by migan_norm_stats_from_conv (Chan) - the norm layer's own statistics pass over the tensor disappears.
    // stats_inst = 0: one group (BatchNorm), chunk = cls * gridDim.x + tile;  1: group = image (InstanceNorm; Ho*Wo % BM == 0)
    float* stats;
    int stats_inst, stats_chunks;
    int oh0[MAX_CLS], ow0[MAX_CLS], Ho[MAX_CLS], Wo[MAX_CLS], tapbeg[MAX_CLS], ntap[MAX_CLS];
    // fastdiv magics per class for m / (Ho*Wo) and rem / Wo (filled by launch_igemm): the pixel decode of the pipelined
    // kernel's prologue and strided epilogue costs ~8 instead of ~80 VALU instructions per row
    unsigned mg_hw[MAX_CLS], mg_w[MAX_CLS];
    int sh_hw[MAX_CLS], sh_w[MAX_CLS];
    int wofs[MAX_TAPS];
    short dh[MAX_TAPS], dw[MAX_TAPS];  // source offset of a tap relative to the CLASS-LOCAL output index times istride
    // (dh << 16) | (dw & 0xffff), filled by launch_igemm for the LDS-DMA kernels: a dword table is read with scalar loads
    // (a 16-bit element of a kernel argument costs a vector global_load and a vmcnt wait that drains the DMA queue)
    int dhw[MAX_TAPS];
};

// Branch-free coordinate map of the gather: v = logical coordinate (output index * stride + tap offset), L = logical
// extent, Lphys = physical extent.  src is always a valid physical coordinate; the return value says whether the tap
// reads data (true) or the zero padding (false).
__device__ __forceinline__ bool map_bf(int v, int L, int Lphys, int mode, int& src) {
    int r = v < 0 ? -v : v;
    r = r >= L ? 2 * L - 2 - r : r;
    bool inr = (unsigned)v < (unsigned)L;
    int s = mode == GATHER_REFLECT ? r : (mode == GATHER_UP2 ? (v >> 1) : v);
    s = s < 0 ? 0 : s;
    s = s > Lphys - 1 ? Lphys - 1 : s;
    src = s;
    return mode == GATHER_REFLECT ? true : inr;
}

// Geometry of one weight-gradient launch: dW[co][t][ci] = sum_p dy[p][co] * gather(x)[p][t][ci]; GEMM M = Co,
// N = taps*Ci, K = pixels, split-K over pixel ranges into per-split slabs (fixed-order reduction afterwards).
struct WgradGeom {
    int N, Hi, Wi, Ci, HiL, WiL;
    int Ho, Wo, Co;
    int R, S, stride, pad_t, pad_l, gather;
    int splits, pix_per_split;  // pixels per split (multiple of 32)
    int tiles_m, tiles_n;       // tile grid of the pipelined kernel (1-D XCD-aware launch)
    unsigned mg_hw, mg_w;       // magic multipliers / shifts for p / (Ho*Wo) and rem / Wo (pipelined kernel)
    int sh_hw, sh_w;
    // dy may be a strided sub-grid of a larger gradient tensor (phase classes of the collapsed Upsample+Conv):
    // pixel (n, oi, oj) of this GEMM lives at dy[n][dy_oh0 + oi*dy_step][dy_ow0 + oj*dy_step]
    int dy_H, dy_W, dy_oh0, dy_ow0, dy_step;
    // optional fused bias gradient: the blocks of column-tile 0 also sum their dy tiles over pixels (the A operand is
    // already in LDS) into bpart[cls*splits + split][Co]; the reduction launch adds the slabs.  NULL: not requested.
    // Measured on MI355X: no faster than the separate column-sum launches (the column-0 blocks become the critical
    // path of a one-wave launch; spreading the rows over all column tiles costs every block more than it saves), so
    // the host mirror leaves it off (functional._FUSE_BIAS).
    float* bpart;
};
