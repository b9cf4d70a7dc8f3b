/*
 * migan.h — C ABI of libmigan.so, the MI355X (gfx950) kernels behind the GAN training hot path.
 *
 * Boundary: the reference has no FFI; its "plugin interface" for this path is the
 * torch.nn layer set that implementations/{dcgan,wgan_gp,cyclegan,pix2pix,srgan} construct, whose
 * forward/backward PyTorch dispatches to ATen.  Each entry point below replaces one ATen op family on
 * that path; the reference call site it serves is cited next to it (paths relative to the reference
 * repo root).  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only: every pointer is a DEVICE pointer owned by the caller (torch's
 *     caching allocator in our host code); the library never allocates, frees or synchronises, so every
 *     launcher is legal inside hipGraph capture.  `stream` is a hipStream_t.
 *   - all tensors are fp32.  4-D activations are NHWC ([N][H][W][C], torch.channels_last memory).
 *   - return value: 0 on success, otherwise a hipError_t code (migan_error_string()).
 *   - activation codes: 0 none, 1 LeakyReLU(slope), 2 ReLU, 3 Tanh, 4 Sigmoid.
 *   - gather modes of the conv loaders: 0 zero padding, 1 reflection padding (nn.ReflectionPad2d folded
 *     into the conv), 2 nearest-upsample x2 then zero padding (nn.Upsample(scale_factor=2) folded in).
 */
#ifndef MIGAN_H
#define MIGAN_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* migan_version(void);
const char* migan_error_string(int code);
/* Debug launch counters (test tooling, no reference counterpart): launches issued by this library since the last reset, summed
 * over the launch sites whose kernel
 * expression contains `substr` (NULL or "" = all).  The parity tests assert with it WHICH kernel family served a geometry
 * (torch's counterpart is the profiler's kernel list; the reference itself never looks).  Host-side bookkeeping only. */
long migan_debug_launch_count(const char* substr);
void migan_debug_launch_reset(void);

/* ---- Convolution family: implicit GEMM on v_mfma_f32_32x32x2_f32 (csrc/conv_igemm.hip) ----------
 * nn.Conv2d forward: dcgan.py:55,59,62,78  cyclegan/models.py:28,32,50,60,75,82,106,118
 *                    pix2pix/models.py:23,79,115,127  srgan/models.py:22,25,38,47,54,62,85,89,100
 * nn.Linear forward (N=batch, H=W=1, R=S=1): dcgan.py:50,92  wgan_gp.py:46,56,73-77  gan.py:41-59
 * x [N][Hi][Wi][Ci]; w_ohwi [Co][R][S][Ci]; bias [Co] or NULL; y [N][Ho][Wo][Co].
 * pad_t/pad_l are the top/left zero (or reflection) pads in logical (upsampled, for gather 2)
 * coordinates; bottom/right padding is implied by Ho/Wo.  Epilogue: y = act(conv + bias). */
int migan_conv2d_fwd(const float* x, const float* w_ohwi, const float* bias, float* y, int N, int Hi, int Wi,
                     int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t, int pad_l, int gather,
                     int act, float slope, void* stream);
/* Conv2d -> activation -> nn.Dropout2d(p) (the discriminator block of dcgan.py:77-78) in one launch:
 * y = act(conv(x)+bias) * mask[n][co], mask from migan_rand_mask.  Co % 4 == 0.  Backward of the act+mask pair:
 * migan_act_bwd_nc (dx = dy * mask[n][c] * act'(y), y = the masked output). */
int migan_conv2d_dropout_fwd(const float* x, const float* w_ohwi, const float* bias, const float* mask_nc, float* y,
                             int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t,
                             int pad_l, int gather, int act, float slope, void* stream);
int migan_act_bwd_nc(const float* dy, const float* y, const float* mask_nc, float* dx, int N, int HW, int C, int act,
                     float slope, void* stream);

/* The same launches with the statistics of the normalisation layer BEHIND the conv taken in the conv's epilogue
 * (dcgan.py:55-56,78-80, cyclegan/models.py:28-29, srgan/models.py:22-23: Conv2d -> [act -> Dropout2d ->] BatchNorm /
 * InstanceNorm): every output tile leaves (mean, M2, count) per column in `stats` ([groups][chunks][Co][3] floats,
 * groups = N for instance != 0 else 1, chunks from the *_stats_chunks query; 0 = geometry not on the pipelined MFMA
 * kernel, use the plain launch), and migan_norm_stats_from_conv combines them (Chan, in double) into mean / invstd +
 * running statistics - the norm layer's own pass over the conv output disappears.  mask_nc may be NULL. */
int migan_conv2d_stats_chunks(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t,
                              int pad_l, int gather, int instance);
int migan_conv2d_fwd_stats(const float* x, const float* w_ohwi, const float* bias, const float* mask_nc, float* y, int N,
                           int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t, int pad_l,
                           int gather, int act, float slope, float* stats, int stats_chunks, int instance, void* stream);
int migan_upconv3x3_stats_chunks(int N, int H, int W, int Ci, int Co, int instance);
int migan_upconv3x3_fwd_stats(const float* x, const float* wf, const float* bias, float* y, int N, int H, int W, int Ci,
                              int Co, int act, float slope, float* stats, int stats_chunks, int instance, void* stream);
int migan_norm_stats_from_conv(const float* part, int nchunks, float* mean, float* invstd, float* running_mean,
                               float* running_var, long long* num_batches_tracked, float momentum, float eps, int G, int C,
                               void* stream);

/* Skinny GEMMs (csrc/skinny_mm.hip): nn.Linear forward / input gradient at <= 64 rows - the MLP critic and generator
 * of wgan_gp.py:42-83 / gan.py:38-81 at the reference batch size.  v_mfma_f32_16x16x4_f32 fed straight from 16-byte
 * global loads (no LDS, no barrier), N/16 (N/32) workgroups; the NN form reads w in its stored [N][K] layout, so the
 * input gradient needs no transposed weight copy.  *_ok() = 1 when the shape qualifies (M <= 64, N % 16 == 0,
 * K % 16 == 0, K >= 32  /  R % 16 == 0, Nc % 32 == 0); otherwise use migan_conv2d_fwd with 1x1 geometry. */
int migan_skinny_nt_ok(int M, int N, int K);
int migan_skinny_nn_ok(int M, int R, int Nc);
int migan_skinny_nt(const float* a, const float* w, const float* bias, float* c, int M, int N, int K, int act, float slope,
                    void* stream);
int migan_skinny_nn(const float* a, const float* w, float* c, int M, int R, int Nc, void* stream);
/* nn.Linear weight + bias gradient at <= 64 rows (the Linear layers of wgan_gp.py:46-78 under d_loss.backward() /
 * g_loss.backward(), wgan_gp.py:173,192): dw[N][K] (+)= dy[M][N]^T x[M][K], db[N] (+)= column sums of dy (db may
 * be NULL) in ONE launch straight into the caller's buffers (no split-K slabs, no reduction, no column-sum launches).
 * N % 16 == 0, K % 64 == 0. */
int migan_skinny_tn_ok(int M, int N, int K);
int migan_skinny_tn(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, int accumulate,
                    int db_accumulate, void* stream);

/* Few-pixel convolutions (csrc/fewpix.hip): nn.Conv2d / nn.ConvTranspose2d with <= 64 pixel rows and >= 1 M weights - the inner
 * U-Net levels of pix2pix/models.py:62-71 at the reference's batch 1 (512->512 and 1024->512, 4x4 stride 2, at 8x8 ... 1x1).
 * torch runs them as ATen convolution / convolution_backward; here every product is one of the skinny GEMMs above on the weight in
 * its STORED layout (Conv2d [Co][Ci*R*S], ConvTranspose2d [Ci][Co*R*S]) around these two index kernels - no weight packs, no
 * split-K slabs, no reduction launches.  migan_fewpix_ok: rows = N*Ho*Wo (Conv2d) or N*Hin*Win (ConvTranspose2d), n / k = the
 * weight's leading dimension / the product of its trailing ones.
 * migan_im2col_small: col[N*Ho*Wo][C*R*S] (column (c, r, s)) of x[N][H][W][C] (NHWC) for taps at h = ho*stride - pt + r.
 * migan_col2im_small: the adjoint, out[N][H][W][J] = act(bias + sum of ycol[(n,ho,wo)][(j,r,s)] over the taps that land on
 * (h, w)) in a fixed order; bias may be NULL. */
int migan_fewpix_ok(int rows, int n, int k);
/* The NT product of that path (pix2pix/models.py:62-71 at batch 1) with K also split over workgroups (a 16.8-33.5 MB weight against
 * <= 64 rows wants 256+ workgroups,
 * not N/16): out[M][N] = act(a[M][K] w[N][K]^T + bias), partial tiles in ws (migan_fewpix_nt_workspace bytes; 0 = no split, the
 * call is migan_skinny_nt), added in a fixed order by a second launch.  Conv2d forward / ConvTranspose2d input gradient. */
size_t migan_fewpix_nt_workspace(int M, int N, int K);
int migan_fewpix_nt(const float* a, const float* w, const float* bias, float* out, float* ws, size_t ws_bytes, int M, int N, int K,
                    int act, float slope, void* stream);
int migan_im2col_small(const float* x, float* col, int N, int H, int W, int C, int Ho, int Wo, int R, int S, int stride, int pt,
                       int pl, void* stream);
int migan_col2im_small(const float* ycol, const float* bias, float* out, int N, int H, int W, int J, int Ho, int Wo, int R, int S,
                       int stride, int pt, int pl, int act, float slope, void* stream);

/* K7 (csrc/critic_fused.hip): one WGAN-GP critic iteration of the MLP critic as six dependent launches - replaces, for
 * wgan_gp.py:68-83 (Discriminator), :119-138 (compute_gradient_penalty, incl. autograd.grad(create_graph=True)) and :160-176
 * (real_validity, fake_validity, d_loss, d_loss.backward()), the ~75 launches of the op-by-op path: D(real), D(fake), the
 * gradient penalty of the interpolates alpha*real + (1-alpha)*fake, d_loss = -mean D(real) + mean D(fake) + lambda*gp, and the
 * gradient of d_loss w.r.t. w1 [H1][Din], b1, w2 [H2][H1], b2, w3 [H2], b3 written into gw1..gb3 (accumulate != 0: added, as
 * optimizer.zero_grad() + backward() would leave them).  real, fake [B][Din]; alpha [B]; out[4] = d_loss, gp, mean D(real),
 * mean D(fake).  B <= 64; Din, H1, H2 % 128 == 0.  ws: migan_critic_fused_workspace() bytes of scratch.  phase: 0 = all six
 * launches; 1..6 = that launch alone (timing harness). */
int migan_critic_fused_ok(int B, int Din, int H1, int H2);
size_t migan_critic_fused_workspace(int B, int Din, int H1, int H2);
int migan_critic_fused(const float* real, const float* fake, const float* alpha, const float* w1, const float* b1,
                       const float* w2, const float* b2, const float* w3, const float* b3, float* gw1, float* gb1, float* gw2,
                       float* gb2, float* gw3, float* gb3, float* out, float* ws, size_t ws_bytes, int B, int Din, int H1,
                       int H2, float slope, float lambda, int accumulate, int phase, void* stream);

/* Forward of an MLP generator at <= 64 rows, one launch per layer (csrc/mlp_fused.hip): replaces, for the no_grad
 * `fake_imgs = generator(z)` of wgan_gp.py:163 (Generator wgan_gp.py:42-65; gan.py:38-61), 5 x Linear + 3 x BatchNorm1d(out, 0.8)
 * (training mode: batch statistics, running statistics and num_batches_tracked updated) + LeakyReLU / Tanh = 14 launches.
 * Host arrays: dims[4*l] = {K, N, has_bn, act code}, fpar[3*l] = {slope, eps, momentum}, ptrs[7*l] = DEVICE pointers {W [N][K], b,
 * gamma, beta, running_mean, running_var, num_batches_tracked (int64)}, NULL where absent.  B <= 64, K % 4 == 0, N % 16 == 0,
 * <= 8 layers.  save = 0: ws holds two ping-pong activation buffers.  tickets: 1024 unsigned ints zeroed ONCE by the caller (left zero
 * by every launch; the row-group workgroups of a BatchNorm1d column tile meet there).  only: 0 = every layer; 1 + l = layer l alone
 * (timing harness). */
int migan_mlp_fused_ok(int B, int nlayers, const int* dims);
size_t migan_mlp_fused_workspace(int B, int nlayers, const int* dims, int save);
int migan_mlp_fused_fwd(const float* x, float* y, int B, int nlayers, const int* dims, const float* fpar, void* const* ptrs,
                        float* ws, size_t ws_bytes, int save, unsigned* tickets, int only, void* stream);
/* ... and its backward, one launch per phase (the generator iteration wgan_gp.py:179-193: both `generator(z)` and the frozen
 * `discriminator(fake_imgs)` are such MLPs): forward with save = 1 (ws then holds every layer's output, and the normalised values
 * and 1/std of the BatchNorm layers), then dy [B][N_last] -> parameter gradients written into (accumulate != 0: added to)
 * gptrs[4*l] = {dW [N][K], db [N], dgamma, dbeta} (device pointers in a host array; NULL = not wanted) and, when dx != NULL, the
 * input gradient dx [B][K_0] (K_0 % 32 == 0).  N % 32 == 0 for every layer but the last (a critic's single output column is fine).
 * only: 0 = every phase; 1 + ph = phase ph alone (timing harness). */
size_t migan_mlp_fused_bwd_workspace(int B, int nlayers, const int* dims);
int migan_mlp_fused_bwd(const float* x, const float* y, const float* dy, const float* save, float* dx, int B, int nlayers,
                        const int* dims, const float* fpar, void* const* ptrs, void* const* gptrs, float* ws, size_t ws_bytes,
                        int accumulate, int only, void* stream);

/* Conv2d input gradient (aten::convolution_backward, grad_input) == nn.ConvTranspose2d forward
 * (pix2pix/models.py:39, k=4 s=2 p=1).  Geometry arguments describe the FORWARD conv; dy [N][Ho][Wo][Co];
 * w_ihwo [Ci][R][S][Co]; dx [N][Hi][Wi][Ci] = act(sum + bias) (bias/act used by the ConvTranspose role).
 * stride 1 or 2 (dense per-parity tap lists, no zero-multiplies). */
int migan_conv2d_dgrad(const float* dy, const float* w_ihwo, const float* bias, float* dx, int N, int Hi, int Wi,
                       int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t, int pad_l, int act,
                       float slope, void* stream);

/* Split-K variants for under-filled GEMMs (csrc/conv_dma.hip): a few output pixels against a long K - the inner U-Net
 * levels of pix2pix/models.py:62-78 (1-64 pixels, 16.8-33.5 MB of weights per layer), the PatchGAN heads
 * cyclegan/models.py:106-118, the discriminator of dcgan.py:77-92.  K is cut over up to 512 workgroups; every slice leaves
 * its 64x64 partial tile in a slab of `ws`, and the last slice to arrive at the tile's ticket adds the slabs in slice order
 * (deterministic) and applies bias / activation / mask.  `ws` = migan_conv_splitk_workspace() bytes, zeroed ONCE by the caller
 * (tickets return to zero after every launch), not shared by launches that may overlap (one per stream); NULL = no split.
 * mask_nc: optional [N][Co] multiplier (the fused nn.Dropout2d of migan_conv2d_dropout_fwd) or NULL. */
size_t migan_conv_splitk_workspace(void);
int migan_conv_splitk_applies(long long maxM, int Co, int Ci_src, int ncls);  /* 1: a workspace would be used (same layers) */
int migan_conv2d_fwd_ws(const float* x, const float* w_ohwi, const float* bias, const float* mask_nc, float* y, int N,
                        int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t, int pad_l,
                        int gather, int act, float slope, float* ws, size_t ws_bytes, void* stream);
int migan_conv2d_dgrad_ws(const float* dy, const float* w_ihwo, const float* bias, float* dx, int N, int Hi, int Wi,
                          int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t, int pad_l, int act,
                          float slope, float* ws, size_t ws_bytes, void* stream);
/* Input gradient of a conv whose INPUT is the output of a fused conv+ReLU (or of the MaxPool2d behind one): the frozen
 * vgg19.features[:18] of srgan/models.py:8-15 as srgan.py:112-113 differentiates it (loss_content.backward through conv, ReLU,
 * conv ...).  dx[i] = relu_out[i] > 0 ? dgrad(dy)[i] : 0 in the epilogue of the input-gradient launch, so the ReLU backward of the
 * producing layer costs no pass over the tensor (ATen: threshold_backward).  relu_out = the conv's saved input [N][Hi][Wi][Ci].
 * Returns hipErrorNotSupported (801) for geometries the LDS-DMA kernels do not serve: run migan_conv2d_dgrad_ws + migan_act_bwd. */
int migan_conv2d_dgrad_relu_ws(const float* dy, const float* w_ihwo, float* dx, const float* relu_out, int N, int Hi, int Wi,
                               int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t, int pad_l, float* ws,
                               size_t ws_bytes, void* stream);

/* Introspection, no reference counterpart (the reference cannot say which ATen kernel served a layer): which tile
 * configuration migan_conv2d_fwd/dgrad will launch for a GEMM of maxM rows (largest parity class),
 * Co columns and a source with Ci_src channels: fast*1000000 + BM*1000 + BN ("fast" = vectorised NHWC loader,
 * Ci_src % 4 == 0 and >= 8); 4000 = the thin-N VALU kernel (Co <= 4).  Pure function; used by bench.py to attribute
 * launches to kernel symbols. */
int migan_igemm_tile_code(long long maxM, int Co, int Ci_src, int ncls);

/* Conv2d / Linear weight gradient (aten::convolution_backward grad_weight; aten::mm in AddmmBackward): what every
 * loss.backward() of the path runs per conv / linear layer - dcgan.py:168,182, wgan_gp.py:173,192, cyclegan.py:204,221,238,
 * pix2pix.py:150,171, srgan.py:128,144.
 * Split-K over pixels + fixed-order reduction (deterministic).  dw_oihw has the torch parameter layout
 * [Co][Ci][R][S].  ws must hold migan_conv2d_wgrad_workspace() bytes.  accumulate != 0: dw += gradient (the
 * final reduction adds into the caller's buffer - what autograd's AccumulateGrad does with a separate
 * aten::add launch per parameter; used to write straight into the optimiser's flat gradient bucket).  The same
 * flag exists on migan_upconv3x3_wgrad, migan_colsum (bias gradients) and migan_norm_bwd (dgamma/dbeta). */
size_t migan_conv2d_wgrad_workspace(int N, int Ho, int Wo, int Co, int R, int S, int Ci);
int migan_conv2d_wgrad(const float* x, const float* dy, float* dw_oihw, float* ws, size_t ws_bytes, int N, int Hi,
                       int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t, int pad_l,
                       int gather, int accumulate, float* db, int db_accumulate, const float* db_slabs, int db_nslab,
                       void* stream);
/* db != NULL: the bias gradient db[Co] = sum over pixels of dy (aten::convolution_backward grad_bias / the sum in
 * AddmmBackward) comes out of the wgrad's fixed-order reduction launch instead of a separate two-launch column sum
 * that re-reads dy from HBM:
 *   db_slabs != NULL: db_nslab x [Co] per-block column sums of dy, written by the streaming kernel that PRODUCED dy
 *     (migan_norm_bwd / migan_norm_bwd_apply / migan_act_bwd_colsum, csum argument); any wgrad path accepts them.
 *   db_slabs == NULL: the column-tile-0 workgroups of the MFMA wgrad sum their dy tiles (already in LDS as the A
 *     operand) - only where migan_conv2d_wgrad_fuses_bias() returns 1; measured no faster than migan_colsum on MI355X
 *     (those workgroups become the tail of the launch), so the host mirror leaves it off (functional._FUSE_BIAS). */
int migan_conv2d_wgrad_fuses_bias(int Co, int R, int S, int Ci, int stride, int gather);
/* Input gradient of nn.ReflectionPad2d(1) -> nn.Conv2d(Ci, Co, 3) (cyclegan/models.py:26-35) straight into
 * dx [N][H][W][Ci] from dy [N][H][W][Co], w_ihwo [Ci][3][3][Co]: the ordinary pad-1 dgrad plus a second small launch
 * that ADDS the terms of the reflected ring onto rows/columns 1 and H-2 (no (H+2)x(W+2) intermediate, no fold pass, and
 * the GEMM keeps M = N*H*W rows).  Needs H, W >= 4, Co % 4 == 0, Co >= 8, Ci > 4; otherwise returns an error and the
 * caller runs migan_conv2d_dgrad on the padded extent + migan_gather2d_bwd. */
int migan_conv2d_dgrad_reflect1(const float* dy, const float* w_ihwo, float* dx, int N, int H, int W, int Ci, int Co,
                                void* stream);
/* Its second launch alone (cyclegan/models.py:26-35 under cyclegan.py:204; dx must already hold the pad-1 input gradient of
 * migan_conv2d_dgrad): a latency-bound launch of
 * mostly tiny workgroups that a caller may overlap with independent work on another stream. */
int migan_conv2d_dgrad_reflect1_ring(const float* dy, const float* w_ihwo, float* dx, int N, int H, int W, int Ci, int Co,
                                     void* stream);
/* migan_conv2d_dgrad_reflect1 with a split-K workspace (migan_conv_splitk_workspace() bytes, zero at rest, one per stream; NULL:
 * none) for both launches: at one image per GPU (cyclegan.py:28) the pad-1 launch is one 64x64 tile per CU and the ring launch a
 * chain of 56 dependent K-tiles - both are cut along K where migan_conv_splitk_applies() says so. */
int migan_conv2d_dgrad_reflect1_ws(const float* dy, const float* w_ihwo, float* dx, int N, int H, int W, int Ci, int Co,
                                   float* ws, size_t ws_bytes, void* stream);
/* ... and its ring launch alone (cyclegan/models.py:26-35 backward) with a workspace (the workspace of the stream it is launched on). */
int migan_conv2d_dgrad_reflect1_ring_ws(const float* dy, const float* w_ihwo, float* dx, int N, int H, int W, int Ci, int Co,
                                        float* ws, size_t ws_bytes, void* stream);

/* Image-input convolutions (3 source channels, stride 1, square 3 / 7 kernel, 32 or 64 output channels, zero or reflection
 * padding: srgan/models.py:85 Conv2d(3,64,3,1,1), vgg19.features[0] behind srgan/models.py:11, cyclegan/models.py:49-50
 * ReflectionPad2d(3)+Conv2d(3,64,7)) on the MFMA units straight from staged image rows
 * (csrc/rgb_conv.hip): K = R*S*3 is the GEMM's reduction as it is - no 32-channel tap tiles, no im2col buffer.
 *   fwd:   x [N][H][W][Ci] (Ci = 3; 1 with a 3x3 kernel), w_hwio [R][S][Ci][Co] (the OIHW weight permuted (2,3,1,0)),
 *          y [N][Ho][Wo][Co] = act(conv + bias), act = none / LeakyReLU / ReLU.  flip != 0: taps read in reverse order - with x = dy and w_hwio = a thin-output
 *          layer's weight [c][Co][R][S] permuted (2,3,0,1) this is that layer's input gradient (dcgan.py:62 backward)
 *   wgrad: dw_oihw [Co][3][R][S] and (db != NULL) db [Co] of y = act(conv(x, w) + b) from dy and the layer's OUTPUT y_act:
 *          the activation backward g = dy * act'(y_act) and the bias column sums are part of the launch (dy, y_act read once;
 *          y_act == NULL with act = 0: dy is the gradient of the pre-activation).  accumulate_w / accumulate_b != 0: +=.
 *          ws >= migan_rgb_conv_wgrad_workspace(Co, R, S) bytes; fixed-order reduction (deterministic).
 * migan_rgb_conv_ok / migan_rgb_conv_wgrad_ok: 1 when the entry takes the geometry (pixels = N*Ho*Wo). */
int migan_rgb_conv_ok(int Ci, int Co, int R, int S, int stride, int gather, long long pixels);
int migan_rgb_conv_fwd(const float* x, const float* w_hwio, const float* bias, float* y, int N, int H, int W, int Ci, int Ho, int Wo,
                       int Co, int R, int S, int pad_t, int pad_l, int gather, int act, float slope, int flip, void* stream);
int migan_rgb_conv_wgrad_ok(int Ci, int Co, int R, int S, int stride, int gather, long long pixels);
size_t migan_rgb_conv_wgrad_workspace(int Co, int R, int S);
int migan_rgb_conv_wgrad(const float* x, const float* dy, const float* y_act, float* dw_oihw, float* db, float* ws, size_t ws_bytes,
                         int N, int H, int W, int Ho, int Wo, int Co, int R, int S, int pad_t, int pad_l, int gather, int act,
                         float slope, int accumulate_w, int accumulate_b, void* stream);

/* Thin-N convolutions (Co <= 4 output channels, stride 1: cyclegan/models.py:82 ReflectionPad2d(3)+Conv2d(64,3,7),
 * srgan/models.py:62 Conv2d(64,3,9,1,4); needs 16 <= S*Co <= 32, Ci % 4 == 0, Ci >= 16) on the MFMA kernels through a width-Toeplitz
 * expansion: the kernel column s moves into the GEMM N dimension (Co' = S*Co rounded up to 4 columns, 84 % of a 32-wide
 * tile for 9x9 instead of 9 %), P[n][h][u][(s,co)] = R x 1 convolution of x, y = act(bias + sum_s P[..][map(w+s-pad_l)][(s,co)]).
 * Backward: q = transposed expansion of dy; dw = wgrad of the R x 1 conv + fold; dx = dgrad of the R x 1 conv.
 *   pack:  w_oihw [Co][Ci][R][S] -> wt [Co'][R][Ci] and wd [Ci][R][Co'] (Co' = migan_thin_toeplitz_cols)
 *   fwd:   ws >= migan_thin_toeplitz_workspace(N, Ho, Wi, Co, S) bytes (the P buffer; q has the same size)
 *   wgrad: ws >= migan_thin_toeplitz_wgrad_workspace(); accumulate != 0: dw += gradient
 *   dgrad: ws >= migan_thin_toeplitz_dgrad_workspace() (row-padded intermediate, reflection padding only)
 * The bias gradient is the column sum of dy (migan_colsum).  gather: 0 zero padding, 1 reflection padding. */
int migan_thin_toeplitz_ok(int Co, int R, int S, int Ci, int stride, int gather);
int migan_thin_toeplitz_cols(int Co, int S);
size_t migan_thin_toeplitz_workspace(int N, int Ho, int Wi, int Co, int S);
int migan_thin_toeplitz_pack(const float* w_oihw, float* wt, float* wd, int Co, int Ci, int R, int S, void* stream);
int migan_thin_toeplitz_fwd(const float* x, const float* wt, const float* bias, float* y, float* ws, size_t ws_bytes, int N,
                            int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int pad_t, int pad_l, int gather,
                            int act, float slope, void* stream);
int migan_thin_toeplitz_expand(const float* dy, float* q, int N, int Ho, int Wo, int Co, int Wi, int S, int pad_l, int gather,
                               void* stream);
size_t migan_thin_toeplitz_wgrad_workspace(int N, int Ho, int Wi, int Ci, int Co, int R, int S);
int migan_thin_toeplitz_wgrad(const float* x, const float* q, float* dw_oihw, float* ws, size_t ws_bytes, int N, int Hi, int Wi,
                              int Ci, int Ho, int Co, int R, int S, int pad_t, int gather, int accumulate, void* stream);
size_t migan_thin_toeplitz_dgrad_workspace(int N, int Hi, int Wi, int Ci, int Ho, int R, int gather);
int migan_thin_toeplitz_dgrad(const float* q, const float* wd, float* dx, float* ws, size_t ws_bytes, int N, int Hi, int Wi,
                              int Ci, int Ho, int Co, int R, int S, int pad_t, int gather, void* stream);

/* Phase-collapsed nn.Upsample(scale_factor=2) -> nn.Conv2d(Ci, Co, 3, stride=1, padding=1)
 * (dcgan.py:54-55,58-59; cyclegan/models.py:74-75): the 4 output phases are 2x2 convs of the un-upsampled input with
 * pre-summed weights -> 16/36 of the dense FLOPs in forward, dgrad and wgrad, no upsampled intermediate.
 * migan_upconv3x3_pack: w_oihw [Co][Ci][3][3] -> wf [Co][16][Ci] (forward) and wd [Ci][16][Co] (dgrad).
 * x [N][H][W][Ci], y / dy [N][2H][2W][Co].  wgrad needs Co % 4 == 0 and Ci % 4 == 0 (else use
 * migan_conv2d_wgrad with gather mode 2, which computes the same gradient densely). */
int migan_upconv3x3_pack(const float* w_oihw, float* wf, float* wd, int Co, int Ci, void* stream);
int migan_upconv3x3_fwd(const float* x, const float* wf, const float* bias, float* y, int N, int H, int W, int Ci,
                        int Co, int act, float slope, void* stream);
int migan_upconv3x3_dgrad(const float* dy, const float* wd, float* dx, int N, int H, int W, int Ci, int Co,
                          void* stream);
size_t migan_upconv3x3_wgrad_workspace(int N, int H, int W, int Co, int Ci);
int migan_upconv3x3_wgrad(const float* x, const float* dy, float* dw_oihw, float* ws, size_t ws_bytes, int N, int H,
                          int W, int Ci, int Co, int accumulate, float* db, int db_accumulate, const float* db_slabs,
                          int db_nslab, void* stream);

/* Small instance-normalised tensors (pix2pix/models.py:25,41 inner U-Net levels and PatchGAN at batch 1, cyclegan/models.py:29
 * at one image per GPU): nn.InstanceNorm2d forward - statistics, normalisation, fused activation, residual add - in ONE launch
 * instead of three; mean / invstd [G][C] are written for the backward (migan_norm_bwd takes the same one-launch route for these
 * shapes).  migan_norm_small_ok: C % 16 == 0, 2 <= P <= 1024, G * C / 16 >= 8, G * P * C <= 2^20. */
int migan_norm_small_ok(int G, int P, int C);
int migan_norm_fwd_small(const float* x, float* y, float* mean, float* invstd, const float* gamma, const float* beta,
                         const float* res, const float* mask, int G, int P, int C, int act, float slope, float eps, void* stream);
/* mask (may be NULL): the nn.Dropout behind the activation of pix2pix/models.py:25-28,41-45 (InstanceNorm2d -> LeakyReLU / ReLU ->
 * Dropout(0.5)) as a [G][P][C] multiplier applied in the same launch; migan_norm_bwd_small is the matching backward
 * (dz = dy * mask * act'(z); csum as for migan_norm_bwd, may be NULL). */
int migan_norm_bwd_small(const float* x, const float* dy, const float* mask, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, float* dx, int G, int P, int C, int act, float slope, float* csum,
                         void* stream);

/* ---- BatchNorm2d/1d (train) and InstanceNorm2d (csrc/norm.hip) ---------------------------------------
 * nn.BatchNorm2d(C[,eps]): dcgan.py:53,56,60,80  srgan/models.py:23,26,47,55,87,90;  nn.BatchNorm1d:
 * wgan_gp.py:49, gan.py:45;  nn.InstanceNorm2d(C): cyclegan/models.py:29,33,51,62,77,108
 * pix2pix/models.py:25,40,117.   Data viewed as [G][P][C]: BatchNorm G=1,P=N*H*W; InstanceNorm G=N,P=H*W.
 * migan_norm_stats: biased variance -> invstd = 1/sqrt(var+eps); running stats (G==1, non-NULL) updated
 * with momentum and the unbiased variance, as torch does; num_batches_tracked (int64 scalar, may be NULL) += 1. */
size_t migan_norm_workspace(int G, int P, int C);
int migan_norm_stats(const float* x, float* mean, float* invstd, float* running_mean, float* running_var,
                     long long* num_batches_tracked, float momentum, float eps, int G, int P, int C, float* ws,
                     size_t ws_bytes, void* stream);
/* y = act((x-mean)*invstd*gamma+beta) [+ res]; gamma/beta/res may be NULL.  The normalise + activation (+ residual add) of
 * dcgan.py:53-61, cyclegan/models.py:28-37, srgan/models.py:22-30 in one pass. */
int migan_norm_apply(const float* x, float* y, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, const float* res, int G, int P, int C, int act, float slope,
                     void* stream);
/* nn.BatchNorm2d -> nn.PReLU() (srgan/models.py:23-24,55-57; single learnable slope prelu_weight[0] on the device) fused:
 * forward in the apply launch; backward as migan_norm_bwd plus dprelu[0] (+)= sum dy*min(z,0) taken as a third sum of the
 * statistics pass - the PReLU layer costs no pass of its own over the tensor.  ws: migan_norm_workspace_prelu() bytes.
 * shuffle_H, shuffle_W > 0 (0 = off): nn.PixelShuffle(2) between the two (srgan/models.py:55-57) as the STORE index map of the
 * apply launch and the LOAD index map of dy in the backward launches: x is [N][H][W][C] (G = 1, P = N*H*W, C % 4 == 0, no
 * residual), y and dy are [N][2H][2W][C/4]; prelu_weight may be NULL then (shuffle without PReLU). */
int migan_norm_apply_prelu(const float* x, float* y, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, const float* res, const float* prelu_weight, int G, int P, int C, int shuffle_H,
                           int shuffle_W, void* stream);
size_t migan_norm_workspace_prelu(int G, int P, int C);
int migan_norm_bwd_prelu(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, const float* prelu_weight, float* dx, float* dgamma, float* dbeta, float* dprelu,
                         int G, int P, int C, float* ws, size_t ws_bytes, int accumulate, int dprelu_accumulate, float* csum,
                         int shuffle_H, int shuffle_W, void* stream);
/* backward through the batch statistics and the fused activation (native_batch_norm_backward + the activation's backward behind
 * dcgan.py:168,182, cyclegan.py:204, srgan.py:128,144); dgamma/dbeta [C] written when G==1.
 * csum (optional): migan_norm_colsum_slabs(G,P,C) x [C] per-block column sums of dx - the bias gradient of the conv in
 * front of the norm layer is reduced from them inside that conv's wgrad launch (migan_conv2d_wgrad db_slabs). */
int migan_norm_colsum_slabs(int G, int P, int C);
int migan_norm_bwd(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, float* dx, float* dgamma, float* dbeta, int G, int P, int C, int act,
                   float slope, float* ws, size_t ws_bytes, int accumulate, float* csum, void* stream);
/* The two halves of migan_norm_bwd, for cross-replica BatchNorm (data parallel): sums [G][C][2] =
 * (sum dyz, sum dyz*xhat) over this rank's pixels are all-reduced (SUM) between them and P_total = world * P. */
int migan_norm_bwd_sums(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, float* sums, float* dgamma, float* dbeta, int G, int P, int C, int act,
                        float slope, float* ws, size_t ws_bytes, int accumulate, void* stream);
int migan_norm_bwd_apply(const float* x, const float* dy, float* dx, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, const float* sums, int G, int P, int C, int act,
                         float slope, long long P_total, float* csum, void* stream);
/* ... and of migan_norm_bwd_prelu (BatchNorm2d [-> PixelShuffle(2)] -> PReLU, srgan/models.py:23-24,55-57, under cross-replica
 * statistics - srgan.py:97-145 at 2 images per rank): dprelu (+)= this rank's part of the slope gradient; ws for the first half:
 * migan_norm_workspace_prelu() bytes. */
int migan_norm_bwd_sums_prelu(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, const float* prelu_weight, float* sums, float* dgamma, float* dbeta,
                              float* dprelu, int G, int P, int C, float* ws, size_t ws_bytes, int accumulate,
                              int dprelu_accumulate, int shuffle_H, int shuffle_W, void* stream);
int migan_norm_bwd_apply_prelu(const float* x, const float* dy, float* dx, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, const float* prelu_weight, const float* sums, int G,
                               int P, int C, long long P_total, float* csum, int shuffle_H, int shuffle_W, void* stream);
/* Cross-replica BatchNorm forward: migan_norm_moments = this rank's mean and BIASED variance [C] (no eps, no running
 * statistics); after an all_gather into [world][2][C], migan_norm_sync_finalize combines the equal shards (Chan, in
 * double) into the global-batch mean / invstd and updates the running statistics with the global unbiased variance -
 * the statistics the single-process reference computes on the whole batch (dcgan.py:53-60, srgan/models.py:23-26). */
int migan_norm_moments(const float* x, float* mean, float* var, int G, int P, int C, float* ws, size_t ws_bytes,
                       void* stream);
int migan_norm_sync_finalize(const float* gathered, int world, long long P_local, float* mean, float* invstd,
                             float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                             float eps, int C, void* stream);
/* eval-mode BatchNorm (sampling with the generators in eval(), cyclegan.py:138-139; esrgan test_on_image.py:24-37):
 * invstd[c] = 1/sqrt(running_var[c] + eps). */
int migan_rsqrt_eps(const float* var, float* invstd, int C, float eps, void* stream);
/* Backward of `activation [-> Dropout2d]` behind a conv, viewed [G = N][P = H*W][C]: dx = dy * mask[g][c] * act'(y)
 * (mask_gc may be NULL, act may be 0, y = the layer output) AND migan_norm_colsum_slabs(G,P,C) x [C] column-sum slabs
 * of dx for the conv's bias gradient (dcgan.py:62-63,78: Conv->Tanh, Conv->LeakyReLU->Dropout2d). */
int migan_act_bwd_colsum(const float* dy, const float* y, const float* mask_gc, float* dx, float* csum, int G, int P,
                         int C, int act, float slope, void* stream);

/* ---- Pointwise / index-remap kernels (csrc/eltwise.hip) ------------------------------------------------
 * nn.LeakyReLU(0.2)/ReLU/Tanh/Sigmoid: dcgan.py:57,63,92  cyclegan/models.py:30,52,82,110 ... */
int migan_act_fwd(const float* x, float* y, size_t n, int act, float slope, void* stream);
int migan_act_bwd(const float* dy, const float* y, float* dx, size_t n, int act, float slope, void* stream);
/* Second-order terms for the conv-critic gradient penalties (dragan.py:144-167, stargan.py:142-161,
 * dualgan.py:116-135), where autograd.grad(..., create_graph=True) differentiates the backward pass again:
 * migan_act_bwd2: out = gg * g * d f'(y)/dy (tanh, sigmoid; zero for LeakyReLU/ReLU).
 * migan_norm_bwd2 (csrc/norm.hip): with d = gradient w.r.t. the norm output (first backward), u = gradient w.r.t. the
 * first backward's dx: gd = d(L)/d(d), gx = d(L)/d(x), dgamma (G == 1) - formulas in norm.hip.  ws: migan_norm_workspace2.
 * migan_dragan_interp: alpha*X + (1-alpha)*(X + 0.5*std(X)*noise) with std from the device scalar var_biased (the biased
 * variance over all n elements, migan_norm_moments on the flattened tensor) - dragan.py:147-149 without a host sync. */
int migan_act_bwd2(const float* g, const float* gg, const float* y, float* out, size_t n, int act, void* stream);
size_t migan_norm_workspace2(int G, int P, int C);
int migan_norm_bwd2(const float* x, const float* d, const float* u, const float* mean, const float* invstd,
                    const float* gamma, float* gd, float* gx, float* dgamma, int dgamma_accumulate, int G, int P, int C,
                    float* ws, size_t ws_bytes, void* stream);
int migan_dragan_interp(const float* x, const float* alpha, const float* noise, const float* var_biased, float* out,
                        size_t n, void* stream);
/* Relativistic average GAN logits (esrgan.py:137,165-166: pred_fake - pred_real.mean(0, keepdim=True)):
 * y[n][p] = alpha*a[n][p] + beta*mean_m b[m][p] over [N][P] tensors; a may be NULL (the backward into b). */
int migan_batch_mean_axpy(const float* a, const float* b, float* y, int N, size_t P, float alpha, float beta, void* stream);
/* nn.PReLU() single shared slope: srgan/models.py:24,38,57.  ws: migan_reduce_workspace() bytes. */
size_t migan_reduce_workspace(void);
int migan_prelu_fwd(const float* x, const float* a, float* y, size_t n, void* stream);
int migan_prelu_bwd(const float* x, const float* dy, const float* a, float* dx, float* da, float* ws, size_t n,
                    void* stream);
/* y = alpha*a + beta*b (b may be NULL): residual adds cyclegan/models.py:37, srgan/models.py:30,68;
 * loss combinations dcgan.py:180, cyclegan.py:202. */
int migan_axpby(const float* a, float alpha, const float* b, float beta, float* y, size_t n, void* stream);
int migan_mul(const float* a, const float* b, float* y, size_t n, void* stream);
/* nn.Dropout2d(0.25) dcgan.py:78: y[n][p][c] = x[n][p][c]*mask[n][c];  nn.Dropout(0.5) pix2pix/models.py:27,44
 * uses migan_mul with a full-size mask.  migan_rand_mask draws mask = Bernoulli(1-p)/(1-p) with
 * Philox4x32-10 at stream position counter[0] (device; NULL = position 0).  counter points to TWO 64-bit words
 * {position, ticket}, both zero-initialised by the caller: the kernel's last-arriving block advances the position
 * and resets the ticket, so the stream moves on graph replay without a second launch. */
int migan_mul_nc(const float* x, const float* mask, float* y, int N, int HW, int C, void* stream);
int migan_rand_mask(float* mask, size_t n, float p, unsigned long long seed, unsigned long long* counter,
                    void* stream);
/* Standalone nn.ReflectionPad2d / nn.ZeroPad2d / nn.Upsample(scale_factor=2) (cyclegan/models.py:27,49,74,117,
 * dcgan.py:54,58, pix2pix/models.py:77-78) and their backward (scatter of preimages, deterministic). */
int migan_gather2d_fwd(const float* x, float* y, int N, int Hi, int Wi, int C, int Ho, int Wo, int pad_t,
                       int pad_l, int mode, void* stream);
int migan_gather2d_bwd(const float* dy, float* dx, int N, int Hi, int Wi, int C, int Ho, int Wo, int pad_t,
                       int pad_l, int mode, void* stream);
/* nn.PixelShuffle(r) srgan/models.py:56: small [N][H][W][C*r*r] <-> large [N][H*r][W*r][C];
 * forward=1 reads small writes large, forward=0 the inverse (its backward). */
int migan_pixel_shuffle(const float* src, float* dst, int N, int H, int W, int C, int r, int forward,
                        void* stream);
/* MaxPool2d(2,2) inside vgg19.features[:18] (srgan/models.py:11-12). */
int migan_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int migan_maxpool2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* ... with the backward of the ReLU in front of the pool (vgg19.features[3:5], [8:10] behind srgan/models.py:11-12, differentiated
 * by srgan.py:128) applied to dx: x is that ReLU's output. */
int migan_maxpool2_relu_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* torch.cat((a,b),1) pix2pix/models.py:50,132 (forward=1) and its backward split (forward=0). */
int migan_cat_channels(float* a, float* b, float* y, size_t P, int Ca, int Cb, int forward, void* stream);
/* dst[dst_row ? dst_row[k] : k][:] = sel[k] >= 0 ? a[sel[k]][:] : b[-1 - sel[k]][:], rows of D floats (D % 4 == 0);
 * sel / dst_row: device int32 [n]; a row with dst_row[k] < 0 is skipped (padding of a fixed-length table: a recorded hipGraph
 * launches n = batch rows every step).  The device-resident image history of cyclegan/utils.py:13-33 (ReplayBuffer):
 * one launch assembles the returned batch from old pool entries and new samples, one writes the new samples into the
 * pool. */
int migan_select_rows(const float* a, const float* b, float* dst, const int* sel, const int* dst_row, int n, size_t D,
                      void* stream);
/* [B][R][C] -> [B][C][R]: NCHW<->NHWC re-layout at the boundary with reference code (.view in
 * dcgan.py:68,96). */
int migan_transpose_batched(const float* src, float* dst, int B, int R, int Cc, void* stream);
int migan_permute4d(const float* src, float* dst, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3,
                    void* stream);
/* Multi-tensor migan_permute4d: one launch for a table of permute copies (all weight packs of a training step - the layout
 * changes ATen makes per convolution call of dcgan.py:143-183 / cyclegan.py:159-239, here once per step).
 * entries: device array of { const float* src; float* dst; unsigned o1, o2, o3, pad; long long s[4]; long long n; } (72 bytes):
 * dst is contiguous with extents (n/(o1*o2*o3), o1, o2, o3) and dst[i0][i1][i2][i3] = src[i0*s[0] + i1*s[1] + i2*s[2] + i3*s[3]];
 * blocks: device array of { int entry; int chunk; }, ceil(n / 1024) consecutive chunks per entry. */
int migan_multi_permute4d(const void* entries, const void* blocks, int nblocks, void* stream);

/* ---- Input pipeline on the device (csrc/image_pipeline.hip) ------------------------------
 * What the reference's DataLoader workers compute per image between the decoded uint8 bitmap and the fp32 batch
 * (cyclegan.py:111-117, srgan/datasets.py:16-33, dcgan.py:120-131, pix2pix/datasets.py), bit-exact with Pillow + torchvision.
 * migan_resample_u8: ONE separable pass of Pillow's 8-bit ImagingResample over a batch [N][Hi][Wi][C] (C <= 4): axis 1 resamples
 *   the width (run first, as Pillow does), axis 0 the height.  kk [out][ksize] int32 coefficients (22 fractional bits) and
 *   bounds [out][2] (first source index, tap count) as Pillow's precompute_coeffs + normalize_coeffs_8bpc produce them
 *   (pytorch_gan_amd.data.pil_resample_coeffs); the pass output is rounded to uint8 like Pillow's intermediate image.
 * migan_u8_to_f32: crop window (crop_yx [N][2], NULL = corner) + horizontal flip (flip [N], NULL = none) + ToTensor (/255) +
 *   Normalize ((x - mean[c]) / std[c]; mean/std both NULL = ToTensor only) -> fp32 [N][h][w][C] (nchw 0) or [N][C][h][w]. */
int migan_resample_u8(const unsigned char* src, unsigned char* dst, const int* kk, const int* bounds, int ksize, int N, int Hi,
                      int Wi, int C, int out, int axis, void* stream);
int migan_u8_to_f32(const unsigned char* src, float* dst, const int* crop_yx, const unsigned char* flip, const float* mean,
                    const float* stdv, int N, int Hi, int Wi, int C, int h, int w, int nchw, void* stream);

/* ---- Reductions, losses, gradient penalty, optimiser (csrc/reduce_loss_adam.hip) -----------------------
 * bias gradients: out[c] = sum_p x[p][c] (aten::convolution_backward grad_bias / the sum of AddmmBackward under every
 * loss.backward() of the path: dcgan.py:168,182, cyclegan.py:204,221,238, srgan.py:128,144). */
size_t migan_colsum_workspace(size_t P, int C);
int migan_colsum(const float* x, float* out, size_t P, int C, float* ws, size_t ws_bytes, int accumulate,
                 void* stream);
/* kind 0 BCELoss (dcgan.py:103), 1 MSELoss (cyclegan.py:57), 2 L1Loss (cyclegan.py:58-59), 3 mean
 * (wgan_gp.py:171,189), 4 BCEWithLogitsLoss (relativistic_gan.py:95: (1-t)*x - log_sigmoid(x)).  Target is t[i] or the constant tconst when t==NULL.  out = mean over n.
 * ws: migan_reduce_workspace() bytes.  bwd: dx = g[0]/n * dloss/dx. */
int migan_loss_fwd(int kind, const float* x, const float* t, float tconst, float* out, size_t n, float* ws,
                   size_t ws_bytes, void* stream);
int migan_loss_bwd(int kind, const float* x, const float* t, float tconst, const float* g, float* dx, size_t n,
                   void* stream);
/* ---- Layers of the DCGAN-block clones, SURVEY.md 8f F2 (csrc/classify.hip) --------------------------------------
 * nn.Embedding(V, D) (acgan.py:50): y[i] = w[idx[i]]; backward dw[v] (+)= sum_{i: idx[i]==v} dy[i] in index order
 * (deterministic).  idx: int64 device tensor. */
int migan_embedding_fwd(const float* w, const long long* idx, float* y, int n, int D, int V, void* stream);
int migan_embedding_bwd(const float* dy, const long long* idx, float* dw, int n, int D, int V, int accumulate,
                        void* stream);
/* nn.Softmax over the class dim of [B][C] (acgan.py:100) and its backward dx = y*(dy - sum_c dy*y). */
int migan_softmax_fwd(const float* x, float* y, int B, int C, void* stream);
int migan_softmax_bwd(const float* y, const float* dy, float* dx, int B, int C, void* stream);
/* nn.CrossEntropyLoss() (mean) on logits [B][C] with int64 class targets (acgan.py:113).  ws: 2*B floats; its second
 * half (ws + B, the row logsumexp values) is the `lse` argument of the backward: dx = g/B * (softmax(x) - onehot). */
int migan_cross_entropy_fwd(const float* x, const long long* target, float* out, float* ws, int B, int C, void* stream);
int migan_cross_entropy_bwd(const float* x, const long long* target, const float* lse, const float* g, float* dx, int B,
                            int C, void* stream);
/* gradients.norm(2, dim=1) and its derivatives (wgan_gp.py:136-137). */
int migan_rownorm_fwd(const float* x, float* out, int B, int D, void* stream);
int migan_rownorm_bwd(const float* x, const float* nrm, const float* dn, float* dx, int B, int D, void* stream);
int migan_rowscale(const float* x, const float* s, float* y, int B, int D, void* stream);
/* ebgan.py:142-148 pullaway_loss: loss[0] = (sum_ij <n_i, n_j> - B) / (B (B-1)) over the row-normalised embeddings e [B][D]
 * (the generator's repelling regulariser, ebgan.py:179); ws: (D + B) floats kept for migan_pullaway_bwd, which writes
 * de [B][D] = g[0] * d loss / d e. */
int migan_pullaway_fwd(const float* e, float* loss, float* ws, int B, int D, void* stream);
int migan_pullaway_bwd(const float* e, const float* ws, const float* g, float* de, int B, int D, void* stream);
/* torch.optim.Adam(lr, betas) step for all tensors of one optimiser in ONE launch (dcgan.py:134-135,169,183).
 * tab: device array of {float* p; const float* g; float* m; float* v; long long n}; blk: device array of
 * {int tensor; int chunk} with chunk size migan_adam_chunk(); step: device float holding the number of updates done
 * (torch keeps Adam's step as an fp32 tensor as well): every block computes with step+1 and the last block to finish
 * publishes it, so a captured hipGraph advances it on replay; ticket: device uint32, zero-initialised by the caller.
 * lr_dev: optional device float that overrides `lr` (learning-rate schedules under graph replay,
 * cyclegan.py:95-103,275-277).  grads are multiplied by grad_scale (1/world_size after a summed all-reduce). */
int migan_adam_chunk(void);
int migan_adam_step(const void* tab, const void* blk, int nblocks, float* step, unsigned* ticket,
                    const float* lr_dev, float lr, float b1, float b2, float eps, float grad_scale, void* stream);

/* Weight-stationary Conv2d(64, 64, 3, 1, 1) (csrc/conv_c64.hip): the residual trunk of srgan/models.py:22-30,47 - forward, and with
 * migan_c64_pack(flip = 1) the input gradient of the same layer (aten::convolution_backward's grad_input behind srgan.py:128).  The
 * weights stay in registers, the input rows in an LDS ring: every input element enters a CU once.
 * migan_c64_conv_ok: 1 when the kernel takes the geometry (W % 32 == 0, >= 1024 row-steps); wp: migan_c64_pack_floats() floats from
 * migan_c64_pack(w_oihw [64][64][3][3]).
 * in_mean / in_invstd [64] (optional, both or neither; migan_norm_stats' outputs) with in_gamma / in_beta [64] (each optional = 1 / 0):
 * x is read through T(v) = in_act(fma(v, sc, sh)), sc = in_invstd[c] * in_gamma[c], sh = in_beta[c] - in_mean[c] * sc - the
 * BatchNorm2d(64, 0.8) -> PReLU of srgan/models.py:23-24 folded into the consumer conv's operand path, in migan_norm_apply's arithmetic
 * (in_act = ACT_LRELU with the slope read from in_slope_ptr, or in_slope when in_slope_ptr == NULL); zero padding is applied after T.
 * accumulate != 0: y +=. */
int migan_c64_conv_ok(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad_t, int pad_l, int pad_b, int pad_r,
                      int gather);
size_t migan_c64_pack_floats(void);
int migan_c64_pack(const float* w_oihw, float* wp, int flip, void* stream);
/* every such layer of a step in one launch: tab = device array of n records {const float* w_oihw; float* wp_fwd; float* wp_dgrad}
 * (a pack pointer may be NULL) */
int migan_c64_pack_multi(const void* tab, int n, void* stream);
int migan_c64_conv_fwd(const float* x, const float* wp, const float* bias, float* y, int N, int H, int W, int act, float slope,
                       int accumulate, const float* in_mean, const float* in_invstd, const float* in_gamma, const float* in_beta,
                       int in_act, float in_slope, const float* in_slope_ptr, void* stream);

/* Weight gradient of the same layer, accumulators stationary (csrc/conv_c64.hip c64_wgrad_kernel): dw_oihw [64][64][3][3] (accumulate: +=)
 * from x [N][H][W][64] - read through the same input map T as the forward - and dy [N][H][W][64]; one [64][576] slab per workgroup in ws
 * (migan_c64_wgrad_workspace() bytes), added in a fixed order; db (optional) from the caller's column-sum slabs of dy, as migan_conv2d_wgrad.
 * aten::convolution_backward's grad_weight behind srgan.py:128 for srgan/models.py:22-27. */
size_t migan_c64_wgrad_workspace(int N, int H, int W);
int migan_c64_conv_wgrad(const float* x, const float* dy, float* dw_oihw, float* ws, size_t ws_bytes, int N, int H, int W,
                         int accumulate, float* db, int db_accumulate, const float* db_slabs, int db_nslab,
                         const float* in_mean, const float* in_invstd, const float* in_gamma, const float* in_beta, int in_act,
                         float in_slope, const float* in_slope_ptr, void* stream);
/* The generator's last block with its BatchNorm folded into the image-output conv (dcgan.py:60-62: nn.BatchNorm2d(64, 0.8),
 * nn.LeakyReLU(0.2), nn.Conv2d(64, channels, 3, stride=1, padding=1), nn.Tanh()).
 * migan_conv2d_fwd_normed: y = act(conv(T(x), w_ohwi) + bias), T(v) = in_act((v - in_mean[c]) * in_invstd[c] * in_gamma[c] + in_beta[c]) applied
 * while the thin-N kernel stages its window (migan_norm_apply's arithmetic; zero padding after T; in_gamma / in_beta may be NULL): the
 * normalised, activated tensor is never stored.  _ok: 1 when the geometry is served (<= 4 output channels, Ci % 4 == 0, Ci <= 256).
 * migan_bn_conv1_bwd (one output channel, 3x3, stride 1, padding 1): from x (the BatchNorm input), dz [N][H][W] (gradient at the conv's
 * pre-activation output), the statistics and the weights - dx, dw_oihw [1][C][3][3], db [1] (optional), dgamma, dbeta and (csum != NULL) the
 * migan_norm_colsum_slabs(1, N*H*W, C) column-sum slabs of dx; the conv's input gradient is recomputed from dz where it is used, never
 * stored.  aten::convolution_backward + native_batch_norm_backward + leaky_relu_backward behind dcgan.py:168. */
int migan_conv2d_fwd_normed_ok(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad_t, int pad_l);
int migan_conv2d_fwd_normed(const float* x, const float* w_ohwi, const float* bias, float* y, int N, int Hi, int Wi, int Ci, int Ho, int Wo,
                            int Co, int R, int S, int stride, int pad_t, int pad_l, int act, float slope, const float* in_mean,
                            const float* in_invstd, const float* in_gamma, const float* in_beta, int in_act, float in_slope, void* stream);
int migan_bn_conv1_bwd_ok(int N, int H, int W, int C);
size_t migan_bn_conv1_bwd_workspace(int N, int H, int W, int C);
int migan_bn_conv1_bwd(const float* x, const float* dz, const float* w_ohwi, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, int act, float slope, float* dx, float* dw_oihw, int dw_accumulate, float* db, int db_accumulate,
                       float* dgamma, float* dbeta, int affine_accumulate, float* csum, float* ws, size_t ws_bytes, int N, int H, int W, int C,
                       void* stream);
/* optimizer.zero_grad() on the flat gradient bucket (dcgan.py:157,175; cyclegan.py:177,211,228): p[0 .. bytes) = 0, 16-byte stores (p and
 * bytes multiples of 4) */
int migan_zero(void* p, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIGAN_H */
