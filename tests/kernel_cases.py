"""Geometries that select the SPECIALISED kernels of the library, with the kernel symbols each must launch.

Dispatch inside libmigan.so is by geometry only (no run-time switches), so "which kernel ran" is a property of the shape: the GPU
suite (tests/test_ops_gpu.py::test_geometry_selects_kernel) runs every case on the MI355X, asserts the launch counters of the C
ABI (migan_debug_launch_count) for the symbols listed here and compares the results with torch on the host; the execution-model
suite runs the same cases under permuted wave schedules (tests/test_kernels_emu_cpu.py).  Test infrastructure only."""
import torch

# name, (N, Ci, H, W, Co, k, stride, pads, gather, act, bias), symbols that must launch (forward + backward of the case)
CONV = [
    # PatchGAN heads (cyclegan/models.py:118, pix2pix/models.py:130): one wave per output pixel; dgrad into 512 channels on 16-pixel tiles
    ("patchgan_head", (2, 512, 16, 16, 1, 4, 1, (2, 2, 1, 1), 0, 0, True), ("thin_conv_wave_kernel", "smallk_tile_kernel<K, 16>")),
    ("patchgan_head_b1", (1, 256, 9, 9, 1, 4, 1, (2, 2, 1, 1), 0, 0, True), ("thin_conv_wave_kernel", "smallk_tile_kernel<K, 16>")),
    # first convs of the image nets (pix2pix/models.py:115, srgan/models.py:85): the mid-K direct kernel
    ("first_conv_6ch", (1, 6, 32, 32, 64, 4, 2, (1, 1, 1, 1), 0, 0, False), ("midk_tile_kernel",)),
    ("first_conv_3ch", (2, 3, 24, 24, 64, 3, 1, (1, 1, 1, 1), 0, 0, True), ("midk_tile_kernel",)),
    # inner U-Net levels (pix2pix/models.py:62-67): few pixels x >= 1 M weights = the weight-streaming path on the stored layout
    ("unet_inner", (1, 512, 4, 4, 512, 4, 2, (1, 1, 1, 1), 0, 0, True), ("im2col_small_kernel", "col2im_small_kernel", "skinny_tn_kernel")),
    ("fewpix_16px", (1, 128, 8, 8, 512, 4, 2, (1, 1, 1, 1), 0, 1, True), ("im2col_small_kernel", "col2im_small_kernel", "skinny_nn_kernel")),
    ("fewpix_1px", (1, 256, 2, 2, 512, 4, 2, (1, 1, 1, 1), 0, 0, False), ("im2col_small_kernel", "col2im_small_kernel")),
    ("fewpix_3x3_s1", (2, 256, 2, 2, 512, 3, 1, (1, 1, 1, 1), 0, 2, True), ("im2col_small_kernel", "col2im_small_kernel")),
    # 512 k-element weight at 64 pixels: tiled OHWI / IHWO packs, split-K slabs
    ("unet_mid", (1, 128, 16, 16, 256, 4, 2, (1, 1, 1, 1), 0, 0, False), ("pack_transpose_kernel",)),
    # Conv2d(64, 64, 3, 1, 1) on W % 32 == 0 columns (srgan/models.py:22-30: the residual trunk): weight-stationary kernels - forward, input
    # gradient, weight gradient (tests/conftest.py lowers its size gate, MIGAN_C64_MIN_STEPS, so that this small shape takes it); 3 strips x 2 images x 13 rows
    ("trunk_c64", (2, 64, 13, 96, 64, 3, 1, (1, 1, 1, 1), 0, 1, True), ("c64_conv_kernel<0>", "c64_wgrad_kernel<0>")),
    # 4 M-element weight at 81 output pixels (above the few-pixel path): tiled packs + the transposing slab reduction
    ("unet_81px", (1, 512, 18, 18, 512, 4, 2, (1, 1, 1, 1), 0, 0, False), ("pack_transpose_kernel", "wgrad_reduce_tr_kernel")),
]
# name, (N, Cin, H, W, Cout, act, bias): nn.ConvTranspose2d(Cin, Cout, 4, 2, 1) (pix2pix/models.py:39)
CONVT = [
    ("fewpix_convT_4px", (1, 512, 2, 2, 128, 2, True), ("im2col_small_kernel", "col2im_small_kernel", "skinny_tn_kernel")),
    ("fewpix_convT_1px", (1, 512, 1, 1, 256, 0, False), ("im2col_small_kernel", "col2im_small_kernel")),
]
# name, (N, C, H, W, act, affine, mask, residual): nn.InstanceNorm2d at <= 1024 pixels (pix2pix/models.py:25,42) in one launch per direction
NORM = [
    ("in_4x4", (1, 512, 4, 4, 0, False, False, False), ("norm_small_fwd_kernel", "norm_small_bwd_kernel")),
    ("in_16x16_lrelu_mask", (2, 64, 16, 16, 1, False, True, False), ("norm_small_fwd_kernel", "norm_small_bwd_kernel")),
    ("in_32x32_res", (2, 64, 32, 32, 0, False, False, True), ("norm_small_fwd_kernel", "norm_small_bwd_kernel")),
]


def _rand(gen, *shape, scale=1.0):
    return torch.randn(*shape, device=gen.device, generator=gen) * scale


def _act(t, act):
    import torch.nn.functional as TF

    return {0: lambda v: v, 1: lambda v: TF.leaky_relu(v, 0.2), 2: torch.relu, 3: torch.tanh}[act](t)


def conv_inputs(case, gen):
    N, Ci, H, W, Co, k, stride, pads, gather, act, bias = case
    x = _rand(gen, N, Ci, H, W)
    w = _rand(gen, Co, Ci, k, k, scale=0.1)
    b = _rand(gen, Co) if bias else None
    Ho = (H + pads[0] + pads[2] - k) // stride + 1
    Wo = (W + pads[1] + pads[3] - k) // stride + 1
    return x, w, b, _rand(gen, N, Co, Ho, Wo)


def run_conv(F, case, tensors):
    """The case through the host mirror (device = wherever `tensors` live) -> {"y", "dx", "dw"[, "db"]}"""
    N, Ci, H, W, Co, k, stride, pads, gather, act, bias = case
    x, w, b, gy = [None if t is None else t.clone().requires_grad_(i < 3) for i, t in enumerate(tensors)]
    y = F.conv2d(x, w, b, stride, pads, gather, act, 0.2)
    y.backward(gy)
    out = {"y": y.detach(), "dx": x.grad, "dw": w.grad}
    if bias:
        out["db"] = b.grad
    return {k_: torch.Tensor.contiguous(F._plain(v).detach().clone()) for k_, v in out.items()}


def ref_conv(case, tensors):
    """The same case on stock torch (host)"""
    import torch.nn.functional as TF

    N, Ci, H, W, Co, k, stride, pads, gather, act, bias = case
    x, w, b, gy = [None if t is None else t.detach().cpu().clone().requires_grad_(i < 3) for i, t in enumerate(tensors)]
    pt, pl, pb, pr = pads
    z = TF.conv2d(TF.pad(x, (pl, pr, pt, pb)), w, b, stride)
    y = _act(z, act)
    y.backward(gy)
    out = {"y": y.detach(), "dx": x.grad, "dw": w.grad, "_pre": z.detach()}
    if bias:
        out["db"] = b.grad
    return out


def convt_inputs(case, gen):
    N, Cin, H, W, Cout, act, bias = case
    return _rand(gen, N, Cin, H, W), _rand(gen, Cin, Cout, 4, 4, scale=0.1), (_rand(gen, Cout) if bias else None), _rand(gen, N, Cout, 2 * H, 2 * W)


def run_convt(F, case, tensors):
    N, Cin, H, W, Cout, act, bias = case
    x, w, b, gy = [None if t is None else t.clone().requires_grad_(i < 3) for i, t in enumerate(tensors)]
    y = F.conv_transpose2d(x, w, b, 2, 1, act, 0.2)
    y.backward(gy)
    out = {"y": y.detach(), "dx": x.grad, "dw": w.grad}
    if bias:
        out["db"] = b.grad
    return {k_: torch.Tensor.contiguous(F._plain(v).detach().clone()) for k_, v in out.items()}


def ref_convt(case, tensors):
    import torch.nn.functional as TF

    N, Cin, H, W, Cout, act, bias = case
    x, w, b, gy = [None if t is None else t.detach().cpu().clone().requires_grad_(i < 3) for i, t in enumerate(tensors)]
    z = TF.conv_transpose2d(x, w, b, 2, 1)
    y = _act(z, act)
    y.backward(gy)
    out = {"y": y.detach(), "dx": x.grad, "dw": w.grad, "_pre": z.detach()}
    if bias:
        out["db"] = b.grad
    return out


def norm_inputs(case, gen):
    N, C, H, W, act, affine, mask, res = case
    cl = torch.channels_last
    x = (_rand(gen, N, C, H, W) * 2 + 0.3).contiguous(memory_format=cl)
    gamma = (_rand(gen, C) * 0.2 + 1.0) if affine else None
    beta = _rand(gen, C) if affine else None
    r = _rand(gen, N, C, H, W).contiguous(memory_format=cl) if res else None
    m = (torch.rand(N, C, H, W, device=gen.device, generator=gen) > 0.5).float().mul_(2.0).contiguous(memory_format=cl) if mask else None
    return x, gamma, beta, r, m, _rand(gen, N, C, H, W).contiguous(memory_format=cl)


def run_norm(F, case, tensors):
    N, C, H, W, act, affine, mask, res = case
    x, gamma, beta, r, m, gy = tensors
    x = x.clone().requires_grad_(True)
    gamma = gamma.clone().requires_grad_(True) if affine else None
    beta = beta.clone().requires_grad_(True) if affine else None
    assert F.norm_small_takes(x, True), "the one-launch InstanceNorm must take this shape"
    y = F.norm(x, gamma, beta, res=r, instance=True, act=act, slope=0.2, mask=m)
    y.backward(gy)
    out = {"y": y.detach(), "dx": x.grad}
    if affine:
        out.update(dgamma=gamma.grad, dbeta=beta.grad)
    return {k_: F._plain(v).detach().clone() for k_, v in out.items()}


def ref_norm(case, tensors):
    import torch.nn.functional as TF

    N, C, H, W, act, affine, mask, res = case
    # NCHW-contiguous copies: stock torch's instance_norm backward on the host is wrong for a channels_last (1, C, H, W) input (checked
    # against an fp64 evaluation: rel 1.4; the contiguous call agrees to 6e-8)
    x, gamma, beta, r, m, gy = [None if t is None else t.detach().cpu().contiguous().clone() for t in tensors]
    x.requires_grad_(True)
    if affine:
        gamma.requires_grad_(True)
        beta.requires_grad_(True)
    z = TF.instance_norm(x, weight=gamma, bias=beta, eps=1e-5)
    if res:
        z = z + r
    y = _act(z, act)
    if mask:
        y = y * m
    y.backward(gy)
    out = {"y": y.detach(), "dx": x.grad, "_pre": z.detach()}
    if affine:
        out.update(dgamma=gamma.grad, dbeta=beta.grad)
    return out


def all_cases():
    """(name, symbols, inputs(gen), run(F, tensors), ref(tensors))"""
    for name, case, syms in CONV:
        yield name, syms, (lambda gen, c=case: conv_inputs(c, gen)), (lambda F, t, c=case: run_conv(F, c, t)), (lambda t, c=case: ref_conv(c, t))
    for name, case, syms in CONVT:
        yield name, syms, (lambda gen, c=case: convt_inputs(c, gen)), (lambda F, t, c=case: run_convt(F, c, t)), (lambda t, c=case: ref_convt(c, t))
    for name, case, syms in NORM:
        yield name, syms, (lambda gen, c=case: norm_inputs(c, gen)), (lambda F, t, c=case: run_norm(F, c, t)), (lambda t, c=case: ref_norm(c, t))


def quiet_scope(F):
    """Run outside any step scope: -> the saved scope state (restore with restore_scope)."""
    saved = (F._CACHE_SCOPE, F._SCOPE_OWNER, F._INPUT_GRAD_ONLY)
    F._CACHE_SCOPE, F._SCOPE_OWNER, F._INPUT_GRAD_ONLY = None, None, False
    return saved


def restore_scope(F, saved):
    F._CACHE_SCOPE, F._SCOPE_OWNER, F._INPUT_GRAD_ONLY = saved
