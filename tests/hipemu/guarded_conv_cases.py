"""TEST INFRASTRUCTURE: the conv cases of the GPU suite on the host execution model with EVERY operand (inputs, packed weights,
bias, outputs, workspaces) placed against an inaccessible page: a kernel that reads or writes past the end of a tensor - which a
GPU run never notices, the neighbouring allocation is mapped - dies with SIGSEGV here.  Outputs and workspaces start as NaN: an element a kernel
forgets to write, or a workspace slab it reads before writing, shows up as a NaN in the result.  Run as a subprocess by
tests/test_kernels_emu_cpu.py::test_no_conv_kernel_leaves_its_tensors (prints the case before running it)."""
import ctypes, mmap, sys, re
import os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np, torch
import hipemu
emu = hipemu.load()
libc = ctypes.CDLL(None)
libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
PAGE = 4096
keep = []
def guarded(t):
    """copy of tensor t whose last byte is (up to 15 B before) the last byte in front of a PROT_NONE page, and whose first byte sits right after one (when size % 4096 == 0 this is exact)"""
    nbytes = t.numel() * t.element_size()
    body = (nbytes + PAGE - 1) // PAGE * PAGE
    m = mmap.mmap(-1, body + 2 * PAGE)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    assert libc.mprotect(base, PAGE, 0) == 0 and libc.mprotect(base + PAGE + body, PAGE, 0) == 0
    start = base + PAGE + body - (nbytes + 15) // 16 * 16
    arr = np.ctypeslib.as_array((ctypes.c_float * t.numel()).from_address(start)) if t.dtype == torch.float32 else None
    g = torch.from_numpy(arr).view(t.shape)
    g.copy_(t)
    keep.append(m)
    return g
import test_kernels_emu_cpu as K
def _conv_case_guarded(case):
    N, Ci, H, W, Co, k, stride, pads, gather, act, bias = case
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Ci, H, W, generator=g); w = torch.randn(Co, Ci, k, k, generator=g) * 0.2
    b = torch.randn(Co, generator=g) if bias else None
    y_ref = TF.conv2d(K._gather_ref(x, pads, gather), w, b, stride)
    Ho, Wo = y_ref.shape[2:]
    P = K._ptr
    sk = guarded(torch.zeros(emu.migan_conv_splitk_workspace() // 4))
    xn = guarded(x.permute(0, 2, 3, 1).contiguous()); wo = guarded(w.permute(0, 2, 3, 1).contiguous()); wi = guarded(w.permute(1, 2, 3, 0).contiguous())
    bg = guarded(b) if bias else None
    y = guarded(torch.full((N, Ho, Wo, Co), float("nan")))
    rc = emu.migan_conv2d_fwd_ws(P(xn), P(wo), P(bg), None, P(y), N, H, W, Ci, Ho, Wo, Co, k, k, stride, pads[0], pads[1], gather, act, 0.2, P(sk), sk.numel() * 4, None)
    assert rc == 0
    assert not torch.isnan(y).any(), 'fwd NaN'
    if Co % 4 == 0:
        # the same launch with the [N][Co] multiplier of the fused nn.Dropout2d (dcgan.py:77-78; 16-byte loads of the mask in the channel-quad
        # epilogue): the mask against a guard page, y2 == y * mask
        mk = guarded((torch.rand(N, Co, generator=g) > 0.25).float() * (1 / 0.75))
        y2 = guarded(torch.full((N, Ho, Wo, Co), float("nan")))
        rc = emu.migan_conv2d_fwd_ws(P(xn), P(wo), P(bg), P(mk), P(y2), N, H, W, Ci, Ho, Wo, Co, k, k, stride, pads[0], pads[1], gather, act, 0.2, P(sk), sk.numel() * 4, None)
        assert rc == 0
        assert torch.allclose(y2, y * mk.view(N, 1, 1, Co), rtol=1e-6, atol=0), 'fwd with the [N][Co] mask'
    if gather == 0:
        gy = guarded(torch.randn(N, Ho, Wo, Co, generator=g))
        if pads[0] == pads[2] and pads[1] == pads[3]:
            dx = guarded(torch.full((N, H, W, Ci), float("nan")))
            assert emu.migan_conv2d_dgrad_ws(P(gy), P(wi), None, P(dx), N, H, W, Ci, Ho, Wo, Co, k, k, stride, pads[0], pads[1], 0, 0.0, P(sk), sk.numel() * 4, None) == 0
            assert not torch.isnan(dx).any(), 'dgrad NaN'
            # the same input gradient through the ReLU that produced this conv's input (mask = that ReLU's output, in the launch's epilogue);
            # 801 = this geometry has no kernel with the epilogue (the host applies the mask as its own pass)
            ro = guarded(torch.relu(torch.randn(N, H, W, Ci, generator=g)))
            dxr = guarded(torch.full((N, H, W, Ci), float("nan")))
            rc = emu.migan_conv2d_dgrad_relu_ws(P(gy), P(wi), P(dxr), P(ro), N, H, W, Ci, Ho, Wo, Co, k, k, stride, pads[0], pads[1], P(sk), sk.numel() * 4, None)
            assert rc in (0, 801), rc
            if rc == 0:
                want = torch.where(ro > 0, dx, torch.zeros(()))   # (bit-equal when both launches are the LDS-DMA kernel; MIGAN_DMA=0 runs the
                assert torch.equal(dxr == 0, want == 0) and K._rel(dxr, want) < 3e-6, 'dgrad with the ReLU mask epilogue'   # plain launch on the other family)
        wsb = emu.migan_conv2d_wgrad_workspace(N, Ho, Wo, Co, k, k, Ci)
        ws = guarded(torch.full((max(wsb // 4, 4),), float("nan"))); dw = guarded(torch.full((Co, Ci, k, k), float("nan")))
        assert emu.migan_conv2d_wgrad(P(xn), P(gy), P(dw), P(ws), wsb, N, H, W, Ci, Ho, Wo, Co, k, k, stride, pads[0], pads[1], gather, 0, None, 0, None, 0, None) == 0
        assert not torch.isnan(dw).any(), 'wgrad NaN'
def _fewpix_case_guarded(N, Ci, H, W, Co):
    """csrc/fewpix.hip on a Conv2d(Ci, Co, 4, 2, 1) and the ConvTranspose2d(Co, Ci, 4, 2, 1) that mirrors it: the two index
    kernels and the three skinny GEMM forms at these (streaming) sizes, every operand against a guard page, outputs NaN-filled,
    results against torch."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(3)
    P = K._ptr
    x = torch.randn(N, Ci, H, W, generator=g); w = torch.randn(Co, Ci, 4, 4, generator=g) * 0.05; b = torch.randn(Co, generator=g)
    y_ref = TF.leaky_relu(TF.conv2d(x, w, b, 2, 1), 0.2)
    Ho, Wo = y_ref.shape[2:]
    M, Kk = N * Ho * Wo, Ci * 16
    assert emu.migan_fewpix_ok(M, Co, Kk) == 1, (M, Co, Kk)
    xn, wg, bg = guarded(x.permute(0, 2, 3, 1).contiguous()), guarded(w), guarded(b)
    col = guarded(torch.full((M, Kk), float("nan")))
    assert emu.migan_im2col_small(P(xn), P(col), N, H, W, Ci, Ho, Wo, 4, 4, 2, 1, 1, None) == 0
    y = guarded(torch.full((N, Ho, Wo, Co), float("nan")))
    assert emu.migan_skinny_nt(P(col), P(wg), P(bg), P(y), M, Co, Kk, 1, 0.2, None) == 0
    assert K._rel(y.permute(0, 3, 1, 2), y_ref) < 3e-6, "fewpix fwd"
    nb = emu.migan_fewpix_nt_workspace(M, Co, Kk)   # the same product with K split over workgroups: partial tiles NaN-filled
    ws = guarded(torch.full((max(nb // 4, 4),), float("nan")))
    y2 = guarded(torch.full((N, Ho, Wo, Co), float("nan")))
    assert emu.migan_fewpix_nt(P(col), P(wg), P(bg), P(y2), P(ws), nb, M, Co, Kk, 1, 0.2, None) == 0
    assert K._rel(y2.permute(0, 3, 1, 2), y_ref) < 3e-6, "fewpix fwd (K split %d B)" % nb
    gy = torch.randn(N, Co, Ho, Wo, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    TF.conv2d(xr, wr, None, 2, 1).backward(gy)
    gyn = guarded(gy.permute(0, 2, 3, 1).contiguous())
    ycol = guarded(torch.full((M, Kk), float("nan")))
    assert emu.migan_skinny_nn(P(gyn), P(wg), P(ycol), M, Co, Kk, None) == 0
    dx = guarded(torch.full((N, H, W, Ci), float("nan")))
    assert emu.migan_col2im_small(P(ycol), None, P(dx), N, H, W, Ci, Ho, Wo, 4, 4, 2, 1, 1, 0, 0.0, None) == 0
    assert K._rel(dx.permute(0, 3, 1, 2), xr.grad) < 3e-6, "fewpix dgrad"
    dw = guarded(torch.full((Co, Ci, 4, 4), float("nan"))); db = guarded(torch.full((Co,), float("nan")))
    assert emu.migan_skinny_tn(P(gyn), P(col), P(dw), P(db), M, Co, Kk, 0, 0, None) == 0
    assert K._rel(dw, wr.grad) < 1e-5 and K._rel(db, gy.sum((0, 2, 3))) < 1e-5, "fewpix wgrad"
    # the transposed conv that maps y's shape back to x's: weight [Co][Ci][4][4] read as [Cin = Co][Cout*16 = Ci*16]
    t_ref = TF.conv_transpose2d(gy, w, None, 2, 1)
    if t_ref.shape[2:] == (H, W):
        tcol = guarded(torch.full((M, Kk), float("nan")))
        assert emu.migan_skinny_nn(P(gyn), P(wg), P(tcol), M, Co, Kk, None) == 0
        t = guarded(torch.full((N, H, W, Ci), float("nan")))
        assert emu.migan_col2im_small(P(tcol), None, P(t), N, H, W, Ci, Ho, Wo, 4, 4, 2, 1, 1, 2, 0.0, None) == 0
        assert K._rel(t.permute(0, 3, 1, 2), torch.relu(t_ref)) < 3e-6, "fewpix convT fwd"


def _upconv_case_guarded(N, Ci, H, W, Co):
    """Upsample(2) + Conv2d(Ci, Co, 3, 1, 1) phase-collapsed (dcgan.py:54-59): pack, forward, input gradient, weight gradient with
    its four-class slab reduction (upconv_wgrad_reduce_kernel: clamped slab indices, the bias slabs behind the partials) - every
    operand against a guard page, outputs and the workspace NaN-filled, results against torch."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(5)
    P = K._ptr
    x = torch.randn(N, Ci, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(Co, generator=g)
    y_ref = TF.conv2d(TF.interpolate(x, scale_factor=2, mode="nearest"), w, b, 1, 1)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xn, wg, bg = guarded(x.detach().permute(0, 2, 3, 1).contiguous()), guarded(w.detach().clone()), guarded(b)
    wf, wd = guarded(torch.full((Co * 16 * Ci,), float("nan"))), guarded(torch.full((Co * 16 * Ci,), float("nan")))
    assert emu.migan_upconv3x3_pack(P(wg), P(wf), P(wd), Co, Ci, None) == 0
    y = guarded(torch.full((N, 2 * H, 2 * W, Co), float("nan")))
    assert emu.migan_upconv3x3_fwd(P(xn), P(wf), P(bg), P(y), N, H, W, Ci, Co, 0, 0.0, None) == 0
    assert K._rel(y.permute(0, 3, 1, 2), y_ref.detach()) < 3e-6, "upconv fwd"
    gyn = guarded(gy.permute(0, 2, 3, 1).contiguous())
    dx = guarded(torch.full((N, H, W, Ci), float("nan")))
    assert emu.migan_upconv3x3_dgrad(P(gyn), P(wd), P(dx), N, H, W, Ci, Co, None) == 0
    assert K._rel(dx.permute(0, 3, 1, 2), x.grad) < 3e-6, "upconv dgrad"
    nb = emu.migan_upconv3x3_wgrad_workspace(N, H, W, Co, Ci)
    ws = guarded(torch.full((max(nb // 4, 4),), float("nan")))
    dw, db = guarded(torch.full((Co, Ci, 3, 3), float("nan"))), guarded(torch.full((Co,), float("nan")))
    assert emu.migan_upconv3x3_wgrad(P(xn), P(gyn), P(dw), P(ws), nb, N, H, W, Ci, Co, 0, P(db), 0, None, 0, None) == 0
    assert K._rel(dw, w.grad) < 1e-5 and K._rel(db, gy.sum((0, 2, 3))) < 1e-5, "upconv wgrad (%d B of slabs)" % nb


def _norm_case_guarded(G, N, HW, C, act):
    """BatchNorm (G = 1) / InstanceNorm (G = N) statistics, apply, backward and the column-sum reduction: the chunked partial passes
    and their one-wave-per-channel finalize kernels (four records per round of loads, clamped record index) - every operand against a
    guard page, outputs and workspaces NaN-filled, results against torch."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(7)
    P = K._ptr
    Pp = N * HW // G   # pixels per group
    x = (torch.randn(N, HW, C, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).requires_grad_(True), torch.randn(C, generator=g).requires_grad_(True)
    xg = x.view(G, Pp, C)
    mean_ref, var_ref = xg.mean(1), xg.var(1, unbiased=False)
    yl = ((xg - mean_ref[:, None]) / torch.sqrt(var_ref[:, None] + 1e-5) * gamma + beta)
    y_ref = TF.leaky_relu(yl, 0.2) if act == 1 else yl
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xs = guarded(x.detach().clone())
    gm, bt = guarded(gamma.detach().clone()), guarded(beta.detach().clone())
    mean, invstd = guarded(torch.full((G * C,), float("nan"))), guarded(torch.full((G * C,), float("nan")))
    nb = emu.migan_norm_workspace(G, Pp, C)
    ws = guarded(torch.full((max(nb // 4, 4),), float("nan")))
    assert emu.migan_norm_stats(P(xs), P(mean), P(invstd), None, None, None, 0.1, 1e-5, G, Pp, C, P(ws), nb, None) == 0
    assert K._rel(mean.view(G, C), mean_ref.detach()) < 1e-5 and K._rel(invstd.view(G, C), 1 / torch.sqrt(var_ref.detach() + 1e-5)) < 1e-5, "stats"
    y = guarded(torch.full((N, HW, C), float("nan")))
    assert emu.migan_norm_apply(P(xs), P(y), P(mean), P(invstd), P(gm), P(bt), None, G, Pp, C, act, 0.2, None) == 0
    assert K._rel(y.view(G, Pp, C), y_ref.detach()) < 3e-6, "apply"
    gyn = guarded(gy.reshape(N, HW, C).contiguous())
    dx = guarded(torch.full((N, HW, C), float("nan")))
    dg, dbt = (guarded(torch.full((C,), float("nan"))), guarded(torch.full((C,), float("nan")))) if G == 1 else (None, None)
    ws.fill_(float("nan"))
    assert emu.migan_norm_bwd(P(xs), P(gyn), P(mean), P(invstd), P(gm), P(bt), P(dx), P(dg), P(dbt), G, Pp, C, act, 0.2, P(ws), nb, 0,
                              None, None) == 0
    assert K._rel(dx, x.grad) < 2e-5, "bwd dx %g" % K._rel(dx, x.grad)
    if G == 1:
        assert K._rel(dg, gamma.grad) < 2e-5 and K._rel(dbt, beta.grad) < 2e-5, "bwd dgamma / dbeta"
    if G == 1 and N % 2 == 0:
        # cross-replica form (dp.enable_sync_batchnorm): two equal shards' moments, gathered and Chan-combined, = the statistics of the
        # whole batch; backward sums of the shards added (the all-reduce) feed the shard-wise apply pass
        Ph = Pp // 2
        allm = guarded(torch.full((2, 2, C), float("nan")))
        for r in range(2):
            sh = guarded(x.detach().view(2, Ph, C)[r].clone())
            mom = guarded(torch.full((2 * C,), float("nan")))
            ws.fill_(float("nan"))
            assert emu.migan_norm_moments(P(sh), P(mom), mom.data_ptr() + 4 * C, 1, Ph, C, P(ws), nb, None) == 0
            allm[r] = mom.view(2, C)
        m2, i2 = guarded(torch.full((C,), float("nan"))), guarded(torch.full((C,), float("nan")))
        assert emu.migan_norm_sync_finalize(P(allm), 2, Ph, P(m2), P(i2), None, None, None, 0.1, 1e-5, C, None) == 0
        assert K._rel(m2, mean) < 1e-5 and K._rel(i2, invstd) < 1e-5, "sync statistics"
        sums = torch.zeros(2 * C)
        parts = []
        for r in range(2):
            sh, gsh = guarded(x.detach().view(2, Ph, C)[r].clone()), guarded(gy.reshape(2, Ph, C)[r].clone())
            s_r = guarded(torch.full((2 * C,), float("nan")))
            ws.fill_(float("nan"))
            assert emu.migan_norm_bwd_sums(P(sh), P(gsh), P(mean), P(invstd), P(gm), P(bt), P(s_r), None, None, 1, Ph, C, act, 0.2, P(ws),
                                           nb, 0, None) == 0
            sums += s_r
            parts.append((sh, gsh))
        sg = guarded(sums.clone())
        for r, (sh, gsh) in enumerate(parts):
            dxr = guarded(torch.full((Ph, C), float("nan")))
            assert emu.migan_norm_bwd_apply(P(sh), P(gsh), P(dxr), P(mean), P(invstd), P(gm), P(bt), P(sg), 1, Ph, C, act, 0.2, Pp, None,
                                            None) == 0
            assert K._rel(dxr, x.grad.view(2, Ph, C)[r]) < 2e-5, ("sync bwd", r, K._rel(dxr, x.grad.view(2, Ph, C)[r]))
    cb = emu.migan_colsum_workspace(N * HW, C)
    cws = guarded(torch.full((max(cb // 4, 4),), float("nan")))
    cs = guarded(torch.full((C,), float("nan")))
    assert emu.migan_colsum(P(gyn), P(cs), N * HW, C, P(cws), cb, 0, None) == 0
    assert K._rel(cs, gy.reshape(-1, C).sum(0)) < 1e-5, "colsum"


def _toeplitz_case_guarded(N, Ci, H, W, Co, k, gather):
    """Image-output conv (cyclegan/models.py:75 ReflectionPad2d(3) + Conv2d(64, 3, 7); srgan/models.py:62 Conv2d(64, 3, 9, 1, 4)) on the
    width-Toeplitz path of csrc/thin_toeplitz.hip: pack, forward (R x 1 GEMM into P + diagonal sum), expansion of dy into Q, weight and
    input gradients from Q - every operand and the P / Q / slab buffers against a guard page, NaN-filled, results against torch."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(9)
    P = K._ptr
    pad = k // 2
    assert emu.migan_thin_toeplitz_ok(Co, k, k, Ci, 1, gather) == 1, (Co, k, Ci, gather)
    x = torch.randn(N, Ci, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, k, k, generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(Co, generator=g)
    y_ref = torch.tanh(TF.conv2d(K._gather_ref(x, (pad,) * 4, gather), w, b, 1))
    gy = torch.randn(y_ref.shape, generator=g)
    y_lin = TF.conv2d(K._gather_ref(x, (pad,) * 4, gather), w, b, 1)
    y_lin.backward(gy)
    Ho, Wo = y_ref.shape[2:]
    cop = emu.migan_thin_toeplitz_cols(Co, k)
    xn, wg, bg = guarded(x.detach().permute(0, 2, 3, 1).contiguous()), guarded(w.detach().clone()), guarded(b)
    wt, wd = guarded(torch.full((cop * k * Ci,), float("nan"))), guarded(torch.full((cop * k * Ci,), float("nan")))
    assert emu.migan_thin_toeplitz_pack(P(wg), P(wt), P(wd), Co, Ci, k, k, None) == 0
    nbp = emu.migan_thin_toeplitz_workspace(N, Ho, W, Co, k)
    pq = guarded(torch.full((nbp // 4,), float("nan")))
    y = guarded(torch.full((N, Ho, Wo, Co), float("nan")))
    assert emu.migan_thin_toeplitz_fwd(P(xn), P(wt), P(bg), P(y), P(pq), nbp, N, H, W, Ci, Ho, Wo, Co, k, k, pad, pad, gather, 3, 0.0,
                                       None) == 0
    assert K._rel(y.permute(0, 3, 1, 2), y_ref.detach()) < 3e-6, "toeplitz fwd"
    gyn = guarded(gy.permute(0, 2, 3, 1).contiguous())
    pq.fill_(float("nan"))
    assert emu.migan_thin_toeplitz_expand(P(gyn), P(pq), N, Ho, Wo, Co, W, k, pad, gather, None) == 0
    nbw = emu.migan_thin_toeplitz_wgrad_workspace(N, Ho, W, Ci, Co, k, k)
    wsw = guarded(torch.full((max(nbw // 4, 4),), float("nan")))
    dw = guarded(torch.full((Co, Ci, k, k), float("nan")))
    assert emu.migan_thin_toeplitz_wgrad(P(xn), P(pq), P(dw), P(wsw), nbw, N, H, W, Ci, Ho, Co, k, k, pad, gather, 0, None) == 0
    assert K._rel(dw, w.grad) < 1e-5, "toeplitz wgrad %g" % K._rel(dw, w.grad)
    nbd = emu.migan_thin_toeplitz_dgrad_workspace(N, H, W, Ci, Ho, k, gather)
    wsd = guarded(torch.full((max(nbd // 4, 4),), float("nan")))
    dx = guarded(torch.full((N, H, W, Ci), float("nan")))
    assert emu.migan_thin_toeplitz_dgrad(P(pq), P(wd), P(dx), P(wsd), nbd, N, H, W, Ci, Ho, Co, k, k, pad, gather, None) == 0
    assert K._rel(dx.permute(0, 3, 1, 2), x.grad) < 3e-6, "toeplitz dgrad %g" % K._rel(dx.permute(0, 3, 1, 2), x.grad)


def _rgb_case_guarded(N, H, W, Co, k, pad, gather, act):
    """Image-input layers (3 source channels: srgan/models.py:85, vgg19.features[0], cyclegan/models.py:49-50) on csrc/rgb_conv.hip: forward
    from staged image rows (the next tile's rows are fetched under the current tile's MFMAs - the fetch that could run past the image) and the
    weight gradient with the activation backward and the bias column sums inside - x, w, bias, dy, y, the slab workspace against guard
    pages, outputs NaN-filled, results against torch."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(13)
    P = K._ptr
    x = torch.randn(N, 3, H, W, generator=g)
    w = (torch.randn(Co, 3, k, k, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True)
    f = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu}[act]
    y_ref = f(TF.conv2d(K._gather_ref(x, (pad,) * 4, gather), w, b, 1))
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    Ho, Wo = y_ref.shape[2:]
    assert emu.migan_rgb_conv_ok(3, Co, k, k, 1, gather, N * Ho * Wo) == 1, (Co, k, gather)
    xn = guarded(x.permute(0, 2, 3, 1).contiguous())
    wk = guarded(w.detach().permute(2, 3, 1, 0).contiguous())
    bg = guarded(b.detach().clone())
    y = guarded(torch.full((N, Ho, Wo, Co), float("nan")))
    assert emu.migan_rgb_conv_fwd(P(xn), P(wk), P(bg), P(y), N, H, W, 3, Ho, Wo, Co, k, k, pad, pad, gather, act, 0.2, 0, None) == 0
    assert K._rel(y.permute(0, 3, 1, 2), y_ref.detach()) < 3e-6, "rgb fwd %g" % K._rel(y.permute(0, 3, 1, 2), y_ref.detach())
    if emu.migan_rgb_conv_wgrad_ok(3, Co, k, k, 1, gather, N * Ho * Wo) == 1:
        nb = emu.migan_rgb_conv_wgrad_workspace(Co, k, k)
        ws = guarded(torch.full((nb // 4,), float("nan")))
        gyn = guarded(gy.permute(0, 2, 3, 1).contiguous())
        dw = guarded(torch.full((Co, 3, k, k), float("nan")))
        db = guarded(torch.full((Co,), float("nan")))
        assert emu.migan_rgb_conv_wgrad(P(xn), P(gyn), P(y) if act else None, P(dw), P(db), P(ws), nb, N, H, W, Ho, Wo, Co, k, k, pad, pad,
                                        gather, act, 0.2, 0, 0, None) == 0
        assert K._rel(dw, w.grad) < 1e-5, "rgb wgrad %g" % K._rel(dw, w.grad)
        assert K._rel(db, b.grad) < 2e-5, "rgb bias gradient %g" % K._rel(db, b.grad)
        # accumulate form: dw += gradient, db += gradient
        assert emu.migan_rgb_conv_wgrad(P(xn), P(gyn), P(y) if act else None, P(dw), P(db), P(ws), nb, N, H, W, Ho, Wo, Co, k, k, pad, pad,
                                        gather, act, 0.2, 1, 1, None) == 0
        assert K._rel(dw, 2 * w.grad) < 1e-5 and K._rel(db, 2 * b.grad) < 2e-5, "rgb wgrad (accumulate)"


def _thin_out_dgrad_case_guarded(N, H, W, c):
    """Input gradient of a thin-OUTPUT 3x3 layer (dcgan.py:62 Conv2d(64, channels, 3, 1, 1)) through migan_rgb_conv_fwd with reversed taps:
    x = dy [N][H][W][c], w = the layer's OIHW weight [c][64][3][3] permuted (2,3,0,1)."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(17)
    P = K._ptr
    dy = torch.randn(N, c, H, W, generator=g)
    w = torch.randn(c, 64, 3, 3, generator=g) * 0.2
    if emu.migan_rgb_conv_ok(c, 64, 3, 3, 1, 0, N * H * W) != 1:
        print("   (not taken: c = %d)" % c, flush=True)
        return
    dx_ref = TF.conv_transpose2d(dy, w, None, 1, 1)
    dyn = guarded(dy.permute(0, 2, 3, 1).contiguous())
    wk = guarded(w.permute(2, 3, 0, 1).contiguous())
    dx = guarded(torch.full((N, H, W, 64), float("nan")))
    assert emu.migan_rgb_conv_fwd(P(dyn), P(wk), None, P(dx), N, H, W, c, H, W, 64, 3, 3, 1, 1, 0, 0, 0.0, 1, None) == 0
    assert K._rel(dx.permute(0, 3, 1, 2), dx_ref) < 3e-6, "thin-output dgrad %g" % K._rel(dx.permute(0, 3, 1, 2), dx_ref)


def _eltwise_case_guarded(N, H, W, C):
    """The elementwise / index kernels of csrc/eltwise.hip on shapes whose element counts are NOT multiples of a vector width: activation
    forward / backward, axpby, MaxPool2d(2) forward / backward / backward-through-ReLU, gather (zero pad, reflection pad, upsample), channel
    concat / split, batched transpose, PixelShuffle - every operand against a guard page, outputs NaN-filled, results against torch."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(13)
    P = K._ptr
    x = torch.randn(N, H, W, C, generator=g)
    xs, n = guarded(x.clone()), x.numel()
    y = guarded(torch.full((N, H, W, C), float("nan")))
    for act, ref in ((1, lambda t: TF.leaky_relu(t, 0.2)), (2, torch.relu), (3, torch.tanh), (4, torch.sigmoid)):
        y.fill_(float("nan"))
        assert emu.migan_act_fwd(P(xs), P(y), n, act, 0.2, None) == 0
        assert K._rel(y, ref(x)) < 2e-6, ("act_fwd", act)
        dy, dx = guarded(torch.randn(N, H, W, C, generator=g)), guarded(torch.full((N, H, W, C), float("nan")))
        assert emu.migan_act_bwd(P(dy), P(y), P(dx), n, act, 0.2, None) == 0
        xr = x.clone().requires_grad_(True)
        ref(xr).backward(dy.clone())
        assert K._rel(dx, xr.grad) < 2e-5, ("act_bwd", act)
    b2, out = guarded(torch.randn(N, H, W, C, generator=g)), guarded(torch.full((N, H, W, C), float("nan")))
    assert emu.migan_axpby(P(xs), 0.5, P(b2), -2.0, P(out), n, None) == 0
    assert K._rel(out, 0.5 * x - 2.0 * b2) < 1e-6, "axpby"
    if H % 2 == 0 and W % 2 == 0:
        xc = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
        pr = TF.max_pool2d(xc, 2)
        gp = torch.randn(pr.shape, generator=g)
        pr.backward(gp)
        py = guarded(torch.full((N, H // 2, W // 2, C), float("nan")))
        assert emu.migan_maxpool2_fwd(P(xs), P(py), N, H, W, C, None) == 0
        assert torch.equal(py.permute(0, 3, 1, 2), pr.detach()), "maxpool fwd"
        gpn, pdx = guarded(gp.permute(0, 2, 3, 1).contiguous()), guarded(torch.full((N, H, W, C), float("nan")))
        assert emu.migan_maxpool2_bwd(P(xs), P(gpn), P(pdx), N, H, W, C, None) == 0
        assert K._rel(pdx.permute(0, 3, 1, 2), xc.grad) < 1e-6, "maxpool bwd"
        xq = x.permute(0, 3, 1, 2).clone().requires_grad_(True)   # the pool behind a ReLU whose backward it applies (srgan/models.py:8-15)
        rq = torch.relu(xq)
        TF.max_pool2d(rq, 2).backward(gp)
        rx = guarded(torch.relu(x).clone())
        pdx.fill_(float("nan"))
        assert emu.migan_maxpool2_relu_bwd(P(rx), P(gpn), P(pdx), N, H, W, C, None) == 0
        assert K._rel(pdx.permute(0, 3, 1, 2), xq.grad) < 1e-6, "maxpool relu bwd"
    for mode, (pt, pl, Ho, Wo), ref in ((0, (1, 2, H + 2, W + 4), lambda t: TF.pad(t, (2, 2, 1, 1))),
                                        (1, (2, 1, H + 4, W + 2), lambda t: TF.pad(t, (1, 1, 2, 2), mode="reflect")),
                                        (2, (0, 0, 2 * H, 2 * W), lambda t: TF.interpolate(t, scale_factor=2, mode="nearest"))):
        gyv = guarded(torch.full((N, Ho, Wo, C), float("nan")))
        assert emu.migan_gather2d_fwd(P(xs), P(gyv), N, H, W, C, Ho, Wo, pt, pl, mode, None) == 0
        xc = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
        r = ref(xc)
        assert torch.equal(gyv.permute(0, 3, 1, 2), r.detach()), ("gather fwd", mode)
        gg = torch.randn(r.shape, generator=g)
        r.backward(gg)
        ggn, gdx = guarded(gg.permute(0, 2, 3, 1).contiguous()), guarded(torch.full((N, H, W, C), float("nan")))
        assert emu.migan_gather2d_bwd(P(ggn), P(gdx), N, H, W, C, Ho, Wo, pt, pl, mode, None) == 0
        assert K._rel(gdx.permute(0, 3, 1, 2), xc.grad) < 1e-6, ("gather bwd", mode)
    Cb = C + 3
    bb = guarded(torch.randn(N, H, W, Cb, generator=g))
    cy = guarded(torch.full((N, H, W, C + Cb), float("nan")))
    assert emu.migan_cat_channels(P(xs), P(bb), P(cy), N * H * W, C, Cb, 1, None) == 0
    assert torch.equal(cy, torch.cat([x, bb], 3)), "cat"
    da, db = guarded(torch.full((N, H, W, C), float("nan"))), guarded(torch.full((N, H, W, Cb), float("nan")))
    assert emu.migan_cat_channels(P(da), P(db), P(cy), N * H * W, C, Cb, 0, None) == 0
    assert torch.equal(da, x) and torch.equal(db, bb), "split"
    ty = guarded(torch.full((N, C, H * W), float("nan")))
    assert emu.migan_transpose_batched(P(xs), P(ty), N, H * W, C, None) == 0
    assert torch.equal(ty, x.view(N, H * W, C).transpose(1, 2)), "transpose"
    if C % 4 == 0:
        sy = guarded(torch.full((N, 2 * H, 2 * W, C // 4), float("nan")))
        assert emu.migan_pixel_shuffle(P(xs), P(sy), N, H, W, C // 4, 2, 1, None) == 0   # C = channels of the shuffled tensor
        assert torch.equal(sy.permute(0, 3, 1, 2), TF.pixel_shuffle(x.permute(0, 3, 1, 2), 2)), "pixel shuffle"
        ux = guarded(torch.full((N, H, W, C), float("nan")))
        assert emu.migan_pixel_shuffle(P(sy), P(ux), N, H, W, C // 4, 2, 0, None) == 0
        assert torch.equal(ux, x), "pixel unshuffle"


def _loss_case_guarded(n, B, D):
    """The mean-reduced losses of csrc/reduce_loss_adam.hip (BCE, MSE, L1, mean, BCE-with-logits; tensor and constant targets) with their
    gradients on n elements, and the per-row norm / row scaling of the gradient penalty on a B x D matrix - every operand and the reduction
    workspace against a guard page, NaN-filled, results against torch."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(17)
    P = K._ptr
    nb = emu.migan_reduce_workspace()
    ws = guarded(torch.full((max(nb // 4, 4),), float("nan")))
    prob, tgt = torch.rand(n, generator=g) * 0.98 + 0.01, torch.rand(n, generator=g)
    raw = torch.randn(n, generator=g)
    refs = {0: lambda x, t: TF.binary_cross_entropy(x, t), 1: lambda x, t: TF.mse_loss(x, t), 2: lambda x, t: TF.l1_loss(x, t),
            3: lambda x, t: x.mean(), 4: lambda x, t: TF.binary_cross_entropy_with_logits(x, t)}
    for kind in (0, 1, 2, 3, 4):
        for const in (False, True):
            if kind == 3 and const:
                continue
            xin = (prob if kind == 0 else raw).clone().requires_grad_(True)
            t = torch.full((n,), 1.0) if const else tgt
            ref = refs[kind](xin, t)
            ref.backward()
            xs = guarded(xin.detach().clone())
            tg = None if (const or kind == 3) else guarded(t.clone())
            out = guarded(torch.full((4,), float("nan")))
            ws.fill_(float("nan"))
            assert emu.migan_loss_fwd(kind, P(xs), P(tg), 1.0, P(out), n, P(ws), nb, None) == 0
            rv = float(ref.detach())
            assert abs(float(out[0]) - rv) <= 2e-6 * max(1.0, abs(rv)), ("loss", kind, const, float(out[0]), rv)
            one, dx = guarded(torch.ones(4)), guarded(torch.full((n,), float("nan")))
            assert emu.migan_loss_bwd(kind, P(xs), P(tg), 1.0, P(one), P(dx), n, None) == 0
            assert K._rel(dx, xin.grad) < 2e-5, ("loss bwd", kind, const)
    x = torch.randn(B, D, generator=g, requires_grad=True)
    nr = x.norm(2, dim=1)
    gn = torch.randn(B, generator=g)
    nr.backward(gn)
    xs, out = guarded(x.detach().clone()), guarded(torch.full((B,), float("nan")))
    assert emu.migan_rownorm_fwd(P(xs), P(out), B, D, None) == 0
    assert K._rel(out, nr.detach()) < 2e-6, "rownorm"
    gng, dx = guarded(gn.clone()), guarded(torch.full((B, D), float("nan")))
    assert emu.migan_rownorm_bwd(P(xs), P(out), P(gng), P(dx), B, D, None) == 0
    assert K._rel(dx, x.grad) < 2e-5, "rownorm bwd"
    sc, y = guarded(torch.rand(B, generator=g)), guarded(torch.full((B, D), float("nan")))
    assert emu.migan_rowscale(P(xs), P(sc), P(y), B, D, None) == 0
    assert torch.equal(y, x.detach() * sc[:, None]), "rowscale"


def _adam_case_guarded(sizes):
    """migan_adam_step (one launch for all tensors of an optimiser; 16-byte accesses on whole 4096-element chunks, scalar tails) on
    parameters of the given sizes, each its OWN guarded allocation, with the gradient / moment buckets in optim.bucket_layout - two steps
    against torch.optim.Adam; the step counter advances by itself and the ticket returns to zero."""
    from pytorch_gan_amd import optim
    g = torch.Generator().manual_seed(19)
    P = K._ptr
    offs, total = optim.bucket_layout(sizes)
    ps = [guarded(torch.randn(n, generator=g)) for n in sizes]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    topt = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
    grad, m, v = guarded(torch.zeros(total)), guarded(torch.zeros(total)), guarded(torch.zeros(total))
    step, ticket, lr = guarded(torch.zeros(4)), guarded(torch.zeros(4)), guarded(torch.full((4,), 2e-4))
    chunk = emu.migan_adam_chunk()
    tab = np.zeros(len(sizes), dtype=optim._ADAM_T)
    blks = []
    for i, (p, o, n) in enumerate(zip(ps, offs, sizes)):
        tab[i] = (p.data_ptr(), grad.data_ptr() + 4 * o, m.data_ptr() + 4 * o, v.data_ptr() + 4 * o, n)
        blks += [(i, c) for c in range((n + chunk - 1) // chunk)]
    blk = np.array(blks, dtype=optim._BLK_T)
    tabg = guarded(torch.from_numpy(np.frombuffer(tab.tobytes(), dtype=np.float32).copy()))
    blkg = guarded(torch.from_numpy(np.frombuffer(blk.tobytes(), dtype=np.float32).copy()))
    for it in range(2):
        for r, o, n in zip(ref, offs, sizes):
            gr = torch.randn(n, generator=g)
            r.grad = gr.clone()
            grad[o:o + n] = gr
        topt.step()
        assert emu.migan_adam_step(P(tabg), P(blkg), len(blks), P(step), P(ticket), P(lr), 2e-4, 0.5, 0.999, 1e-8, 1.0, None) == 0
        assert float(step[0]) == it + 1 and int(ticket.view(torch.int32)[0]) == 0, (float(step[0]), ticket)
        for p, r in zip(ps, ref):
            assert K._rel(p, r.detach()) < 1e-6, ("adam", it, p.numel(), K._rel(p, r.detach()))


def _skinny_case_guarded(M, N, Kk):
    """The <= 64-row GEMMs of csrc/skinny_mm.hip behind nn.Linear at small batch (wgan_gp.py:46-78): forward NT with bias + LeakyReLU,
    input gradient NN, weight / bias gradient TN - row counts that are no multiple of the 16-row MFMA tile included (the kernels clamp
    the row index) - every operand against a guard page, outputs NaN-filled, results against torch."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(23)
    P = K._ptr
    x = torch.randn(M, Kk, generator=g, requires_grad=True)
    w = (torch.randn(N, Kk, generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(N, generator=g, requires_grad=True)
    y_lin = TF.linear(x, w, b)
    gy = torch.randn(M, N, generator=g)
    y_lin.backward(gy)
    xg, wg, bg, gyg = guarded(x.detach().clone()), guarded(w.detach().clone()), guarded(b.detach().clone()), guarded(gy.clone())
    if emu.migan_skinny_nt_ok(M, N, Kk):
        y = guarded(torch.full((M, N), float("nan")))
        assert emu.migan_skinny_nt(P(xg), P(wg), P(bg), P(y), M, N, Kk, 1, 0.2, None) == 0
        assert K._rel(y, TF.leaky_relu(y_lin.detach(), 0.2)) < 3e-6, "skinny nt"
    if emu.migan_skinny_nn_ok(M, N, Kk):
        dx = guarded(torch.full((M, Kk), float("nan")))
        assert emu.migan_skinny_nn(P(gyg), P(wg), P(dx), M, N, Kk, None) == 0
        assert K._rel(dx, x.grad) < 3e-6, "skinny nn"
    if emu.migan_skinny_tn_ok(M, N, Kk):
        dw, db = guarded(torch.full((N, Kk), float("nan"))), guarded(torch.full((N,), float("nan")))
        assert emu.migan_skinny_tn(P(gyg), P(xg), P(dw), P(db), M, N, Kk, 0, 0, None) == 0
        assert K._rel(dw, w.grad) < 1e-5 and K._rel(db, b.grad) < 1e-5, "skinny tn"


def _norm_prelu_case_guarded(N, H, W, C, shuffle):
    """BatchNorm2d [PixelShuffle(2)] PReLU of srgan/models.py:23-24,55-57 inside the norm launches (apply with the single PReLU slope and
    the shuffle as the store index map; backward with the slope's gradient): every operand against a guard page, outputs and workspaces
    NaN-filled, results against torch."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(29)
    P = K._ptr
    Pp = N * H * W
    x = (torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3).requires_grad_(True)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).requires_grad_(True), torch.randn(C, generator=g).requires_grad_(True)
    pw = torch.full((1,), 0.25, requires_grad=True)
    yb = TF.batch_norm(x, None, None, gamma, beta, True, 0.1, 0.8)
    y_ref = TF.prelu(TF.pixel_shuffle(yb, 2) if shuffle else yb, pw)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xs = guarded(x.detach().permute(0, 2, 3, 1).contiguous())
    gm, bt, pg = guarded(gamma.detach().clone()), guarded(beta.detach().clone()), guarded(pw.detach().clone())
    mean, invstd = guarded(torch.full((C,), float("nan"))), guarded(torch.full((C,), float("nan")))
    nb = emu.migan_norm_workspace(1, Pp, C)
    ws = guarded(torch.full((max(nb // 4, 4),), float("nan")))
    assert emu.migan_norm_stats(P(xs), P(mean), P(invstd), None, None, None, 0.1, 0.8, 1, Pp, C, P(ws), nb, None) == 0
    sh, sw = (H, W) if shuffle else (0, 0)
    y = guarded(torch.full(tuple(y_ref.permute(0, 2, 3, 1).shape), float("nan")))
    assert emu.migan_norm_apply_prelu(P(xs), P(y), P(mean), P(invstd), P(gm), P(bt), None, P(pg), 1, Pp, C, sh, sw, None) == 0
    assert K._rel(y.permute(0, 3, 1, 2), y_ref.detach()) < 3e-6, ("apply_prelu", K._rel(y.permute(0, 3, 1, 2), y_ref.detach()))
    nbp = emu.migan_norm_workspace_prelu(1, Pp, C)
    wsp = guarded(torch.full((max(nbp // 4, 4),), float("nan")))
    gyn = guarded(gy.permute(0, 2, 3, 1).contiguous())
    dx = guarded(torch.full((N, H, W, C), float("nan")))
    dg, dbt, dp = guarded(torch.full((C,), float("nan"))), guarded(torch.full((C,), float("nan"))), guarded(torch.full((4,), float("nan")))
    assert emu.migan_norm_bwd_prelu(P(xs), P(gyn), P(mean), P(invstd), P(gm), P(bt), P(pg), P(dx), P(dg), P(dbt), P(dp), 1, Pp, C, P(wsp),
                                    nbp, 0, 0, None, sh, sw, None) == 0
    assert K._rel(dx.permute(0, 3, 1, 2), x.grad) < 2e-5, ("bwd_prelu dx", K._rel(dx.permute(0, 3, 1, 2), x.grad))
    assert K._rel(dg, gamma.grad) < 2e-5 and K._rel(dbt, beta.grad) < 2e-5, "bwd_prelu dgamma / dbeta"
    assert abs(float(dp[0]) - float(pw.grad)) <= 2e-5 * max(1.0, abs(float(pw.grad))), ("dprelu", float(dp[0]), float(pw.grad))


def _guarded_i64(t):
    buf = guarded(torch.zeros(2 * t.numel()))
    out = buf.view(torch.int64)
    out.copy_(t.reshape(-1))
    return out


def _classify_case_guarded(B, C, D):
    """csrc/classify.hip behind the DCGAN-block clones (acgan / cgan / infogan ...): softmax and cross entropy over B x C logits with their
    gradients, the label embedding (V = C rows of D floats) with its scatter-add gradient; plus a weight pack (migan_permute4d) and a
    dropout-mask draw - every operand against a guard page, outputs NaN-filled, results against torch."""
    import torch.nn.functional as TF
    g = torch.Generator().manual_seed(31)
    P = K._ptr
    x = torch.randn(B, C, generator=g, requires_grad=True)
    t = torch.randint(0, C, (B,), generator=g)
    sm = TF.softmax(x, 1)
    gs = torch.randn(B, C, generator=g)
    sm.backward(gs)
    xs, y = guarded(x.detach().clone()), guarded(torch.full((B, C), float("nan")))
    assert emu.migan_softmax_fwd(P(xs), P(y), B, C, None) == 0
    assert K._rel(y, sm.detach()) < 2e-6, "softmax"
    gsg, dx = guarded(gs.clone()), guarded(torch.full((B, C), float("nan")))
    assert emu.migan_softmax_bwd(P(y), P(gsg), P(dx), B, C, None) == 0
    assert K._rel(dx, x.grad) < 2e-5, "softmax bwd"
    x2 = x.detach().clone().requires_grad_(True)
    ce = TF.cross_entropy(x2, t)
    ce.backward()
    tg = _guarded_i64(t)
    out, ws = guarded(torch.full((4,), float("nan"))), guarded(torch.full((2 * B,), float("nan")))
    assert emu.migan_cross_entropy_fwd(P(xs), tg.data_ptr(), P(out), P(ws), B, C, None) == 0
    assert abs(float(out[0]) - float(ce.detach())) <= 2e-6 * max(1.0, abs(float(ce.detach()))), "cross entropy"
    one, dce = guarded(torch.ones(4)), guarded(torch.full((B, C), float("nan")))
    assert emu.migan_cross_entropy_bwd(P(xs), tg.data_ptr(), ws.data_ptr() + 4 * B, P(one), P(dce), B, C, None) == 0
    assert K._rel(dce, x2.grad) < 2e-5, "cross entropy bwd"
    w = torch.randn(C, D, generator=g, requires_grad=True)
    e = TF.embedding(t, w)
    ge = torch.randn(B, D, generator=g)
    e.backward(ge)
    wg, ey = guarded(w.detach().clone()), guarded(torch.full((B, D), float("nan")))
    assert emu.migan_embedding_fwd(P(wg), tg.data_ptr(), P(ey), B, D, C, None) == 0
    assert torch.equal(ey, e.detach()), "embedding"
    geg, dw = guarded(ge.clone()), guarded(torch.full((C, D), float("nan")))
    assert emu.migan_embedding_bwd(P(geg), tg.data_ptr(), P(dw), B, D, C, 0, None) == 0
    assert K._rel(dw, w.grad) < 1e-5, "embedding bwd"
    d = (C, D, 3, 3)
    src = torch.randn(*d, generator=g)
    sg = guarded(src.clone())
    for perm in ((0, 2, 3, 1), (1, 2, 3, 0)):
        dst = guarded(torch.full(tuple(d[q] for q in perm), float("nan")))
        assert emu.migan_permute4d(P(sg), P(dst), d[0], d[1], d[2], d[3], perm[0], perm[1], perm[2], perm[3], None) == 0
        assert torch.equal(dst, src.permute(*perm).contiguous()), ("permute4d", perm)
    n = B * C + 3
    mk = guarded(torch.full((n,), float("nan")))
    assert emu.migan_rand_mask(P(mk), n, 0.25, 1234, None, None) == 0
    vals = set(mk.unique().tolist())
    assert vals <= {0.0, 1.0 / 0.75} or all(abs(v) < 1e-6 or abs(v - 1 / 0.75) < 1e-6 for v in vals), vals


cases = K._gpu_conv_cases() + K.KTAIL_CASES
if len(sys.argv) > 1 and sys.argv[1] == "classify":
    for c in [(1, 2, 3), (7, 10, 100), (64, 10, 62), (33, 101, 17), (128, 1000, 5)]:
        print("classify", c, flush=True)
        _classify_case_guarded(*c)
        keep.clear()
    print("ALL OK")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "norm_prelu":
    for c in [(2, 6, 6, 64, False), (1, 5, 7, 12, False), (2, 4, 4, 256, True), (1, 3, 5, 8, True), (4, 24, 24, 64, False), (2, 12, 12, 256, True)]:
        print("norm_prelu", c, flush=True)
        _norm_prelu_case_guarded(*c)
        keep.clear()
    print("ALL OK")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "skinny":
    for c in [(1, 16, 64), (7, 32, 128), (64, 512, 1024), (33, 256, 512), (64, 16, 256), (5, 1024, 1024), (64, 1024, 128), (17, 48, 192)]:
        print("skinny", c, flush=True)
        _skinny_case_guarded(*c)
        keep.clear()
    print("ALL OK")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "adam":
    for c in [[1], [3, 64, 1], [4096], [4097, 5], [8192 + 4, 100, 4096 * 3], [128 * 100, 1, 64 * 3 * 9, 3]]:
        print("adam", c, flush=True)
        _adam_case_guarded(c)
        keep.clear()
    print("ALL OK")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "loss":
    for c in [(1, 1, 1), (7, 3, 5), (64, 64, 1024), (1000, 5, 333), (128 * 25, 64, 3072), (65537, 9, 1027)]:
        print("loss", c, flush=True)
        _loss_case_guarded(*c)
        keep.clear()
    print("ALL OK")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "eltwise":
    for c in [(1, 3, 5, 1), (2, 6, 10, 3), (1, 8, 8, 4), (3, 7, 9, 5), (2, 12, 4, 8), (1, 16, 18, 66), (1, 30, 30, 64)]:
        print("eltwise", c, flush=True)
        _eltwise_case_guarded(*c)
        keep.clear()
    print("ALL OK")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "rgb":
    # (the entries take >= 16384 output pixels; widths around the 128-pixel row tile: ragged, exact, a 5-pixel image of tails only)
    for c in [(1, 110, 150, 64, 3, 1, 0, 1), (2, 64, 130, 32, 3, 1, 0, 0), (1, 128, 131, 64, 7, 3, 1, 2), (1, 3300, 5, 64, 3, 1, 0, 2),
              (2, 64, 128, 64, 3, 1, 0, 2)]:
        print("rgb", c, flush=True)
        _rgb_case_guarded(*c)
        keep.clear()
    for c in [(2, 64, 130, 1), (1, 130, 127, 3), (1, 128, 128, 1)]:
        print("thin-output dgrad", c, flush=True)
        _thin_out_dgrad_case_guarded(*c)
        keep.clear()
    print("ALL OK")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "toeplitz":
    for c in [(1, 64, 20, 24, 3, 7, 1), (2, 64, 16, 16, 3, 9, 0), (1, 16, 13, 17, 3, 7, 0), (1, 32, 24, 40, 2, 9, 1), (3, 64, 11, 9, 4, 7, 1),
              (1, 64, 40, 32, 3, 7, 1)]:   # the last: strip-walking weight gradient, two row segments, mirrored virtual rows
        print("toeplitz", c, flush=True)
        _toeplitz_case_guarded(*c)
        keep.clear()
    print("ALL OK")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "upconv":
    # split counts 1 ... 20 (N x 8 x 8 source pixels): every round width and remainder of the four-class reduction
    for c in [(1, 16, 8, 8, 16), (3, 16, 8, 8, 16), (5, 16, 8, 8, 32), (9, 32, 8, 8, 16), (19, 16, 8, 8, 16), (40, 16, 8, 8, 16), (80, 16, 8, 8, 16),
              (2, 128, 6, 6, 64)]:
        print("upconv", c, flush=True)
        _upconv_case_guarded(*c)
        keep.clear()
    print("ALL OK")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "norm":
    # chunk counts from 1 to > 256 per (group, channel): every remainder of the four-records-per-round finalize loops
    for c in [(1, 2, 6, 8, 0), (1, 4, 64, 16, 1), (1, 8, 1000, 64, 1), (1, 16, 4096, 128, 0), (1, 3, 5000, 12, 1), (4, 4, 300, 32, 1),
              (2, 2, 4096, 64, 0), (1, 5, 777, 3, 0)]:
        print("norm", c, flush=True)
        _norm_case_guarded(*c)
        keep.clear()
    print("ALL OK")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "fewpix":
    for c in [(1, 256, 2, 2, 512), (1, 64, 16, 16, 1024), (3, 128, 4, 6, 512), (1, 512, 8, 8, 128), (2, 1024, 2, 2, 256)]:
        print("fewpix", c, flush=True)
        _fewpix_case_guarded(*c)
        keep.clear()
    print("ALL OK")
    sys.exit(0)
i0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for i, c in enumerate(cases):
    if i < i0: continue
    print(i, c, flush=True)
    _conv_case_guarded(c)
    keep.clear()
print("ALL OK")
