"""Kernel LOGIC without a GPU: the unchanged HIP sources of pytorch-gan_amd/csrc compiled for the host against the execution
model in tests/hipemu (one fiber per HIP thread, 64-lane waves, __syncthreads, wave shuffles, the fp32 MFMA fragment layouts
as exact fmaf chains, LDS-DMA with the buffer descriptor's range check, split-K tickets) and driven

  * through the C ABI of include/migan.h against torch's CPU ops, and
  * through the product's own Python host mirror (autograd Functions, modules, fused Adam, the training-step bodies of
    steps.py) - by running the BODIES OF THE GPU PARITY TESTS of test_steps_gpu.py on CPU tensors against the oracle.

This is a checker, never a product path: the product has no CPU mode, the patching lives in tests/hipemu/host.py.  What the
model cannot see (timing, missing waits on asynchronous loads, memory ordering) stays with the `-m gpu` tests; what it does
see it sees strictly - e.g. an LDS-DMA lane that passes the descriptor's range check and still leaves the tensor fails the
launch here (the hardware would silently read the neighbouring allocation)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as TF

import hipemu

pytestmark = pytest.mark.skipif(not hipemu.available(), reason="no host clang++ / not x86-64: the execution model cannot be built")


_BUILD_ERROR = []


def _load_or_skip():
    """A host toolchain problem (the model is compiled with the container's clang++) must not read as a kernel failure."""
    if _BUILD_ERROR:
        pytest.skip(_BUILD_ERROR[0])
    try:
        return hipemu.load()
    except Exception as e:  # noqa: BLE001
        _BUILD_ERROR.append("the execution model could not be built here: %s" % str(e)[:300])
        pytest.skip(_BUILD_ERROR[0])


@pytest.fixture(scope="module")
def emu():
    lib = _load_or_skip()
    lib.hipemu_reset_counts()
    return lib


def _rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))


def _ptr(t):
    return None if t is None else t.data_ptr()


def _gather_ref(x, pads, gather):
    t, l, b, r = pads
    if gather == 2:
        x = TF.interpolate(x, scale_factor=2, mode="nearest")
    if gather == 1:
        return TF.pad(x, (l, r, t, b), mode="reflect")
    return TF.pad(x, (l, r, t, b))


def _conv_case(emu, case, sk):
    """forward / input gradient / weight gradient of one geometry through migan_conv2d_{fwd_ws,dgrad_ws,wgrad} -> rel. errors"""
    N, Ci, H, W, Co, k, stride, pads, gather, act, bias = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Ci, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, k, k, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(Co, generator=g) if bias else None
    y_lin = TF.conv2d(_gather_ref(x, pads, gather), w, b, stride)
    y_ref = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu, 3: torch.tanh}[act](y_lin)
    Ho, Wo = y_ref.shape[2:]
    out = {}
    xn = x.detach().permute(0, 2, 3, 1).contiguous()
    y = torch.empty(N, Ho, Wo, Co)
    w_ohwi, w_ihwo = w.detach().permute(0, 2, 3, 1).contiguous(), w.detach().permute(1, 2, 3, 0).contiguous()
    rc = emu.migan_conv2d_fwd_ws(_ptr(xn), _ptr(w_ohwi), _ptr(b), None, _ptr(y), N, H, W, Ci,
                                 Ho, Wo, Co, k, k, stride, pads[0], pads[1], gather, act, 0.2, _ptr(sk), sk.numel() * 4, None)
    assert rc == 0, (case, "fwd", emu.hipemu_last_message())
    out["fwd"] = _rel(y.permute(0, 3, 1, 2), y_ref.detach())
    if gather == 0:
        gy = torch.randn(y_lin.shape, generator=g)
        y_lin.backward(gy)
        gyn = gy.permute(0, 2, 3, 1).contiguous()
        if pads[0] == pads[2] and pads[1] == pads[3]:
            dx = torch.empty(N, H, W, Ci)
            rc = emu.migan_conv2d_dgrad_ws(_ptr(gyn), _ptr(w_ihwo), None, _ptr(dx), N, H, W, Ci,
                                           Ho, Wo, Co, k, k, stride, pads[0], pads[1], 0, 0.0, _ptr(sk), sk.numel() * 4, None)
            assert rc == 0, (case, "dgrad", emu.hipemu_last_message())
            out["dgrad"] = _rel(dx.permute(0, 3, 1, 2), x.grad)
        wsb = emu.migan_conv2d_wgrad_workspace(N, Ho, Wo, Co, k, k, Ci)
        ws = torch.empty(max(wsb // 4, 1))
        dw = torch.empty(Co, Ci, k, k)
        rc = emu.migan_conv2d_wgrad(_ptr(xn), _ptr(gyn), _ptr(dw), _ptr(ws), wsb, N, H, W, Ci, Ho, Wo, Co, k, k, stride, pads[0],
                                    pads[1], gather, 0, None, 0, None, 0, None)
        assert rc == 0, (case, "wgrad", emu.hipemu_last_message())
        out["wgrad"] = _rel(dw, w.grad)
    return out


def _gpu_conv_cases():
    from test_ops_gpu import CONV_CASES

    return [c for c in CONV_CASES if c[0] * c[2] * c[3] <= 100000]   # all but the 401k-pixel case and the 147k-pixel trunk


def test_every_gpu_conv_case_on_the_execution_model(emu):
    """The CONV_CASES of test_ops_gpu.py (the GPU suite's list: every kernel family of conv_igemm.hip / conv_dma.hip - LDS-DMA
    tiles, tap-inner up-conv, K-tails, split-K with tickets, thin-N, small-K, gemv, parity-class dgrads, DMA / incremental /
    thin wgrads) against torch on the host, same tolerances as on the GPU."""
    sk = torch.zeros(emu.migan_conv_splitk_workspace() // 4)
    emu.hipemu_reset_counts()
    for case in _gpu_conv_cases():
        for what, err in _conv_case(emu, case, sk).items():
            assert err <= (1e-5 if what == "wgrad" else 3e-6), (case, what, err)
        assert int(sk[:1024].view(torch.int32).abs().sum()) == 0, ("split-K tickets not back at zero", case)
    # the list really went through the kernels it is meant for
    for sym in ("igemm_dma_kernel", "wgrad_dma_kernel", "thin_conv_kernel", "thin_conv_wave_kernel", "smallk", "midk_tile_kernel", "wgrad_reduce"):
        assert emu.hipemu_launch_count(sym.encode()) > 0, sym


def test_generic_kernels_serve_channel_counts_that_are_no_multiple_of_four(emu):
    """igemm_kernel / wgrad_kernel (csrc/conv_igemm.hip: scalar gathers, any channel count) are what runs when the source channels are
    not a multiple of 4 and the small-K / thin kernels do not take the shape: forward, input gradient (its GEMM's channels are Co) and
    weight gradient against torch, with the launch counters saying which kernel served."""
    sk = torch.zeros(emu.migan_conv_splitk_workspace() // 4)
    for case in [(2, 6, 12, 12, 40, 3, 1, (1, 1, 1, 1), 0, 1, True), (3, 10, 9, 9, 72, 3, 2, (1, 1, 1, 1), 0, 0, True),
                 (2, 7, 10, 10, 130, 5, 1, (2, 2, 2, 2), 0, 2, False)]:
        emu.hipemu_reset_counts()
        for what, err in _conv_case(emu, case, sk).items():
            assert err <= (1e-5 if what == "wgrad" else 3e-6), (case, what, err)
        assert emu.hipemu_launch_count(b"igemm_kernel<") >= 1 and emu.hipemu_launch_count(b"wgrad_kernel<") == 1, case


def test_split_count_sweep_of_the_weight_gradient_reductions(emu):
    """The slab reductions behind the split-K weight gradients load up to eight slabs per round with the round width following the
    slabs that are left (csrc/conv_igemm.hip slab_sum / upconv_slab_round: 8 / 4 / 2 / 1, clamped index, guarded add).  Sweep the split
    count over every width and remainder - 1 ... 12 on one split lane, 17 / 20 / 33 on four, 70 on sixteen; the phase-collapsed
    up-conv form over 1 ... 20 - and compare with torch."""
    seen, seen_up = set(), set()
    Co = Ci = 16
    for N in (1, 2, 3, 4, 5, 6, 7, 9, 12, 17, 20, 33, 70):
        g = torch.Generator().manual_seed(N)
        x = torch.randn(N, Ci, 16, 16, generator=g)
        w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.2).requires_grad_(True)
        y = TF.conv2d(x, w, None, 1, 1)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        xn, gyn = x.permute(0, 2, 3, 1).contiguous(), gy.permute(0, 2, 3, 1).contiguous()
        wsb = emu.migan_conv2d_wgrad_workspace(N, 16, 16, Co, 3, 3, Ci)
        ws, dw = torch.empty(wsb // 4), torch.empty(Co, Ci, 3, 3)
        rc = emu.migan_conv2d_wgrad(_ptr(xn), _ptr(gyn), _ptr(dw), _ptr(ws), wsb, N, 16, 16, Ci, 16, 16, Co, 3, 3, 1, 1, 1, 0, 0, None,
                                    0, None, 0, None)
        assert rc == 0, (N, emu.hipemu_last_message())
        assert _rel(dw, w.grad) <= 2e-6, (N, _rel(dw, w.grad))
        seen.add(wsb // (Co * Ci * 9 * 4))
    assert seen >= {1, 2, 3, 4, 5, 6, 7, 9, 12, 17, 20, 33, 70}, seen
    for N in (1, 2, 3, 5, 6, 9, 13, 19, 40, 80):
        g = torch.Generator().manual_seed(100 + N)
        x = torch.randn(N, Ci, 8, 8, generator=g)
        w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.2).requires_grad_(True)
        y = TF.conv2d(TF.interpolate(x, scale_factor=2, mode="nearest"), w, None, 1, 1)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        xn, gyn = x.permute(0, 2, 3, 1).contiguous(), gy.permute(0, 2, 3, 1).contiguous()
        wsb = emu.migan_upconv3x3_wgrad_workspace(N, 8, 8, Co, Ci)
        ws, dw = torch.empty(wsb // 4), torch.empty(Co, Ci, 3, 3)
        rc = emu.migan_upconv3x3_wgrad(_ptr(xn), _ptr(gyn), _ptr(dw), _ptr(ws), wsb, N, 8, 8, Ci, Co, 0, None, 0, None, 0, None)
        assert rc == 0, (N, emu.hipemu_last_message())
        assert _rel(dw, w.grad) <= 2e-6, (N, _rel(dw, w.grad))
        seen_up.add(wsb // (4 * Co * (4 * Ci + 1) * 4))   # 4 classes x splits x (Co x 4 Ci partials + Co bias slab)
    assert len(seen_up) >= 6 and max(seen_up) >= 17, seen_up   # 17+: a whole round of eight on every split lane
    # 32 source channels: the form that reads every slab element once (upconv_wgrad_reduce2_kernel: 8 split lanes, two splits per lane and
    # round); 10 / 33 / 75 splits leave lanes with one split of a pair or none
    Ci, seen32 = 32, set()
    emu.hipemu_reset_counts()
    for N in (40, 132, 160, 300):
        g = torch.Generator().manual_seed(200 + N)
        x = torch.randn(N, Ci, 8, 8, generator=g)
        w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.2).requires_grad_(True)
        y = TF.conv2d(TF.interpolate(x, scale_factor=2, mode="nearest"), w, None, 1, 1)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        xn, gyn = x.permute(0, 2, 3, 1).contiguous(), gy.permute(0, 2, 3, 1).contiguous()
        wsb = emu.migan_upconv3x3_wgrad_workspace(N, 8, 8, Co, Ci)
        ws, dw = torch.empty(wsb // 4), torch.full((Co, Ci, 3, 3), 0.5)
        rc = emu.migan_upconv3x3_wgrad(_ptr(xn), _ptr(gyn), _ptr(dw), _ptr(ws), wsb, N, 8, 8, Ci, Co, 1, None, 0, None, 0, None)   # accumulate
        assert rc == 0, (N, emu.hipemu_last_message())
        assert _rel(dw - 0.5, w.grad) <= 2e-6, (N, _rel(dw - 0.5, w.grad))
        seen32.add(wsb // (4 * Co * (4 * Ci + 1) * 4))
    assert min(seen32) < 32 <= max(seen32) and any(s_ % 32 for s_ in seen32 if s_ > 32), seen32
    assert emu.hipemu_launch_count(b"upconv_wgrad_reduce2_kernel") == 4


def test_results_do_not_depend_on_the_wave_schedule(emu):
    """A model that switches fibers only at synchronisation points runs the waves of a workgroup in ONE order between two
    barriers: a missing __syncthreads (two waves touching the same LDS / global word with no barrier between them) is invisible in
    that order - and changes the result in another.  Every conv case (all kernel families), the specialised-kernel cases of
    tests/kernel_cases.py (small-tensor InstanceNorm, few-pixel conv path, PatchGAN heads, packs) and the fused critic kernel: waves in index order, in
    reverse order and in a seeded random order per scheduler pass must give bit-identical results."""
    import hipemu.host
    import kernel_cases

    sk = torch.zeros(emu.migan_conv_splitk_workspace() // 4)
    cases = _gpu_conv_cases() + KTAIL_CASES

    def conv_pass():
        return [sorted(_conv_case(emu, c, sk).items()) for c in cases]

    def mirror_pass():
        from pytorch_gan_amd import functional as F

        out = {}
        with hipemu.host.emulated_device():
            saved = kernel_cases.quiet_scope(F)
            try:
                with torch.enable_grad():
                    for name, _syms, inputs, run, _ref in kernel_cases.all_cases():
                        gen = torch.Generator(device="cpu")
                        gen.manual_seed(99)
                        out[name] = {k: v.clone() for k, v in run(F, inputs(gen)).items()}
            finally:
                kernel_cases.restore_scope(F, saved)
        return out

    try:
        emu.hipemu_set_wave_schedule(0, 1)
        conv_ref, mirror_ref = conv_pass(), mirror_pass()
        for mode, seed in ((1, 1), (2, 7)):
            emu.hipemu_set_wave_schedule(mode, seed)
            got = conv_pass()
            for c, a, b in zip(cases, conv_ref, got):
                assert a == b, ("wave schedule changes a conv result", mode, c, a, b)
            if mode == 1:
                m = mirror_pass()
                for name, tensors in mirror_ref.items():
                    for k, v in tensors.items():
                        assert torch.equal(v, m[name][k]), ("wave schedule changes a result", name, k)
    finally:
        emu.hipemu_set_wave_schedule(0, 1)


KTAIL_CASES = [
    (2, 40, 10, 10, 72, 3, 1, (1, 1, 1, 1), 0, 1, True),     # Ci = 40: one full K-tile + a tail of 8 channels
    (2, 100, 6, 6, 36, 3, 1, (1, 1, 1, 1), 0, 0, False),     # Ci = 100
    (3, 72, 8, 8, 96, 3, 1, (1, 1, 1, 1), 0, 0, True),       # split-K with a K-tail
    (3, 32, 9, 7, 48, 3, 1, (1, 1, 1, 1), 0, 2, True),       # dgrad GEMM K = Co = 48
    (1, 36, 5, 5, 33, 1, 1, (0, 0, 0, 0), 0, 0, True),       # 1x1, one tap: the tail chunk of the LAST weight row
]


@pytest.mark.parametrize("case", KTAIL_CASES)
def test_lds_dma_channel_tail_stays_inside_the_tensors(emu, case):
    """Regression (found by this model): in the K-tail variant of igemm_dma_kernel the weight chunk behind Ci was fetched with
    its row offset in range and the channel offset in the SGPR operand, which the descriptor's range check does not cover -
    the last row of the last tap read behind the weight tensor (times a zero A chunk: harmless unless the bytes there are a
    NaN).  The model fails such a launch; results must match as well."""
    sk = torch.zeros(emu.migan_conv_splitk_workspace() // 4)
    for what, err in _conv_case(emu, case, sk).items():
        assert err <= 1e-5, (case, what, err)


def test_no_conv_kernel_leaves_its_tensors(emu):
    """Every operand of every conv case ends at an inaccessible page (tests/hipemu/guarded_conv_cases.py): an over-read or
    over-write past a tensor is a SIGSEGV of the child, with the case it was running as the last line of its output."""
    import subprocess

    script = os.path.join(os.path.dirname(os.path.abspath(hipemu.__file__)), "guarded_conv_cases.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ALL OK"), (r.returncode, r.stdout[-400:], r.stderr[-800:])


def test_register_staged_conv_kernels_stay_inside_their_tensors(emu):
    """The same with MIGAN_DMA=0 MIGAN_DMA_WGRAD=0: every geometry on the register-staged family (igemm_pipe_kernel, wgrad_inc_kernel,
    wgrad_pipe_kernel) - the shipped alternative of the LDS-DMA kernels and what serves the geometries those decline."""
    import subprocess

    script = os.path.join(os.path.dirname(os.path.abspath(hipemu.__file__)), "guarded_conv_cases.py")
    env = dict(os.environ, MIGAN_DMA="0", MIGAN_DMA_WGRAD="0")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ALL OK"), (r.returncode, r.stdout[-400:], r.stderr[-800:])


def test_fewpix_kernels_stay_inside_their_tensors(emu):
    """The same for csrc/fewpix.hip: im2col / col2im and the three skinny GEMM forms at few-pixel-conv sizes, every operand
    against an inaccessible page, NaN-filled outputs, results against torch."""
    import subprocess

    script = os.path.join(os.path.dirname(os.path.abspath(hipemu.__file__)), "guarded_conv_cases.py")
    r = subprocess.run([sys.executable, script, "fewpix"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ALL OK"), (r.returncode, r.stdout[-400:], r.stderr[-800:])


@pytest.mark.parametrize("family", ["upconv", "norm", "norm_prelu", "toeplitz", "rgb", "eltwise", "loss", "adam", "skinny", "classify"])
def test_upconv_and_norm_kernels_stay_inside_their_tensors(emu, family):
    """The same for the phase-collapsed up-conv (pack, forward, input gradient, weight gradient + its four-class slab reduction over
    1 ... 20 splits) and for the norm / column-sum passes (chunked partial kernels + one-wave-per-channel finalize kernels over 1 ... > 256
    chunks): the reductions load several slabs / records per round through a CLAMPED index - an index one past the range would read the
    guard page here and nothing a GPU run notices; and for the width-Toeplitz path of the image-output 7x7 / 9x9 convs (pack, forward,
    expansion, both gradients; odd widths, 2-4 output channels, zero and reflection padding), the image-input convs of csrc/rgb_conv.hip
    (forward with its row prefetch, the fused weight gradient, the thin-output layer's input gradient) and the elementwise / index kernels of
    csrc/eltwise.hip on element counts that are no multiple of a vector width; and the mean-reduced losses with their gradients, the
    per-row norm and the row scaling of the gradient penalty; and the one-launch Adam over parameters of 1 ... 12800 elements (16-byte
    accesses on whole chunks, scalar tails) against torch.optim.Adam; the <= 64-row GEMMs behind nn.Linear at small batch; BatchNorm2d [PixelShuffle] PReLU inside the norm launches; and softmax / cross entropy / embedding, a weight pack and a mask draw."""
    import subprocess

    script = os.path.join(os.path.dirname(os.path.abspath(hipemu.__file__)), "guarded_conv_cases.py")
    r = subprocess.run([sys.executable, script, family], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ALL OK"), (r.returncode, r.stdout[-400:], r.stderr[-800:])


def test_splitk_tickets_and_determinism(emu):
    """pix2pix/models.py:66 geometry (4 output pixels, K = 8192): slices add in slice order whoever arrives last - with the
    model's workgroups on 1, 3 and 8 OS threads (different arrival orders) the result is bit-identical."""
    sk = torch.zeros(emu.migan_conv_splitk_workspace() // 4)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 4, 4, 512, generator=g)
    w = torch.randn(512, 4, 4, 512, generator=g) * 0.05
    outs = []
    for threads in (1, 3, 8):
        emu.hipemu_set_threads(threads)
        y = torch.empty(1, 2, 2, 512)
        rc = emu.migan_conv2d_fwd_ws(_ptr(x), _ptr(w), None, None, _ptr(y), 1, 4, 4, 512, 2, 2, 512, 4, 4, 2, 1, 1, 0, 0, 0.0,
                                     _ptr(sk), sk.numel() * 4, None)
        assert rc == 0
        outs.append(y)
    emu.hipemu_set_threads(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert int(sk[:1024].view(torch.int32).abs().sum()) == 0
    ref = TF.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, 2, 1)
    assert _rel(outs[0].permute(0, 3, 1, 2), ref) <= 3e-6


@pytest.mark.parametrize("shape,perm", [((96, 80, 4, 4), (0, 2, 3, 1)), ((96, 80, 4, 4), (1, 2, 3, 0)), ((130, 70, 3, 3), (1, 2, 3, 0)),
                                        ((64, 100, 7, 7), (0, 2, 3, 1)), ((70, 40, 4, 4), (0, 2, 3, 1)), ((5, 6, 7, 8), (3, 1, 0, 2))])
def test_weight_packs(emu, shape, perm):
    """OHWI / IHWO packs: pack_transpose_kernel (>= 64 k elements) and the generic permute4_kernel are exact copies."""
    w = torch.randn(*shape)
    out = torch.empty(w.numel())
    emu.hipemu_reset_counts()
    assert emu.migan_permute4d(_ptr(w), _ptr(out), *shape, *perm, None) == 0
    assert torch.equal(out, w.permute(*perm).contiguous().view(-1))
    tiled = w.numel() >= (1 << 16) and perm in ((0, 2, 3, 1), (1, 2, 3, 0))
    assert emu.hipemu_launch_count(b"pack_transpose_kernel") == (1 if tiled else 0)


def _critic_reference(real, fake, alpha, params, lam, slope):
    """wgan_gp.py:68-83,119-138,160-176 with torch autograd (double backward through autograd.grad(create_graph=True))."""
    ps = [t.clone().requires_grad_(True) for t in params]

    def D(x):
        h = TF.leaky_relu(x @ ps[0].t() + ps[1], slope)
        h = TF.leaky_relu(h @ ps[2].t() + ps[3], slope)
        return h @ ps[4].t() + ps[5]

    rv, fv = D(real), D(fake)
    xh = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
    dv = D(xh)
    g = torch.autograd.grad(dv, xh, torch.ones_like(dv), create_graph=True, retain_graph=True)[0]
    gp = ((g.norm(2, dim=1) - 1) ** 2).mean()
    d_loss = -rv.mean() + fv.mean() + lam * gp
    d_loss.backward()
    return [float(d_loss), float(gp), float(rv.mean()), float(fv.mean())], [p.grad for p in ps]


@pytest.mark.parametrize("B,dims", [(64, (1024, 512, 256)), (8, (1024, 512, 256)), (33, (256, 128, 128)), (1, (128, 256, 128)), (17, (384, 128, 256))])
def test_fused_critic_kernels_against_autograd(emu, B, dims):
    """K7 (csrc/critic_fused.hip) through the C ABI: six launches; losses and all six parameter gradients against torch's double
    backward; gradients written (first call, into NaN-filled buffers) and accumulated (two more calls)."""
    Din, H1, H2 = dims
    g = torch.Generator().manual_seed(B)
    real = torch.rand(B, Din, generator=g) * 2 - 1
    fake = torch.tanh(torch.randn(B, Din, generator=g))
    alpha = torch.rand(B, 1, generator=g)

    def lin(o, i):
        k = 1 / i ** 0.5
        return (torch.rand(o, i, generator=g) * 2 - 1) * k, (torch.rand(o, generator=g) * 2 - 1) * k

    params = [*lin(H1, Din), *lin(H2, H1), *lin(1, H2)]
    want, gref = _critic_reference(real, fake, alpha, params, 10.0, 0.2)
    grads = [torch.full_like(t, float("nan")) for t in params]   # the first call WRITES every gradient
    out = torch.zeros(4)
    wsb = emu.migan_critic_fused_workspace(B, Din, H1, H2)
    ws = torch.full((wsb // 4,), float("nan"))
    for rep in range(3):
        rc = emu.migan_critic_fused(_ptr(real), _ptr(fake), _ptr(alpha), *[_ptr(t) for t in params], *[_ptr(t) for t in grads],
                                    _ptr(out), _ptr(ws), wsb, B, Din, H1, H2, 0.2, 10.0, int(rep > 0), 0, None)
        assert rc == 0, emu.hipemu_last_message()
        for a, b in zip(out.tolist(), want):
            assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (out.tolist(), want)
        for got, ref in zip(grads, gref):
            if float(ref.norm()) > 0:
                assert _rel(got, ref * (rep + 1)) <= 3e-6
            else:
                assert float(got.abs().max()) == 0.0     # b3: -1 + 1


@pytest.mark.parametrize("B,img_shape", [(64, (1, 32, 32)), (8, (1, 32, 32)), (33, (1, 28, 28)), (2, (3, 16, 16))])
def test_fused_generator_forward_against_torch(emu, B, img_shape):
    """csrc/mlp_fused.hip through the C ABI against the oracle's MlpGenerator in training mode: output, BatchNorm1d running
    statistics (momentum, unbiased variance) and num_batches_tracked; K = 100 (a K tail), N = 784 (49 column tiles)."""
    import copy
    import ctypes

    from oracle import reference_models as M

    torch.manual_seed(B)
    G = M.MlpGenerator(img_shape)
    G.train()
    for m in G.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.1, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    G2 = copy.deepcopy(G)
    z = torch.randn(B, 100)
    with torch.no_grad():
        want = G(z).view(B, -1)
    mods, groups, i = list(G2.model), [], 0
    while i < len(mods):
        lin, bn, act, slope = mods[i], None, 0, 0.0
        i += 1
        if i < len(mods) and isinstance(mods[i], torch.nn.BatchNorm1d):
            bn = mods[i]
            i += 1
        if i < len(mods) and isinstance(mods[i], torch.nn.LeakyReLU):
            act, slope = 1, mods[i].negative_slope
            i += 1
        elif i < len(mods) and isinstance(mods[i], torch.nn.Tanh):
            act = 3
            i += 1
        groups.append((lin, bn, act, slope))
    n = len(groups)
    dims = (ctypes.c_int * (4 * n))(*[v for (l, bn, a, s) in groups for v in (l.in_features, l.out_features, int(bn is not None), a)])
    fpar = (ctypes.c_float * (3 * n))(*[v for (l, bn, a, s) in groups for v in (s, bn.eps if bn else 0.0, bn.momentum if bn else 0.0)])
    plist = []
    for l, bn, a, s in groups:
        plist += [_ptr(l.weight), _ptr(l.bias)]
        plist += [_ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean), _ptr(bn.running_var), _ptr(bn.num_batches_tracked)] if bn else [None] * 5
    ptrs = (ctypes.c_void_p * len(plist))(*plist)
    assert emu.migan_mlp_fused_ok(B, n, dims)
    wsb = emu.migan_mlp_fused_workspace(B, n, dims, 0)
    ws = torch.full((wsb // 4,), float("nan"))
    y = torch.empty(B, groups[-1][0].out_features)
    tickets = torch.zeros(1024, dtype=torch.int32)
    assert emu.migan_mlp_fused_fwd(_ptr(z), _ptr(y), B, n, dims, fpar, ptrs, _ptr(ws), wsb, 0, _ptr(tickets), 0, None) == 0, emu.hipemu_last_message()
    assert int(tickets.abs().sum()) == 0, "the column tickets must be back at zero"
    assert _rel(y, want) <= 3e-6
    for a, b in zip(G2.buffers(), G.buffers()):
        assert torch.allclose(a.double(), b.double(), rtol=1e-5, atol=1e-6)


def _mlp_groups(seq):
    mods, groups, i = list(seq), [], 0
    while i < len(mods):
        lin, bn, act, slope = mods[i], None, 0, 0.0
        i += 1
        if i < len(mods) and isinstance(mods[i], torch.nn.BatchNorm1d):
            bn = mods[i]
            i += 1
        if i < len(mods) and isinstance(mods[i], torch.nn.LeakyReLU):
            act, slope = 1, mods[i].negative_slope
            i += 1
        elif i < len(mods) and isinstance(mods[i], torch.nn.Tanh):
            act = 3
            i += 1
        groups.append((lin, bn, act, slope))
    return groups


@pytest.mark.parametrize("which,B", [("generator", 64), ("generator", 8), ("critic", 64), ("critic", 33)])
def test_fused_mlp_backward_against_autograd(emu, which, B):
    """csrc/mlp_fused.hip, the pair a generator iteration is made of: forward that keeps its activations + backward (one launch per
    layer / phase), gradients written into NaN-filled buffers.  generator: Linear / BatchNorm1d (training) / LeakyReLU / Tanh of wgan_gp.py:42-65, all parameter gradients incl.
    dgamma / dbeta;  critic: wgan_gp.py:68-83 with its single output column, parameter gradients and the input gradient."""
    import copy
    import ctypes

    from oracle import reference_models as M

    torch.manual_seed(B)
    if which == "generator":
        model, x, dy, want_dx = M.MlpGenerator((1, 32, 32)).model, torch.randn(B, 100), torch.randn(B, 1024) * 0.1, False
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.uniform_(-0.2, 0.2)
    else:
        model, x, dy, want_dx = M.MlpCritic((1, 32, 32)).model, torch.randn(B, 1024), torch.randn(B, 1), True
    model.train()
    ref = copy.deepcopy(model)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy)
    groups = _mlp_groups(model)
    n = len(groups)
    dims = (ctypes.c_int * (4 * n))(*[v for (l, bn, a, s) in groups for v in (l.in_features, l.out_features, int(bn is not None), a)])
    fpar = (ctypes.c_float * (3 * n))(*[v for (l, bn, a, s) in groups for v in (s, bn.eps if bn else 0.0, bn.momentum if bn else 0.0)])
    plist, glist, gts = [], [], []
    for l, bn, a, s in groups:
        plist += [_ptr(l.weight), _ptr(l.bias)]
        plist += [_ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean), _ptr(bn.running_var), _ptr(bn.num_batches_tracked)] if bn else [None] * 5
        g = [torch.full_like(l.weight, float("nan")), torch.full_like(l.bias, float("nan"))] + \
            ([torch.full_like(bn.weight, float("nan")), torch.full_like(bn.bias, float("nan"))] if bn else [None, None])
        gts.append(g)
        glist += [_ptr(q) for q in g]
    ptrs, gptrs = (ctypes.c_void_p * len(plist))(*plist), (ctypes.c_void_p * len(glist))(*glist)
    wsb = emu.migan_mlp_fused_workspace(B, n, dims, 1)
    save = torch.full((wsb // 4,), float("nan"))
    bwb = emu.migan_mlp_fused_bwd_workspace(B, n, dims)
    bws = torch.full((bwb // 4,), float("nan"))
    y = torch.empty(B, groups[-1][0].out_features)
    dx = torch.full_like(x, float("nan")) if want_dx else None
    tickets = torch.zeros(1024, dtype=torch.int32)
    assert emu.migan_mlp_fused_fwd(_ptr(x), _ptr(y), B, n, dims, fpar, ptrs, _ptr(save), wsb, 1, _ptr(tickets), 0, None) == 0
    assert int(tickets.abs().sum()) == 0
    assert emu.migan_mlp_fused_bwd(_ptr(x), _ptr(y), _ptr(dy), _ptr(save), _ptr(dx), B, n, dims, fpar, ptrs, gptrs, _ptr(bws), bwb,
                                   0, 0, None) == 0, emu.hipemu_last_message()
    assert _rel(y, yr.detach()) <= 3e-6
    if want_dx:
        assert _rel(dx, xr.grad) <= 5e-6
    for (l, bn, _, _), g in zip(_mlp_groups(ref), gts):
        assert _rel(g[0], l.weight.grad) <= 5e-6
        if bn is None:
            assert _rel(g[1], l.bias.grad) <= 5e-6
        else:   # a bias in front of BatchNorm has a zero true gradient: rounding noise on both sides
            assert float((g[1] - l.bias.grad).abs().max()) <= 1e-5
            assert _rel(g[2], bn.weight.grad) <= 5e-6 and _rel(g[3], bn.bias.grad) <= 5e-6


def _run_gpu_test_body(module_name, test_name, *args):
    """Run the body of a `-m gpu` parity test on CPU tensors: its device is "cpu", its kernels are the execution model."""
    import importlib

    import hipemu.host
    import util

    _load_or_skip()
    T = importlib.import_module(module_name)
    saved = (getattr(T, "DEV", None), getattr(T, "gpu_copy", None))
    T.DEV = "cpu"
    T.gpu_copy = lambda m, device="cpu": util.gpu_copy(m, "cpu")
    try:
        with hipemu.host.emulated_device() as lib:
            lib.hipemu_reset_counts()
            getattr(T, test_name)(*args)
            assert lib.hipemu_launch_count(b"_kernel") > 0, "no kernel ran"
            return lib
    finally:
        T.DEV, T.gpu_copy = saved


STEP_BODIES = [
    ("test_dcgan_steps", (True,)),            # dcgan.py:143-183, 3 steps: paired D pass, chained BatchNorm statistics, dropout masks
    ("test_dcgan_steps_three_channels", ()),  # dcgan.py:28 --channels 3 at 64x64 (bench.py's extra.dcgan_ch3)
    ("test_wgan_gp_steps", (False,)),         # wgan_gp.py:119-193, 6 critic iterations: double backward through the skinny GEMMs
    ("test_wgan_gp_steps", (True,)),          # ... and on the fused WGAN-GP kernels (K7)
    ("test_fused_wgan_gp_kernels_serve_the_baseline_batch", ()),   # batch 64: launch counters + fused vs op-by-op vs oracle
    ("test_dragan_steps", ()),                # dragan.py:176-217: conv-critic gradient penalty
    ("test_srgan_step", ()),                  # srgan.py:97-145: PixelShuffle epilogue, VGG features, Toeplitz 9x9
    ("test_pix2pix_step", ()),                # pix2pix.py:123-172 at 256x256: split-K, ConvTranspose, PatchGAN head, 4-8 M-element weights
]


@pytest.mark.parametrize("name,args", STEP_BODIES, ids=["%s%s" % (n, list(a) if a else "") for n, a in STEP_BODIES])
def test_step_parity_bodies_on_the_execution_model(name, args):
    """The training-step parity tests of test_steps_gpu.py, unchanged, against the oracle: losses of every step, weights after
    Adam, BatchNorm buffers - computed by the HIP kernels' source running on the host."""
    lib = _run_gpu_test_body("test_steps_gpu", name, *args)
    if name == "test_wgan_gp_steps":  # six fused iterations
        # (each critic iteration = six launches, csrc/critic_fused.hip)
        assert lib.hipemu_launch_count(b"critic_fused_p") == (6 * 6 if args[0] else 0)
        # 6 no_grad forwards + the fused generator iterations 0 and 5 (2 saving forwards each)
        # one launch per layer, the generator's 100 -> 128 layer inside the launch of the next: generator 4, critic-as-MLP 3 ->
        # 6 * 4 + 2 * (4 + 3); backward: critic (top + 3 chain phases, dx) 4 + generator (top + 4 chain phases + gradients) 6, twice
        assert lib.hipemu_launch_count(b"mlp_fused_fwd_kernel") == (38 if args[0] else 0)
        assert lib.hipemu_launch_count(b"mlp_fused_bwd_kernel") == (20 if args[0] else 0)
    if name == "test_pix2pix_step":   # the kernels this workload is there for
        for sym in (b"thin_conv_wave_kernel", b"wgrad_reduce_tr_kernel", b"pack_transpose_kernel", b"true>"):
            assert lib.hipemu_launch_count(sym) > 0, sym


@pytest.mark.parametrize("idx", [0, 2, 3])
def test_conv3x3_tap_inner_k_order_on_the_execution_model(idx, monkeypatch):
    """test_ops_gpu.py::test_conv3x3_tap_inner_k_order without a GPU: the nine-taps-inner K order of igemm_dma_kernel (TAPS_IN = 9) and the
    tap-outer order on the same shapes - reflection gather on 128 x 128 tiles, stride 2, ragged M / N."""
    import test_ops_gpu

    _run_gpu_test_body("test_ops_gpu", "test_conv3x3_tap_inner_k_order", __import__("pytorch_gan_amd"), test_ops_gpu.TAP9_CASES[idx], monkeypatch)


def test_trunk_block_batchnorm_prelu_folded_into_the_conv_on_the_execution_model(monkeypatch):
    """test_ops_gpu.py::test_trunk_block_batchnorm_prelu_folded_into_the_conv without a GPU: the input map of c64_conv_kernel<2> /
    c64_wgrad_kernel<2> (BatchNorm2d(64, 0.8) -> PReLU applied on the way into the LDS row ring) inside the srgan residual block,
    against torch in fp64 and against the separate norm launches."""
    lib = _run_gpu_test_body("test_ops_gpu", "test_trunk_block_batchnorm_prelu_folded_into_the_conv", __import__("pytorch_gan_amd"), (1, 9, 64),
                             monkeypatch)
    assert lib.hipemu_launch_count(b"c64_wgrad_kernel<2>") > 0


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_generator_tail_batchnorm_folded_into_the_image_conv_on_the_execution_model(idx, monkeypatch):
    """test_ops_gpu.py::test_generator_tail_batchnorm_folded_into_the_image_conv without a GPU: thin_conv_kernel<1, true>'s input map and the
    two backward walks that recompute the conv's input gradient (csrc/norm.hip bn_conv1_bwd_*), ragged maps and image borders included."""
    import test_ops_gpu

    lib = _run_gpu_test_body("test_ops_gpu", "test_generator_tail_batchnorm_folded_into_the_image_conv", __import__("pytorch_gan_amd"),
                             test_ops_gpu.BN_CONV1_CASES[idx], monkeypatch)
    assert lib.hipemu_launch_count(b"bn_conv1_bwd_apply_kernel") > 0


def test_cross_replica_batchnorm_two_ranks_on_the_execution_model(tmp_path, monkeypatch):
    """SURVEY.md 8e with real kernels and no GPU: the body of test_dp_gpu.py::test_cross_replica_batchnorm_equals_full_batch -
    two torch.distributed.run ranks (gloo) each run half of a DCGAN batch with enable_sync_batchnorm() (local moments ->
    all_gather -> Chan combination; backward sums all-reduced), their kernels on the execution model, against the
    single-process full-batch step: losses, averaged gradients, BatchNorm running statistics."""
    monkeypatch.setenv("MIGAN_TEST_EMU", "1")
    monkeypatch.setenv("MIGAN_TEST_DEVICE", "cpu")
    _run_gpu_test_body("test_dp_gpu", "test_cross_replica_batchnorm_equals_full_batch", tmp_path)


def test_cross_replica_batchnorm_keeps_the_fused_prelu_form(emu, monkeypatch):
    """The fused BatchNorm2d + PReLU [+ PixelShuffle] launches (srgan/models.py:23-24,55-57) under cross-replica statistics: forward
    from the gathered moments, backward in two halves around the all-reduce of the two batch sums (migan_norm_bwd_sums_prelu /
    migan_norm_bwd_apply_prelu).  A stand-in exchange for TWO ranks holding the SAME shard (all_gather = the moments twice, all_reduce =
    twice the sums, P_total = 2 P): the global statistics then equal the local ones, so output, input gradient, affine gradients and the
    slope gradient must equal the single-rank fused launches'."""
    import hipemu.host
    from pytorch_gan_amd import functional as F

    class TwoEqualShards:
        world = 2

        def all_gather(self, t):
            return torch.cat([t, t])

        def all_reduce_sum(self, t):
            t.mul_(2.0)
            return t

    torch.manual_seed(3)
    x0 = (torch.randn(2, 8, 6, 4) * 1.5 + 0.3).contiguous(memory_format=torch.channels_last)
    g0, b0 = torch.rand(8) + 0.5, torch.randn(8)
    for shuffle in (0, 2):
        res = []
        for sync in (None, TwoEqualShards()):
            x = x0.clone().requires_grad_(True)
            gamma, beta = g0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            pw = torch.full((1,), 0.25).requires_grad_(True)
            with hipemu.host.emulated_device():
                monkeypatch.setattr(F, "_SYNC_BN", sync)
                y = F.norm(x, gamma, beta, eps=0.8, prelu=pw, shuffle=shuffle)
                g = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).contiguous(memory_format=torch.channels_last)
                y.backward(g)
            res.append([t.detach().clone() for t in (y, x.grad, gamma.grad, beta.grad, pw.grad)])
        monkeypatch.setattr(F, "_SYNC_BN", None)
        for a, b, what in zip(res[0], res[1], ("y", "dx", "dgamma", "dbeta", "dprelu")):
            assert _rel(b.detach(), a.detach()) <= 2e-6, (shuffle, what, _rel(b.detach(), a.detach()))
    # and against torch: BatchNorm2d(8, 0.8) -> PReLU
    ref_x = x0.clone().requires_grad_(True)
    ref = TF.prelu(TF.batch_norm(ref_x, None, None, gamma.detach(), beta.detach(), True, 0.1, 0.8), torch.full((1,), 0.25))
    x = x0.clone().requires_grad_(True)
    with hipemu.host.emulated_device():
        monkeypatch.setattr(F, "_SYNC_BN", TwoEqualShards())
        y = F.norm(x, gamma.detach(), beta.detach(), eps=0.8, prelu=torch.full((1,), 0.25))
    monkeypatch.setattr(F, "_SYNC_BN", None)
    assert _rel(y, ref) <= 1e-5


def test_srgan_two_ranks_cross_replica_batchnorm_on_the_execution_model(tmp_path, monkeypatch):
    """Row N3 without a GPU: test_dp_gpu.py::test_srgan_two_images_per_rank_... at 8 -> 32 pixels with 2 residual blocks - two gloo
    ranks x 2 images, cross-replica BatchNorm with the fused PReLU / PixelShuffle launches, against the oracle's 4-image step."""
    monkeypatch.setenv("MIGAN_TEST_EMU", "1")
    monkeypatch.setenv("MIGAN_TEST_DEVICE", "cpu")
    _load_or_skip()
    import test_dp_gpu

    # every kernel runs in the rank processes; the parent holds the oracle's step
    test_dp_gpu.test_srgan_two_images_per_rank_with_cross_replica_batchnorm_equals_the_full_batch_oracle(tmp_path)


def test_cross_replica_batchnorm_recording_protocol_on_the_execution_model(tmp_path, monkeypatch):
    """test_dp_gpu.py::test_cross_replica_batchnorm_step_replays_as_graph_segments without a GPU: two gloo ranks run three DCGAN steps
    with a recording segmenter installed - 24 BatchNorm collectives + 2 optimiser exchanges go through _Segmenter.cut() on the thread
    that runs the step (backward nodes included), on operands that outlive the step; results equal three plain steps bit for bit."""
    monkeypatch.setenv("MIGAN_TEST_EMU", "1")
    monkeypatch.setenv("MIGAN_TEST_DEVICE", "cpu")
    _load_or_skip()
    import test_dp_gpu

    test_dp_gpu.test_cross_replica_batchnorm_step_replays_as_graph_segments(tmp_path)   # every kernel runs in the rank processes


def test_data_parallel_step_orders_on_the_execution_model(tmp_path, monkeypatch):
    """test_dp_gpu.py::test_both_step_orders_of_the_data_parallel_step_equal_the_full_batch without a GPU: two gloo ranks, CycleGAN
    at 32x32, 'sequential' and 'fork' order of the discriminator updates relative to the generator bucket's exchange - bit-identical
    to each other, and the first step equal to the single-process step on the whole batch."""
    monkeypatch.setenv("MIGAN_TEST_EMU", "1")
    monkeypatch.setenv("MIGAN_TEST_DEVICE", "cpu")
    _run_gpu_test_body("test_dp_gpu", "test_both_step_orders_of_the_data_parallel_step_equal_the_full_batch", tmp_path)


@pytest.mark.skipif(os.environ.get("MIGAN_EMU_SLOW") != "1", reason="70 s on 8 cores: MIGAN_EMU_SLOW=1")
def test_esrgan_full_depth_on_the_execution_model():
    """test_steps_gpu.py::test_esrgan_full_depth_steps (23 RRDB generator, warm-up + relativistic iteration) on the execution model;
    the VGG19 tail on 2x2 maps runs as few-pixel 3x3 stride-1 convs (csrc/fewpix.hip)."""
    lib = _run_gpu_test_body("test_steps_gpu", "test_esrgan_full_depth_steps")
    assert lib.hipemu_launch_count(b"im2col_small_kernel") > 0


@pytest.mark.skipif(os.environ.get("MIGAN_EMU_SLOW") != "1", reason="3.5 minutes on 8 cores: MIGAN_EMU_SLOW=1")
def test_pix2pix_trajectory_on_the_execution_model():
    """test_steps_gpu.py::test_pix2pix_trajectory_5_steps (five pix2pix iterations at 256x256 three ways: HIP kernels, oracle fp32, oracle
    fp64) with the kernels on the execution model - whose arithmetic is the hardware's: tools/abi_check prints the same differences
    here and on the MI355X (profiles/r03_abi_check.txt)."""
    lib = _run_gpu_test_body("test_steps_gpu", "test_pix2pix_trajectory_5_steps")
    assert lib.hipemu_launch_count(b"fewpix_nt_kernel") == 40 and lib.hipemu_launch_count(b"norm_small_fwd_kernel") == 80


@pytest.mark.skipif(os.environ.get("MIGAN_EMU_SLOW") != "1", reason="4 minutes on 8 cores: MIGAN_EMU_SLOW=1")
def test_cyclegan_steps_on_the_execution_model():
    _run_gpu_test_body("test_steps_gpu", "test_cyclegan_steps")


def test_replay_buffer_planned_picks_equal_the_eager_picks(emu):
    """cyclegan/utils.py:13-33 for a recorded step: ReplayBuffer.plan(B) draws the picks of the next call in front of it into the
    buffer's static table (fixed length 3*B, -1 = no pool update: select_rows skips the row), push_and_pop then launches over the
    table - the path a hipGraph replay takes.  Same `random` seed => the same returned batches and the same history as the eager
    call and as the reference's list logic, through fill-up, picks, and a slot replaced twice in one call."""
    import random

    import hipemu.host
    from oracle import reference_models as M
    from pytorch_gan_amd import steps

    torch.manual_seed(3)
    batches = [torch.randn(3, 2, 4, 4) for _ in range(12)]
    with hipemu.host.emulated_device():
        out = {}
        for mode in ("eager", "planned"):
            buf = steps.ReplayBuffer(4)
            random.seed(11)
            got = []
            for i, b in enumerate(batches):
                if mode == "planned" and i > 0:   # the pool exists after the first (eager) call, as after CycleGanRunner's warm-up
                    buf.plan(3)
                    assert buf._planned == 3
                got.append(buf.push_and_pop(b.clone()).clone())
                assert buf._planned is None
            out[mode] = (got, torch.cat(buf.samples()).clone())
        with pytest.raises(RuntimeError):
            buf.plan(3)
            buf.push_and_pop(batches[0][:2].clone())   # planned for 3 samples, called with 2
    ref = M.ReplayBuffer(4)
    random.seed(11)
    want = [ref.push_and_pop(b.clone()) for b in batches]
    for a, b, c in zip(out["eager"][0], out["planned"][0], want):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(out["eager"][1], out["planned"][1]) and torch.equal(out["eager"][1], torch.cat(ref.data))


@pytest.mark.parametrize("idx", [0, 2, 3])
def test_image_input_conv_kernels_on_the_execution_model(idx):
    """test_ops_gpu.py::test_rgb_conv_layers (csrc/rgb_conv.hip: forward and the fused activation-backward + bias + weight-gradient
    launch of the 3-channel layers) on the execution model, LDS poisoned: 3x3 with ragged row tiles, 32 output channels, 7x7 under
    reflection padding."""
    _load_or_skip()
    import pytorch_gan_amd as pg
    import test_ops_gpu

    _run_gpu_test_body("test_ops_gpu", "test_rgb_conv_layers", pg, test_ops_gpu.RGB_CASES[idx])


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_thin_output_conv_kernel_on_the_execution_model(idx):
    """test_ops_gpu.py::test_thin_output_3x3_conv (dcgan.py:62; its input gradient also through migan_rgb_conv_fwd with reversed taps)."""
    _load_or_skip()
    import pytorch_gan_amd as pg

    cases = [(2, 64, 128, 1, 3), (1, 100, 170, 3, 0), (3, 48, 130, 2, 1)]
    _run_gpu_test_body("test_ops_gpu", "test_thin_output_3x3_conv", pg, cases[idx])


def test_step_plans_belong_to_the_step_state(emu):
    """the plans that batch a step's weight packs / dropout masks into one launch used to be one per DEVICE, so two
    step bodies alternating on a device overwrote each other's plan every step (tables rebuilt, arena re-allocated, per-weight
    pack launches again).  They now live on the step state: two DCGAN states stepping alternately each settle into one
    multi-tensor pack launch and one batched mask draw per step, and the plans die with their state."""
    import gc
    import weakref

    import hipemu.host
    import util
    from oracle import reference_steps as S
    from pytorch_gan_amd import steps

    torch.manual_seed(0)
    with hipemu.host.emulated_device() as lib:
        states = []
        for seed in (0, 1):
            torch.manual_seed(seed)
            base = S.make_dcgan(32)
            states.append(steps.make_gan_state(util.gpu_copy(base.G, "cpu"), util.gpu_copy(base.D, "cpu")))
        imgs, z = torch.rand(4, 1, 32, 32) * 2 - 1, torch.randn(4, 100)
        per_step = []
        for rnd in range(4):
            for s in states:
                lib.hipemu_reset_counts()
                steps.dcgan_step(s, imgs, z)
                per_step.append((lib.hipemu_launch_count(b"multi_permute4_kernel"),
                                 lib.hipemu_launch_count(b"permute4_kernel") - lib.hipemu_launch_count(b"multi_permute4_kernel")
                                 + lib.hipemu_launch_count(b"pack_transpose_kernel"), lib.hipemu_launch_count(b"rand_mask_kernel")))
        # from each state's second step on: ONE multi-tensor pack launch, no per-weight pack launches, ONE mask draw
        assert per_step[2:] == [(1, 0, 1)] * 6, per_step
        plan = states[0].__dict__["_migan_plans"][("pack", "cpu")]
        ref = weakref.ref(plan)
        del plan, s
        states.clear()
        gc.collect()
        assert ref() is None, "a step state's plans outlived it"


@pytest.mark.parametrize("which,cfg", [("test_fewpix_conv2d", (1, 512, 2, 2, 512, False, 1)), ("test_fewpix_conv2d", (2, 256, 4, 4, 256, True, 1)),
                                       ("test_fewpix_conv2d", (3, 64, 6, 10, 1024, True, 2)),
                                       ("test_fewpix_conv_transpose2d", (1, 1024, 2, 2, 512, False, 0)),
                                       ("test_fewpix_conv_transpose2d", (4, 256, 4, 4, 256, True, 0)),
                                       ("test_fewpix_conv_transpose2d", (2, 128, 5, 6, 512, False, 0))],
                         ids=lambda v: v if isinstance(v, str) else "x".join(map(str, v[:5])))
def test_fewpix_convs_on_the_execution_model(emu, which, cfg):
    """csrc/fewpix.hip (Conv2d / ConvTranspose2d with <= 64 pixel rows and >= 1 M weights: the inner U-Net levels of pix2pix at
    batch 1): the bodies of the GPU parity tests against torch, and what ran - the two index kernels around the skinny GEMMs on the
    weight as stored: no weight pack, no tiled / split-K kernel, no weight-gradient slab reduction."""
    import pytorch_gan_amd as pg

    lib = _run_gpu_test_body("test_ops_gpu", which, pg, cfg)
    assert lib.hipemu_launch_count(b"im2col_small_kernel") == 1 and lib.hipemu_launch_count(b"col2im_small_kernel") == 1
    # the NT product: K split over workgroups (+ the ordered sum of the partial tiles) where the shape allows, else the plain launch
    split = lib.hipemu_launch_count(b"fewpix_nt_kernel")
    assert split == lib.hipemu_launch_count(b"fewpix_nt_reduce_kernel") and split + lib.hipemu_launch_count(b"skinny_nt_kernel") == 1
    assert split == (1 if (cfg[1] if which == "test_fewpix_conv2d" else cfg[4]) * 16 % 2048 == 0 else 0)
    assert lib.hipemu_launch_count(b"skinny_nn_kernel") == 1 and lib.hipemu_launch_count(b"skinny_tn_kernel") == 1
    for sym in (b"permute4_kernel", b"pack_transpose_kernel", b"igemm_", b"wgrad_", b"smallk_"):
        assert lib.hipemu_launch_count(sym) == 0, sym


@pytest.mark.parametrize("case", list(__import__("kernel_cases").all_cases()), ids=[c[0] for c in __import__("kernel_cases").all_cases()])
def test_geometry_selects_kernel_on_the_execution_model(emu, case):
    """tests/kernel_cases.py on the execution model: the body of test_ops_gpu.py::test_geometry_selects_kernel - every shape is served
    by the specialised kernel written for it (the library's own launch counters, the ones the GPU test reads) and agrees with torch."""
    import pytorch_gan_amd as pg

    _run_gpu_test_body("test_ops_gpu", "test_geometry_selects_kernel", pg, case)


def test_abi_check_harness_on_the_execution_model(emu, tmp_path):
    """tools/abi_check.cpp - the torch-free program that checks and times the latency-regime kernels on the hardware - built against
    the execution model instead of the HIP runtime: its arguments, geometries and host references must hold here too.  In this build
    every buffer the program allocates ends at an inaccessible page and starts out as NaN, so the fused WGAN-GP kernels (critic_fused's
    six launches, mlp_fused forward / backward with their ticketed BatchNorm1d hand-offs) and the one-launch InstanceNorm are also
    checked for accesses outside their operands and for workspace words read before they are written."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["bash", os.path.join(root, "tools", "build_abi_check.sh"), "host"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-800:]
    import re
    import shutil

    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):   # the program's HIP form must keep compiling too (no GPU needed)
        r = subprocess.run(["bash", os.path.join(root, "tools", "build_abi_check.sh")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "built tools/abi_check.bin" in r.stdout, r.stdout[-400:] + r.stderr[-1200:]
    sections = ["norm", "mlp", "critic"]
    digests = {}
    for sec in sections:
        r = subprocess.run(["/tmp/abi_check_host", sec], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.strip().endswith("ALL OK") and "FAIL" not in r.stdout, (sec, r.stdout[-1500:], r.stderr[-400:])
        digests[sec] = re.findall(r"digest ([0-9a-f]{16})", r.stdout)
    assert len(digests["mlp"]) == 2 and len(digests["critic"]) == 1, digests
    # the ticketed BatchNorm1d hand-offs and the K-slice reductions of the fused kernels: waves in reverse order, workgroups on 3 OS
    # threads (other arrival orders at the tickets) - bit-identical outputs, losses and gradients
    env = dict(os.environ, HIPEMU_SCHED="rev", HIPEMU_THREADS="3")
    for sec in ("mlp", "critic"):
        r = subprocess.run(["/tmp/abi_check_host", sec], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and r.stdout.strip().endswith("ALL OK"), (sec, r.stdout[-1500:], r.stderr[-400:])
        assert re.findall(r"digest ([0-9a-f]{16})", r.stdout) == digests[sec], (sec, "results depend on the wave schedule / arrival order")


def test_smoke_body_on_the_execution_model(emu, capsys):
    """__graft_entry__.smoke() - the driver's first call on the GPU box - with its kernels on the execution model."""
    import __graft_entry__ as entry
    import hipemu.host

    with hipemu.host.emulated_device():
        saved = torch.cuda.is_available
        torch.cuda.is_available = lambda: True
        try:
            entry.smoke()
        finally:
            torch.cuda.is_available = saved
    assert "smoke ok" in capsys.readouterr().out


def test_bench_builders_and_roofline_accounting_run(emu):
    """bench.py's own code around the step - the DCGAN and WGAN-GP builders, the per-launch accounting of `roofline` (the
    wrappers index the C entries' argument lists) - on the execution model at a small batch: a Python error there would only
    show as a bench line without its `roofline` object on the GPU box."""
    import types

    import bench
    import hipemu.host
    from pytorch_gan_amd.dp import LocalStepper

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert os.path.samefile(os.path.dirname(bench.__file__), root)
    args = types.SimpleNamespace(batch=4, no_graph=True, global_batch=0, sync_bn=False)
    with hipemu.host.emulated_device():
        w = bench.BUILDERS["dcgan"](LocalStepper(), 0, torch.device("cpu"), args, 3)
        out = w.run(0)
        assert all(torch.isfinite(out[k]) for k in ("g_loss", "d_loss"))
        r = bench.roofline(w, 0, 1)
        assert r["kernel"] in r["conv_kernels"] and r["launches_per_step"] >= 1
        assert any(k.startswith("upconv_wgrad") for k in r["conv_kernels"]) and r["hbm"]["kernel"].startswith("norm_")
        args.batch = 0
        g = bench.BUILDERS["wgan_gp"](LocalStepper(), 0, torch.device("cpu"), args, 3)
        for i in range(2, 7):
            out = g.run(i)
        assert g.state._k7_plan.ok and g.state._k7_gen_plan.ok


def test_a_wait_that_is_one_tile_short_is_noticed(emu, tmp_path):
    """LDS-DMA is asynchronous in the model: data lands when the issuing wave waits for it (`s_waitcnt vmcnt(N)` / the wait the
    compiler puts in front of __syncthreads()).  Mutation check of that property: the four-stage split-K pipeline of
    igemm_dma_kernel with its counted wait loosened by one K-tile reads a stage that has not landed - the model must show it
    (the unmutated kernel is exact in test_splitk_tickets_and_determinism)."""
    import ctypes
    import shutil

    from hipemu import build_emu

    src = build_emu.CSRC
    tmp = str(tmp_path / "csrc")
    os.makedirs(tmp)
    for f in os.listdir(src):
        if f.endswith((".hip", ".h", ".inc", ".py")):
            shutil.copy(os.path.join(src, f), tmp)
    text = open(os.path.join(tmp, "conv_dma.hip")).read()
    old = '::"n"((NS - 2) * IPT)'
    assert text.count(old) == 1
    open(os.path.join(tmp, "conv_dma.hip"), "w").write(text.replace(old, '::"n"((NS - 1) * IPT)'))
    saved = (build_emu.CSRC, build_emu.BUILD, list(build_emu.FLAGS))
    try:
        build_emu.CSRC, build_emu.BUILD = tmp, str(tmp_path / "_build")
        build_emu.FLAGS = [tmp if a == src else a for a in build_emu.FLAGS]
        lib = ctypes.CDLL(build_emu.build(only=["conv_igemm.hip", "conv_dma.hip"]))
    finally:
        build_emu.CSRC, build_emu.BUILD, build_emu.FLAGS = saved
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.migan_conv2d_fwd_ws.argtypes = [P] * 5 + [I] * 14 + [ctypes.c_float, P, ctypes.c_size_t, P]
    lib.migan_conv_splitk_workspace.restype = ctypes.c_size_t
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 4, 4, 512, generator=g)
    w = torch.randn(512, 4, 4, 512, generator=g) * 0.05
    sk = torch.zeros(lib.migan_conv_splitk_workspace() // 4)
    y = torch.empty(1, 2, 2, 512)
    assert lib.migan_conv2d_fwd_ws(_ptr(x), _ptr(w), None, None, _ptr(y), 1, 4, 4, 512, 2, 2, 512, 4, 4, 2, 1, 1, 0, 0, 0.0, _ptr(sk),
                                   sk.numel() * 4, None) == 0
    ref = TF.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, 2, 1)
    err = _rel(y.permute(0, 3, 1, 2), ref)   # NaN since the model poisons LDS (the stage that has not landed holds 0xFF bytes), else large
    assert not (err <= 1e-2), "the model did not notice a stage read before it was waited for"


def test_a_deadlock_is_reported_not_hung(emu, tmp_path):
    """The model's own safety net: a kernel whose threads do not all reach a barrier ends the launch with an error."""
    src = tmp_path / "dead.cpp"
    src.write_text('''
#include <hip/hip_runtime.h>
__global__ void dead_kernel(int* out) {
    if (threadIdx.x < 32) __syncthreads();      // half the workgroup waits for a barrier the other half never reaches ...
    else while (out[0] == 0) __builtin_amdgcn_s_sleep(1);   // ... because it spins on a flag nobody sets
    out[1] = 1;
}
extern "C" __attribute__((visibility("default"))) int run_dead(int* out) {
    hipLaunchKernelGGL(dead_kernel, dim3(1), dim3(64), 0, 0, out);
    return hipGetLastError();
}
''')
    import ctypes
    import subprocess

    from hipemu import build_emu

    so = str(tmp_path / "dead.so")
    subprocess.check_call([build_emu.CXX] + build_emu.FLAGS + ["-shared", str(src), os.path.join(os.path.dirname(build_emu.__file__), "hipemu.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.hipemu_last_message.restype = ctypes.c_char_p
    out = torch.zeros(2, dtype=torch.int32)
    assert lib.run_dead(ctypes.c_void_p(out.data_ptr())) != 0
    assert b"deadlock" in lib.hipemu_last_message()


@pytest.mark.parametrize("shape", [(1, 8, 8, 32, 64), (1, 6, 8, 32, 40)], ids=["32-64ch", "k-tail-40ch"])
def test_relu_handoff_on_the_execution_model(emu, shape):
    """conv, ReLU, conv | MaxPool2d chains (vgg19.features[:18]): the ReLU backward inside the consumer's input-gradient epilogue
    (igemm_dma_kernel's `omask`) / the pool's backward - body of test_ops_gpu.py::test_relu_backward_handed_to_the_consumer."""
    import pytorch_gan_amd as pg

    lib = _run_gpu_test_body("test_ops_gpu", "test_relu_backward_handed_to_the_consumer", pg, shape)
    assert lib.hipemu_launch_count(b"igemm_dma_kernel") > 0


@pytest.mark.parametrize("Co", [40, 38], ids=["quads", "ragged-channels"])
def test_lds_dma_epilogue_forms_on_the_execution_model(emu, Co):
    """The epilogue of igemm_dma_kernel (a lane owns channel quads of one pixel): 16-byte stores when Co % 4 == 0, the per-channel form
    when not, each with the [N][Co] multiplier of the fused Dropout2d (`oscale`), the ReLU mask of the consumer hand-off (`omask`) and the
    accumulating ring launch of the reflection input gradient (`accum`) - against torch on the host."""
    g = torch.Generator().manual_seed(5)
    N, Ci, H, W = 3, 32, 12, 12
    sk = torch.zeros(emu.migan_conv_splitk_workspace() // 4)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.2
    b = torch.randn(Co, generator=g)
    mask = (torch.rand(N, Co, generator=g) > 0.4).float() * 2.0 if Co % 4 == 0 else None   # (the C ABI takes the multiplier for whole quads only)
    xn, w_ohwi = x.permute(0, 2, 3, 1).contiguous(), w.permute(0, 2, 3, 1).contiguous()
    emu.hipemu_reset_counts()
    # forward with bias, LeakyReLU and the per-(image, channel) multiplier
    y = torch.empty(N, H, W, Co)
    assert emu.migan_conv2d_fwd_ws(_ptr(xn), _ptr(w_ohwi), _ptr(b), _ptr(mask), _ptr(y), N, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 0, 1, 0.2,
                                   _ptr(sk), sk.numel() * 4, None) == 0, emu.hipemu_last_message()
    ref = TF.leaky_relu(TF.conv2d(x, w, b, 1, 1), 0.2)
    if mask is not None:
        ref = ref * mask[:, :, None, None]
    assert _rel(y.permute(0, 3, 1, 2), ref) <= 3e-6
    # input gradient of a conv whose input was a fused conv+ReLU: dx = (dy (*) w^T) where relu_out > 0  (the conv maps Co -> Ci here, so the
    # GEMM's output channels are Co: the same quad / ragged split)
    w2 = torch.randn(Ci, Co, 3, 3, generator=g) * 0.2   # Conv2d(Co, Ci): weight [Ci][Co][3][3]
    dy = torch.randn(N, Ci, H, W, generator=g)
    relu_out = torch.relu(torch.randn(N, Co, H, W, generator=g))
    dx = torch.empty(N, H, W, Co)
    w2_ihwo = w2.permute(1, 2, 3, 0).contiguous()       # [Co][3][3][Ci]
    assert emu.migan_conv2d_dgrad_relu_ws(_ptr(dy.permute(0, 2, 3, 1).contiguous()), _ptr(w2_ihwo), _ptr(dx),
                                          _ptr(relu_out.permute(0, 2, 3, 1).contiguous()), N, H, W, Co, H, W, Ci, 3, 3, 1, 1, 1, _ptr(sk),
                                          sk.numel() * 4, None) == 0, emu.hipemu_last_message()
    ref = TF.conv_transpose2d(dy, w2, None, 1, 1) * (relu_out > 0)
    assert _rel(dx.permute(0, 3, 1, 2), ref) <= 3e-6
    # ReflectionPad2d(1) + Conv3x3 input gradient straight into H x W: interior launch + accumulating ring launch
    if Co % 4 == 0:
        xr = torch.randn(N, Co, H, W, generator=g, requires_grad=True)
        TF.conv2d(TF.pad(xr, (1, 1, 1, 1), mode="reflect"), w2).backward(dy)
        dxr = torch.empty(N, H, W, Co)
        assert emu.migan_conv2d_dgrad_reflect1_ws(_ptr(dy.permute(0, 2, 3, 1).contiguous()), _ptr(w2_ihwo), _ptr(dxr), N, H, W, Co, Ci,
                                                  _ptr(sk), sk.numel() * 4, None) == 0, emu.hipemu_last_message()
        assert _rel(dxr.permute(0, 3, 1, 2), xr.grad) <= 3e-6
    assert emu.hipemu_launch_count(b"igemm_dma_kernel") >= 2
    assert int(sk[:1024].view(torch.int32).abs().sum()) == 0
