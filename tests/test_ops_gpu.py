"""Per-op parity of the HIP kernels (through the C ABI + autograd wrappers) against torch CPU fp32.
Shapes follow SURVEY.md Appendix A at reduced batch.  Run with `pytest -m gpu` on an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from util import TOL_BIAS, TOL_FWD, TOL_WGRAD, Launches, assert_close, rel_fro

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pg():
    import pytorch_gan_amd as pg

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return pg


def _leaf(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _ref_gather(x, pads, gather):
    pt, pl, pb, pr = pads
    if gather == 2:
        x = TF.interpolate(x, scale_factor=2, mode="nearest")
    if gather == 1:
        return TF.pad(x, (pl, pr, pt, pb), mode="reflect")
    return TF.pad(x, (pl, pr, pt, pb))


CONV_CASES = [
    # N, Ci, H, W, Co, k, stride, pads(t,l,b,r), gather, act, bias
    (2, 128, 16, 16, 128, 3, 1, (1, 1, 1, 1), 0, 0, True),      # dcgan.py:55 / fast path 128x128 tile
    (2, 128, 16, 16, 64, 3, 1, (1, 1, 1, 1), 2, 1, True),       # Upsample+conv+LeakyReLU fused (dcgan.py:58-61)
    (2, 64, 16, 16, 1, 3, 1, (1, 1, 1, 1), 0, 3, True),         # dcgan.py:62 + Tanh, Co=1
    (4, 1, 32, 32, 16, 3, 2, (1, 1, 1, 1), 0, 1, True),         # dcgan.py:78 first D block (Ci=1, generic path)
    (4, 16, 16, 16, 32, 3, 2, (1, 1, 1, 1), 0, 0, True),        # Ci=16 generic
    (2, 64, 16, 16, 128, 3, 2, (1, 1, 1, 1), 0, 0, True),       # cyclegan/models.py:60
    (1, 256, 12, 12, 256, 3, 1, (1, 1, 1, 1), 1, 0, True),      # ResidualBlock: ReflectionPad2d(1)+conv
    (1, 3, 20, 20, 64, 7, 1, (3, 3, 3, 3), 1, 0, True),         # c7s1-64 with reflection pad 3 (generic)
    (1, 64, 20, 20, 3, 7, 1, (3, 3, 3, 3), 1, 3, True),         # c7s1-3 + Tanh
    (2, 3, 32, 32, 64, 4, 2, (1, 1, 1, 1), 0, 1, True),         # PatchGAN first block
    (2, 64, 16, 16, 128, 4, 2, (1, 1, 1, 1), 0, 0, False),      # pix2pix UNetDown (bias=False)
    (2, 512, 6, 6, 1, 4, 1, (2, 2, 1, 1), 0, 0, True),          # ZeroPad2d((1,0,1,0)) + Conv2d(512,1,4,padding=1)
    (1, 128, 8, 8, 3, 4, 1, (2, 2, 1, 1), 2, 3, True),          # pix2pix final: Upsample+ZeroPad+conv+Tanh
    (1, 3, 12, 12, 64, 9, 1, (4, 4, 4, 4), 0, 0, True),         # srgan conv1 9x9
    (1, 64, 12, 12, 3, 9, 1, (4, 4, 4, 4), 0, 3, True),         # srgan conv3 9x9 + Tanh
    (3, 32, 9, 7, 48, 3, 1, (1, 1, 1, 1), 0, 2, True),          # ragged sizes, M/N tails
    (2, 96, 5, 5, 160, 3, 2, (1, 1, 1, 1), 0, 0, True),         # odd spatial with stride 2 (parity classes uneven)
    (2, 40, 10, 10, 72, 3, 1, (1, 1, 1, 1), 0, 1, True),        # Ci % 32 != 0: K-tail variant, 2 chunks per tap
    (2, 8, 12, 12, 24, 4, 2, (1, 1, 1, 1), 1, 0, True),         # Ci = 8: K-tail with a single masked chunk, reflect
    (2, 100, 6, 6, 36, 3, 1, (1, 1, 1, 1), 0, 0, False),        # Ci = 100 (4 chunks, tail of 4 channels)
    (3, 1, 14, 14, 8, 3, 1, (1, 1, 1, 1), 0, 2, True),          # small-K direct kernel, stride 1
    (2, 64, 20, 18, 32, 3, 1, (1, 1, 1, 1), 1, 2, True),        # reflect pad 1, ragged spatial (carry logic of wgrad_inc)
    (2, 64, 11, 13, 64, 7, 1, (3, 3, 3, 3), 1, 0, False),       # reflect pad 3, 7x7 (49 taps), 64x128 wgrad tiles
    (5, 32, 6, 6, 160, 4, 2, (1, 1, 1, 1), 0, 0, True),         # 4x4 s2 with Ho*Wo = 9 < 32: several images per K-tile
    (2, 32, 448, 448, 48, 3, 1, (1, 1, 1, 1), 0, 0, True),      # 401k-pixel GEMM (3136 M-tiles), N tail; no act (kink flips)
    (16, 64, 96, 96, 64, 3, 1, (1, 1, 1, 1), 0, 0, True),       # srgan/models.py:22-27 trunk at the bench batch (2304 tiles of 64 x 64)
    (2, 32, 12, 12, 38, 3, 1, (1, 1, 1, 1), 0, 1, True),        # Co % 4 != 0 on the LDS-DMA kernels: per-channel form of the quad epilogue
    # under-filled GEMMs (pix2pix/models.py:62-71): M = 1, 4, 16, 64 output pixels against K = 8192 / 4096 / 2048:
    # LDS-DMA kernel with four stages, split-K over up to 64 slices, in-kernel ticket reduction + epilogue
    (1, 512, 2, 2, 512, 4, 2, (1, 1, 1, 1), 0, 1, False),       # M = 1
    (1, 512, 4, 4, 512, 4, 2, (1, 1, 1, 1), 0, 0, True),        # M = 4
    (1, 256, 8, 8, 512, 4, 2, (1, 1, 1, 1), 0, 2, True),        # M = 16
    (1, 128, 16, 16, 256, 4, 2, (1, 1, 1, 1), 0, 1, False),     # M = 64
    (3, 72, 8, 8, 96, 3, 1, (1, 1, 1, 1), 0, 0, True),          # split-K with a K-tail (Ci % 32 != 0), ragged N, M = 192
    (8, 64, 8, 8, 128, 3, 2, (1, 1, 1, 1), 0, 1, True),         # dcgan.py:78 last D block shape (M = 128, K = 576)
    # first convs of the image nets (3 / 6 source channels, 16 < taps * Ci <= 128): the mid-K direct kernel
    (1, 6, 32, 32, 64, 4, 2, (1, 1, 1, 1), 0, 1, False),        # pix2pix/models.py:115 Conv2d(6, 64, 4, 2, 1), K = 96
    (2, 3, 24, 24, 64, 3, 1, (1, 1, 1, 1), 0, 1, True),         # srgan/models.py:85 Conv2d(3, 64, 3, 1, 1), K = 27
    (3, 3, 9, 7, 16, 3, 1, (1, 1, 1, 1), 1, 2, True),           # ragged M (189 pixels), reflection pad, Co = 16
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(pg, case):
    N, Ci, H, W, Co, k, stride, pads, gather, act, bias = case
    F = pg.functional
    x = _leaf(N, Ci, H, W, seed=1).requires_grad_(True)
    w = _leaf(Co, Ci, k, k, seed=2, scale=0.2).requires_grad_(True)
    b = _leaf(Co, seed=3).requires_grad_(True) if bias else None
    y_ref = TF.conv2d(_ref_gather(x, pads, gather), w, b, stride)
    y_ref = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu, 3: torch.tanh}[act](y_ref)
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)

    xg = x.detach().to(DEV).requires_grad_(True)
    wg = w.detach().to(DEV).requires_grad_(True)
    bg = b.detach().to(DEV).requires_grad_(True) if bias else None
    y = F.conv2d(xg, wg, bg, stride, pads, gather, act, 0.2)
    assert y.shape == y_ref.shape
    assert y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.to(DEV))
    assert_close(y, y_ref, TOL_FWD, "conv fwd")
    assert_close(xg.grad, x.grad, TOL_FWD, "conv dgrad")
    assert_close(wg.grad, w.grad, TOL_WGRAD, "conv wgrad")
    if bias:
        assert_close(bg.grad, b.grad, TOL_BIAS, "conv bias grad")


TAP9_CASES = [   # (more than 128 tiles of 64 x 64: below that the split-K form of the kernel - tap-outer only - takes the layer)
    (2, 64, 48, 48, 128, 1, 1, "ReflectionPad2d(1)+Conv3x3 (cyclegan/models.py:26-33)"),
    (2, 64, 48, 48, 128, 1, 0, "zero pad"),
    (2, 64, 96, 96, 128, 2, 0, "srgan/models.py:89 stride-2 block"),
    (3, 96, 40, 36, 80, 1, 0, "ragged M / N"),
]


@pytest.mark.parametrize("case", TAP9_CASES, ids=["ci%d_co%d_s%d_g%d" % (c[1], c[4], c[5], c[6]) for c in TAP9_CASES])
def test_conv3x3_tap_inner_k_order(pg, case, monkeypatch):
    """3x3 layers on the LDS-DMA kernel in both K orders - tap-outer (MIGAN_DMA_TAPS9=0) and channel-chunk outer / nine taps inner (= 2: also
    below the 65 536-pixel gate of production) - forward and input gradient against torch, and the launch counters say the nine-tap
    instantiation really served the second run (csrc/conv_dma.hip TAPS_IN = 9; dcgan / cyclegan / srgan / vgg19 3x3 layers at size)."""
    N, Ci, H, W, Co, stride, gather, _ = case
    F = pg.functional
    pads = (1, 1, 1, 1)
    x = _leaf(N, Ci, H, W, seed=1).requires_grad_(True)
    w = _leaf(Co, Ci, 3, 3, seed=2, scale=0.2).requires_grad_(True)
    y_ref = TF.conv2d(_ref_gather(x, pads, gather), w, None, stride)
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    for knob in ("0", "2"):
        monkeypatch.setenv("MIGAN_DMA_TAPS9", knob)
        xg, wg = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
        with Launches() as nl:
            y = F.conv2d(xg, wg, None, stride, pads, gather, 0, 0.0)
            y.backward(gy.to(DEV))
            nine = nl(", 9, false")   # igemm_dma_kernel<BM, BN, WM, WN, BK, 9, false, ...>
        assert (nine >= 1) == (knob == "2"), (knob, nine)
        assert_close(y, y_ref, TOL_FWD, "conv fwd, TAPS9=" + knob)
        assert_close(xg.grad, x.grad, TOL_FWD, "conv dgrad, TAPS9=" + knob)
        assert_close(wg.grad, w.grad, TOL_WGRAD, "conv wgrad, TAPS9=" + knob)


RGB_CASES = [
    # N, H, W, Co, k, pad, gather, act, what
    (1, 110, 150, 64, 3, 1, 0, 1, "srgan/models.py:85 Conv2d(3,64,3,1,1)+LeakyReLU, ragged row tiles"),
    (2, 96, 128, 64, 3, 1, 0, 2, "vgg19.features[0:2] Conv2d(3,64,3,padding=1)+ReLU"),
    (1, 128, 130, 32, 3, 1, 0, 0, "32 output channels, no activation"),
    (1, 128, 128, 64, 7, 3, 1, 0, "cyclegan/models.py:49-50 ReflectionPad2d(3)+Conv2d(3,64,7)"),
]


@pytest.mark.parametrize("case", RGB_CASES, ids=["k%d_co%d_g%d_a%d" % (c[4], c[3], c[6], c[7]) for c in RGB_CASES])
def test_rgb_conv_layers(pg, case):
    """The image-input layers (3 source channels) on csrc/rgb_conv.hip against torch on the host: forward; the weights-only backward
    (the discriminator's first conv in its own update: activation backward + bias column sums + weight gradient in ONE launch, asserted
    with the launch counters); the backward with an input gradient as well (general path) - same results either way."""
    N, H, W, Co, k, pad, gather, act, _ = case
    F = pg.functional
    pads = (pad,) * 4
    x = _leaf(N, 3, H, W, seed=1).requires_grad_(True)
    w = _leaf(Co, 3, k, k, seed=2, scale=0.2).requires_grad_(True)
    b = _leaf(Co, seed=3).requires_grad_(True)
    y_ref = TF.conv2d(_ref_gather(x, pads, gather), w, b, 1)
    y_ref = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu}[act](y_ref)
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    fused_ok = k <= 7
    for want_dx in (False, True):
        xg = x.detach().to(DEV).requires_grad_(want_dx)
        wg = w.detach().to(DEV).requires_grad_(True)
        bg = b.detach().to(DEV).requires_grad_(True)
        with Launches() as n:
            y = F.conv2d(xg, wg, bg, 1, pads, gather, act, 0.2)
            assert n("rgb_conv_fwd_kernel") == 1, "forward not on the image-input kernel"
            y.backward(gy.to(DEV))
            fused = n("rgb_conv_wgrad_kernel")
            assert fused == (1 if (fused_ok and not want_dx) else 0), (want_dx, fused)
            if fused:
                assert n("act_bwd") == 0 and n("colsum") == 0, "an activation-backward / column-sum pass ran beside the fused launches"
        assert_close(y, y_ref, TOL_FWD, "rgb conv fwd")
        assert_close(wg.grad, w.grad, TOL_WGRAD, "rgb conv wgrad (dx %s)" % want_dx)
        assert_close(bg.grad, b.grad, TOL_BIAS, "rgb conv bias grad (dx %s)" % want_dx)
        if want_dx:
            assert_close(xg.grad, x.grad, TOL_FWD, "rgb conv dgrad")
    # gradient slots (the optimiser's bucket): the fused launch adds into them
    if fused_ok:
        wg = w.detach().to(DEV).requires_grad_(True)
        bg = b.detach().to(DEV).requires_grad_(True)
        wg.grad, bg.grad = torch.full_like(wg, 0.25), torch.full_like(bg, -0.5)
        with torch.no_grad():
            pass
        y = F.conv2d(x.detach().to(DEV), wg, bg, 1, pads, gather, act, 0.2)
        y.backward(gy.to(DEV))
        assert_close(wg.grad - 0.25, w.grad, TOL_WGRAD, "rgb conv wgrad into a slot")
        assert_close(bg.grad + 0.5, b.grad, TOL_BIAS, "rgb conv bias grad into a slot")


@pytest.mark.parametrize("case", [(2, 64, 128, 1, 3), (1, 100, 170, 3, 0), (3, 48, 130, 2, 1)], ids=["dcgan_g_conv3", "ch3_ragged", "two_ch"])
def test_thin_output_3x3_conv(pg, case):
    """dcgan.py:62 Conv2d(64, channels, 3, stride=1, padding=1) + Tanh against torch; and its INPUT gradient - 64 channels back from 1 or 3 -
    also through the C ABI's image-input forward kernel with the taps reversed (migan_rgb_conv_fwd, flip = 1)."""
    N, H, W, Co, act = case
    F = pg.functional
    x = _leaf(N, 64, H, W, seed=1).requires_grad_(True)
    w = _leaf(Co, 64, 3, 3, seed=2, scale=0.1).requires_grad_(True)
    b = _leaf(Co, seed=3).requires_grad_(True)
    y_ref = TF.conv2d(x, w, b, 1, 1)
    y_ref = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, 0.2), 3: torch.tanh}[act](y_ref)
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    xg, wg, bg = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    y = F.conv2d(xg, wg, bg, 1, (1, 1, 1, 1), 0, act, 0.2)
    y.backward(gy.to(DEV))
    if Co in (1, 3) and act == 0 or Co == 1:
        # the C ABI's other form of this input gradient: the image-input forward kernel on dy with reversed taps (migan_rgb_conv_fwd, flip = 1)
        from pytorch_gan_amd._lib import check, lib

        g_pre = gy if act == 0 else gy * (1 - y_ref.detach() ** 2)   # through Tanh for the dcgan.py:62 case
        dyd = F.to_nhwc(g_pre.to(DEV))
        wk = w.detach().permute(2, 3, 0, 1).contiguous().to(DEV)
        dx = torch.empty((N, 64, H, W), device=DEV).contiguous(memory_format=torch.channels_last)
        assert lib.migan_rgb_conv_ok(Co, 64, 3, 3, 1, 0, N * H * W) == 1
        check(lib.migan_rgb_conv_fwd(dyd.data_ptr(), wk.data_ptr(), None, dx.data_ptr(), N, H, W, Co, H, W, 64, 3, 3, 1, 1, 0, 0, 0.0, 1,
                                     torch.cuda.current_stream().cuda_stream), "rgb_conv_fwd flip")
        assert_close(dx, x.grad, TOL_FWD, "thin-output dgrad on the image-input kernel (taps reversed)")
    assert_close(y, y_ref, TOL_FWD, "thin-output fwd")
    assert_close(xg.grad, x.grad, TOL_FWD, "thin-output dgrad")
    assert_close(wg.grad, w.grad, TOL_WGRAD, "thin-output wgrad")
    assert_close(bg.grad, b.grad, 2.5 * TOL_BIAS, "thin-output bias grad")   # 17 000-49 000-term column sums in two summation orders


def _kernel_cases():
    import kernel_cases

    return list(kernel_cases.all_cases())


@pytest.mark.parametrize("case", _kernel_cases(), ids=[c[0] for c in _kernel_cases()])
def test_geometry_selects_kernel(pg, case):
    """Dispatch inside the library is by geometry alone: each shape of tests/kernel_cases.py must be served by the specialised
    kernel written for it (launch counters of the C ABI) AND agree with torch on the host."""
    import kernel_cases

    name, syms, inputs, run, ref = case
    F = pg.functional
    gen = torch.Generator(device="cpu")
    gen.manual_seed(1234)
    tensors = inputs(gen)
    want = ref(tensors)
    saved = kernel_cases.quiet_scope(F)
    try:
        with torch.enable_grad(), Launches() as n:
            got = run(F, tuple(None if t is None else t.to(DEV) for t in tensors))
            for sym in syms:
                assert n(sym) > 0, "%s: no launch of %s" % (name, sym)
    finally:
        kernel_cases.restore_scope(F, saved)
    kink = float((want["_pre"].abs() > 1e-5).float().min()) == 0.0   # an activation input within rounding of 0 may take the other branch
    for k, v in got.items():
        tol = {"y": TOL_FWD, "dx": 5e-5 if name.startswith("in_") else TOL_FWD, "db": TOL_BIAS}.get(k, TOL_WGRAD)
        assert_close(v, want[k], 1e-3 if (kink and k != "y") else tol, "%s %s" % (name, k))


def test_splitk_conv_is_deterministic(pg):
    """The in-kernel split-K reduction adds the slabs in slice order whoever arrives last: runs are bit-identical, and the
    tickets are back at zero afterwards (pix2pix/models.py:63 geometry: 64 output pixels, K = 2048, 512 k weights - below the few-pixel
    path's 1 M-weight threshold, so the split-K tile kernel serves it: asserted with the launch counters)."""
    F = pg.functional
    x = _leaf(1, 128, 16, 16, seed=11).to(DEV)
    w = _leaf(256, 128, 4, 4, seed=12, scale=0.05).to(DEV)
    F._SK_WS.clear()
    with Launches() as n:
        outs = [F.conv2d(x, w, None, 2, (1, 1, 1, 1), 0, 0, 0.0).clone() for _ in range(4)]
        assert n("4, true>") == 4 and n("im2col_small") == 0, "the split-K kernel (NS = 4, SPLITK) must serve this shape"
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert F._SK_WS, "the split-K workspace was not requested"
    for ws in F._SK_WS.values():
        assert int(ws[:1024].view(torch.int32).abs().sum()) == 0
    ref = TF.conv2d(x.cpu(), w.cpu(), None, 2, 1)
    assert_close(outs[0], ref, TOL_FWD, "split-K conv")


TOEP_CASES = [
    # N, Ci, H, W, (Co, R, S), pads(t,l,b,r), gather, act, bias
    (1, 64, 20, 20, (3, 7, 7), (3, 3, 3, 3), 1, 3, True),     # cyclegan/models.py:82 c7s1-3: ReflectionPad2d(3) + 7x7 + Tanh
    (2, 64, 12, 12, (3, 9, 9), (4, 4, 4, 4), 0, 3, True),     # srgan/models.py:62 conv3 9x9 + Tanh
    (2, 32, 70, 75, (3, 7, 7), (3, 3, 3, 3), 0, 0, True),     # W > 64: two column tiles per row, ragged second tile
    (1, 16, 10, 67, (4, 5, 5), (2, 1, 2, 3), 0, 1, False),    # Co = 4, asymmetric zero padding
    (1, 16, 9, 12, (3, 7, 7), (3, 2, 3, 1), 1, 0, True),      # reflection, asymmetric, Wo < W
    (2, 24, 11, 13, (3, 3, 7), (1, 3, 1, 3), 1, 2, True),     # R != S, Ci % 32 != 0
    (2, 64, 24, 32, (3, 9, 9), (4, 4, 4, 4), 0, 3, True),     # 64 channels, W % 16 == 0: the weight gradient walks column strips (LDS ring)
    (1, 64, 40, 16, (3, 9, 9), (4, 4, 4, 4), 0, 0, False),    # ... one strip cut into two row segments (halo rows re-fetched)
    (1, 64, 24, 48, (3, 7, 7), (3, 3, 3, 3), 1, 3, True),     # ... 7x7 under reflection padding (virtual rows mirrored); 8 x 16 M blocks
]


@pytest.mark.parametrize("case", TOEP_CASES)
def test_thin_toeplitz_conv(pg, case, monkeypatch):
    """Thin-N conv through the width-Toeplitz expansion (csrc/thin_toeplitz.hip) against torch CPU and against the direct
    VALU kernels it replaces; forward, input, weight and bias gradients."""
    N, Ci, H, W, (Co, R, S), pads, gather, act, bias = case
    F = pg.functional
    from pytorch_gan_amd._lib import lib

    assert lib.migan_thin_toeplitz_ok(Co, R, S, Ci, 1, gather) == 1
    monkeypatch.setattr(F, "_TOEP_MIN_PIXELS", 0)
    x = _leaf(N, Ci, H, W, seed=1).requires_grad_(True)
    w = _leaf(Co, Ci, R, S, seed=2, scale=0.2).requires_grad_(True)
    b = _leaf(Co, seed=3).requires_grad_(True) if bias else None
    y_ref = TF.conv2d(_ref_gather(x, pads, gather), w, b, 1)
    y_ref = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu, 3: torch.tanh}[act](y_ref)
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(F, "_TOEPLITZ", on)
        xg = x.detach().to(DEV).requires_grad_(True)
        wg = w.detach().to(DEV).requires_grad_(True)
        bg = b.detach().to(DEV).requires_grad_(True) if bias else None
        y = F.conv2d(xg, wg, bg, 1, pads, gather, act, 0.2)
        assert y.grad_fn.toep is on
        with Launches() as nl:
            y.backward(gy.to(DEV))
            if on and Ci == 64 and R == S and R in (7, 9) and W % 16 == 0:
                assert nl("toep_wgrad_ring_kernel") == 1, "the strip-walking weight gradient did not run"
        assert_close(y, y_ref, TOL_FWD, "toeplitz=%s fwd" % on)
        assert_close(xg.grad, x.grad, TOL_FWD, "toeplitz=%s dgrad" % on)
        assert_close(wg.grad, w.grad, TOL_WGRAD, "toeplitz=%s wgrad" % on)
        if bias:
            assert_close(bg.grad, b.grad, TOL_BIAS, "toeplitz=%s bias grad" % on)
        outs[on] = (y.detach(), xg.grad, wg.grad)
    for a, c, what in zip(outs[True], outs[False], ("fwd", "dgrad", "wgrad")):
        assert_close(a, c, 4e-6, "toeplitz vs direct " + what)


def test_pack_plan_one_launch_per_step(pg, monkeypatch):
    """functional._PackPlan: the OHWI / IHWO weight packs a step asked for are produced by ONE migan_multi_permute4d launch
    at the start of the next step; a weight whose optimiser steps in the middle of a step is re-packed on its next use."""
    F = pg.functional
    from pytorch_gan_amd import optim
    from pytorch_gan_amd._lib import lib

    w1 = torch.nn.Parameter(_leaf(32, 16, 3, 3, seed=1, scale=0.1).to(DEV))
    w2 = torch.nn.Parameter(_leaf(8, 32, 4, 4, seed=2, scale=0.1).to(DEV))
    opt = optim.Adam([w1, w2], lr=1e-2)
    x = _leaf(2, 16, 12, 12, seed=3)
    singles, multis = [], []
    orig_p, orig_m = lib.migan_permute4d, lib.migan_multi_permute4d
    monkeypatch.setattr(lib, "migan_permute4d", lambda *a: (singles.append(1), orig_p(*a))[1])
    monkeypatch.setattr(lib, "migan_multi_permute4d", lambda *a: (multis.append(a[2]), orig_m(*a))[1])

    def net(xin, a, b):
        return F.conv2d(F.conv2d(xin, a, None, 1, (1, 1, 1, 1), 0, 1, 0.2), b, None, 2, (1, 1, 1, 1))

    def ref(a, b):
        xr = x.clone().requires_grad_(True)
        y = TF.conv2d(TF.leaky_relu(TF.conv2d(xr, a.detach().cpu(), None, 1, 1), 0.2), b.detach().cpu(), None, 2, 1)
        y.sum().backward()
        return y.detach(), xr.grad

    def step(update_in_the_middle=False):
        with F.weight_cache_scope():
            opt.zero_grad()
            xg = x.to(DEV).requires_grad_(True)
            y = net(xg, w1, w2)
            y.sum().backward()
            if update_in_the_middle:
                opt.step()          # new weights, new epoch: the pre-filled packs of this scope are stale now
                y2 = net(x.to(DEV), w1, w2)
                return y.detach(), xg.grad, y2.detach()
            return y.detach(), xg.grad

    y_ref, g_ref = ref(w1, w2)
    y, g = step()
    assert len(singles) == 4 and not multis          # first step: nothing planned yet
    assert_close(y, y_ref, TOL_FWD, "step 1 fwd")
    singles.clear()
    y, g = step()
    assert len(multis) == 1 and not singles          # second step: one launch for all four packs
    assert_close(y, y_ref, TOL_FWD, "planned packs fwd")
    assert_close(g, g_ref, TOL_FWD, "planned packs dgrad")
    singles.clear()
    multis.clear()
    y, g, y_after = step(update_in_the_middle=True)
    assert len(multis) == 1 and len(singles) == 2    # the two OHWI packs are redone after the update
    y_new, _ = ref(w1, w2)
    assert_close(y_after, y_new, TOL_FWD, "forward after a mid-step optimiser update")
    assert rel_fro(y_new, y_ref) > 1e-3                # (the update did change the output)
    singles.clear()
    multis.clear()
    monkeypatch.setattr(F, "_BATCH_PACKS", False)
    y_off, g_off = step()
    assert not multis and len(singles) == 4
    monkeypatch.setattr(F, "_BATCH_PACKS", True)
    y_on, g_on = step()
    assert torch.equal(y_on, y_off) and torch.equal(g_on, g_off)


def test_conv2d_nchw_input_is_relaid(pg):
    """An NCHW-contiguous activation (as produced by `.view` in dcgan.py:68) is accepted and re-laid out."""
    F = pg.functional
    x = _leaf(2, 64, 8, 8, seed=5)
    w = _leaf(32, 64, 3, 3, seed=6, scale=0.1)
    y = F.conv2d(x.to(DEV), w.to(DEV), None, 1, (1, 1, 1, 1))
    assert_close(y, TF.conv2d(x, w, None, 1, 1), TOL_FWD, "nchw in")


@pytest.mark.parametrize("shape", [(2, 64, 8, 8, 32, True), (1, 512, 1, 1, 512, False), (2, 128, 5, 7, 64, False)])
def test_conv_transpose2d(pg, shape):
    N, Cin, H, W, Cout, bias = shape
    F = pg.functional
    x = _leaf(N, Cin, H, W, seed=1).requires_grad_(True)
    w = _leaf(Cin, Cout, 4, 4, seed=2, scale=0.1).requires_grad_(True)
    b = _leaf(Cout, seed=3).requires_grad_(True) if bias else None
    y_ref = TF.conv_transpose2d(x, w, b, 2, 1)
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    xg, wg = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
    bg = b.detach().to(DEV).requires_grad_(True) if bias else None
    y = F.conv_transpose2d(xg, wg, bg, 2, 1)
    y.backward(gy.to(DEV))
    assert_close(y, y_ref, TOL_FWD, "convT fwd")
    assert_close(xg.grad, x.grad, TOL_FWD, "convT dgrad")
    assert_close(wg.grad, w.grad, TOL_WGRAD, "convT wgrad")
    if bias:
        assert_close(bg.grad, b.grad, TOL_WGRAD, "convT bias")


# N, Ci, H, W, Co, bias, act - Conv2d(Ci, Co, 4, 2, 1) with <= 64 output pixels and >= 1 M weights: the inner U-Net levels of
# pix2pix/models.py:62-67 at batch 1 (d5 ... d8) and scaled-down relatives: the few-pixel path of csrc/fewpix.hip (asserted)
FEWPIX_CONV = [(1, 512, 16, 16, 512, False, 0), (1, 512, 8, 8, 512, False, 1), (1, 512, 2, 2, 512, False, 1), (2, 256, 4, 4, 256, True, 1),
               (1, 128, 8, 8, 512, True, 0), (3, 64, 6, 10, 1024, True, 2)]


@pytest.mark.parametrize("cfg", FEWPIX_CONV, ids=["%dx%dx%dx%d-%d" % c[:5] for c in FEWPIX_CONV])
def test_fewpix_conv2d(pg, cfg):
    N, Ci, H, W, Co, bias, act = cfg
    F = pg.functional
    x = _leaf(N, Ci, H, W, seed=1).requires_grad_(True)
    w = _leaf(Co, Ci, 4, 4, seed=2, scale=0.05).requires_grad_(True)
    b = _leaf(Co, seed=3).requires_grad_(True) if bias else None
    z_ref = TF.conv2d(x, w, b, 2, 1)
    y_ref = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu}[act](z_ref)
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    xg, wg = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
    bg = b.detach().to(DEV).requires_grad_(True) if bias else None
    with Launches() as n:
        y = F.conv2d(xg, wg, bg, 2, (1, 1, 1, 1), F.GATHER_ZERO, act, 0.2)
        y.backward(gy.to(DEV))
        # forward im2col + NT product, input gradient NN + col2im, weight gradient TN straight into the bucket: no tiled kernel, no pack
        assert n("im2col_small_kernel") == 1 and n("col2im_small_kernel") == 1 and n("skinny_nn_kernel") == 1 and n("skinny_tn_kernel") == 1
        assert n("fewpix_nt_kernel") + n("skinny_nt_kernel") == 1
        assert n("igemm") == 0 and n("wgrad") == 0 and n("permute4") == 0 and n("pack_transpose") == 0
    assert_close(y, y_ref, TOL_FWD, "fewpix conv fwd")
    keep = (z_ref.detach().abs() > 1e-5).float()   # pre-activations at the kink may take the other branch
    if float(keep.min()) == 1.0:
        assert_close(xg.grad, x.grad, TOL_FWD, "fewpix conv dgrad")
        assert_close(wg.grad, w.grad, TOL_WGRAD, "fewpix conv wgrad")
        if bias:
            assert_close(bg.grad, b.grad, TOL_WGRAD, "fewpix conv bias")
    else:
        assert_close(xg.grad, x.grad, 1e-3, "fewpix conv dgrad (kink)")
        assert_close(wg.grad, w.grad, 1e-3, "fewpix conv wgrad (kink)")


# N, Cin, H, W, Cout, bias, act - ConvTranspose2d(Cin, Cout, 4, 2, 1) with <= 64 INPUT pixels: u1 ... u4 of pix2pix/models.py:68-71
FEWPIX_CONVT = [(1, 1024, 2, 2, 512, False, 0), (1, 1024, 8, 8, 512, False, 0), (1, 512, 8, 8, 256, True, 2), (4, 256, 4, 4, 256, True, 0),
                (2, 128, 5, 6, 512, False, 0)]


@pytest.mark.parametrize("cfg", FEWPIX_CONVT, ids=["%dx%dx%dx%d-%d" % c[:5] for c in FEWPIX_CONVT])
def test_fewpix_conv_transpose2d(pg, cfg):
    N, Cin, H, W, Cout, bias, act = cfg
    F = pg.functional
    x = _leaf(N, Cin, H, W, seed=1).requires_grad_(True)
    w = _leaf(Cin, Cout, 4, 4, seed=2, scale=0.05).requires_grad_(True)
    b = _leaf(Cout, seed=3).requires_grad_(True) if bias else None
    z_ref = TF.conv_transpose2d(x, w, b, 2, 1)
    y_ref = torch.relu(z_ref) if act == 2 else z_ref
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    xg, wg = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
    bg = b.detach().to(DEV).requires_grad_(True) if bias else None
    with Launches() as n:
        y = F.conv_transpose2d(xg, wg, bg, 2, 1, act, 0.0)
        y.backward(gy.to(DEV))
        assert n("im2col_small_kernel") == 1 and n("col2im_small_kernel") == 1 and n("skinny_tn_kernel") == 1
        assert n("igemm") == 0 and n("wgrad") == 0 and n("permute4") == 0 and n("pack_transpose") == 0
    assert_close(y, y_ref, TOL_FWD, "fewpix convT fwd")
    tol_g = (TOL_FWD, TOL_WGRAD) if float((z_ref.detach().abs() > 1e-5).float().min()) == 1.0 or act == 0 else (1e-3, 1e-3)
    assert_close(xg.grad, x.grad, tol_g[0], "fewpix convT dgrad")
    assert_close(wg.grad, w.grad, tol_g[1], "fewpix convT wgrad")
    if bias:
        assert_close(bg.grad, b.grad, max(TOL_WGRAD, tol_g[1]), "fewpix convT bias")


@pytest.mark.parametrize("dims", [(128, 100, 8192), (64, 1024, 512), (64, 256, 1), (7, 33, 5), (128, 2048, 1)])
def test_linear(pg, dims):
    B, K, Nf = dims
    F = pg.functional
    x = _leaf(B, K, seed=1).requires_grad_(True)
    w = _leaf(Nf, K, seed=2, scale=0.1).requires_grad_(True)
    b = _leaf(Nf, seed=3).requires_grad_(True)
    y_ref = TF.linear(x, w, b)
    gy = _leaf(B, Nf, seed=4)
    y_ref.backward(gy)
    xg, wg, bg = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    y = F.linear(xg, wg, bg)
    y.backward(gy.to(DEV))
    assert_close(y, y_ref, TOL_FWD, "linear fwd")
    assert_close(xg.grad, x.grad, TOL_FWD, "linear dgrad")
    assert_close(wg.grad, w.grad, TOL_WGRAD, "linear wgrad")
    assert_close(bg.grad, b.grad, TOL_WGRAD, "linear bias")


def test_linear_lrelu_double_backward(pg):
    """The WGAN-GP pattern (wgan_gp.py:119-138): grad of D(x) w.r.t. x with create_graph, then backward of a
    function of that grad into the weights."""
    F = pg.functional
    B = 16
    x = _leaf(B, 64, seed=1)
    W1, b1 = _leaf(48, 64, seed=2, scale=0.3), _leaf(48, seed=3)
    W2, b2 = _leaf(1, 48, seed=4, scale=0.3), _leaf(1, seed=5)

    def run(dev, lin, lrelu, norm_pen):
        xs = x.detach().clone().to(dev).requires_grad_(True)
        ps = [t.detach().clone().to(dev).requires_grad_(True) for t in (W1, b1, W2, b2)]
        out = lin(lrelu(lin(xs, ps[0], ps[1])), ps[2], ps[3])
        g = torch.autograd.grad(out, xs, torch.ones(B, 1, device=dev), create_graph=True, retain_graph=True)[0]
        pen = norm_pen(g)
        pen.backward()
        return pen, ps

    pen_r, ps_r = run("cpu", TF.linear, lambda t: TF.leaky_relu(t, 0.2), lambda g: ((g.norm(2, dim=1) - 1) ** 2).mean())
    pen_g, ps_g = run(DEV, F.linear, lambda t: F.activation(t, F.ACT_LRELU, 0.2),
                      lambda g: F.loss(F.LOSS_MSE, F.rownorm(g), None, 1.0))
    # the same with LeakyReLU in the GEMM epilogue (what nn.Sequential does with Linear -> LeakyReLU, wgan_gp.py:73-77)
    fused_first = [True]

    def lin_fused(t, w, b):
        if fused_first[0]:
            fused_first[0] = False
            return F.linear(t, w, b, F.ACT_LRELU, 0.2)
        return F.linear(t, w, b)

    pen_f, ps_f = run(DEV, lin_fused, lambda t: t, lambda g: F.loss(F.LOSS_MSE, F.rownorm(g), None, 1.0))
    assert abs(pen_f.item() - pen_g.item()) <= 1e-6 * max(1.0, abs(pen_g.item()))
    assert_close(ps_f[0].grad, ps_g[0].grad, 1e-6, "dW1 of penalty, fused vs separate activation")
    assert_close(ps_f[2].grad, ps_g[2].grad, 1e-6, "dW2 of penalty, fused vs separate activation")
    assert abs(pen_r.item() - pen_g.item()) <= 1e-5 * max(1.0, abs(pen_r.item()))
    assert_close(ps_g[0].grad, ps_r[0].grad, TOL_WGRAD, "dW1 of penalty")
    assert_close(ps_g[2].grad, ps_r[2].grad, TOL_WGRAD, "dW2 of penalty")
    assert ps_r[3].grad is None and ps_g[3].grad is None  # last-layer bias gets no gradient from the penalty
    # hidden-layer bias: torch materialises exact zeros; the HIP path leaves it untouched (None == zeros in the
    # optimiser's pre-zeroed flat bucket)
    assert float(ps_r[1].grad.abs().max()) == 0.0
    assert ps_g[1].grad is None or float(ps_g[1].grad.abs().max()) == 0.0


# the last two are HBM-sized (> 2^21 elements, 1024 statistic chunks)
@pytest.mark.parametrize("cfg", [(8, 128, 16, 16, 0.8, 1), (8, 64, 8, 8, 1e-5, 0), (4, 32, 5, 3, 0.8, 2), (16, 10, 4, 4, 1e-5, 1),
                                 (8, 64, 72, 72, 1e-5, 0), (6, 10, 200, 190, 0.8, 0)])  # big ones without activation (kink flips)
def test_batchnorm2d_train(pg, cfg):
    N, C, H, W, eps, act = cfg
    F = pg.functional
    x = (_leaf(N, C, H, W, seed=1) * 2 + 0.3).requires_grad_(True)
    gamma, beta = (_leaf(C, seed=2) + 1.5).requires_grad_(True), _leaf(C, seed=3).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    z_ref = TF.batch_norm(x, rm, rv, gamma, beta, True, 0.1, eps)
    y_ref = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu}[act](z_ref)
    gy = _leaf(N, C, H, W, seed=4)
    y_ref.backward(gy)
    xg, gg, bg = (t.detach().to(DEV).requires_grad_(True) for t in (x, gamma, beta))
    rmg, rvg = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    y = F.norm(xg, gg, bg, None, rmg, rvg, True, 0.1, eps, False, act, 0.2)
    y.backward(gy.to(DEV))
    assert_close(y, y_ref, TOL_FWD, "bn fwd")
    # The fused activation derivative is taken from the recomputed pre-activation; an element within rounding of the
    # kink may land on the other side than torch's stored output (1 of 2.6M elements moves rel_fro by 4e-4), so such
    # elements are excluded from the dx comparison.
    keep = (z_ref.detach().abs() > 1e-5).float()
    assert_close(xg.grad.cpu() * keep, x.grad * keep, 2e-5, "bn dx")
    assert_close(gg.grad, gamma.grad, TOL_WGRAD, "bn dgamma")
    assert_close(bg.grad, beta.grad, TOL_WGRAD, "bn dbeta")
    assert_close(rmg, rm, TOL_WGRAD, "running_mean")
    assert_close(rvg, rv, TOL_WGRAD, "running_var")


def test_batchnorm1d_and_eval(pg):
    F = pg.functional
    x = _leaf(64, 256, seed=1).requires_grad_(True)
    gamma, beta = (_leaf(256, seed=2) + 1.5).requires_grad_(True), _leaf(256, seed=3).requires_grad_(True)
    rm, rv = torch.zeros(256), torch.ones(256)
    y_ref = TF.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 0.8)
    y_ref.backward(_leaf(64, 256, seed=4))
    xg, gg, bg = (t.detach().to(DEV).requires_grad_(True) for t in (x, gamma, beta))
    rmg, rvg = torch.zeros(256, device=DEV), torch.ones(256, device=DEV)
    y = F.norm(xg, gg, bg, None, rmg, rvg, True, 0.1, 0.8)
    y.backward(_leaf(64, 256, seed=4).to(DEV))
    assert_close(y, y_ref, TOL_FWD, "bn1d fwd")
    assert_close(xg.grad, x.grad, 2e-5, "bn1d dx")
    assert_close(rvg, rv, TOL_WGRAD, "bn1d running_var")
    y_eval = F.norm(xg.detach(), gg.detach(), bg.detach(), None, rmg, rvg, False, 0.1, 0.8)
    assert_close(y_eval, TF.batch_norm(x.detach(), rm, rv, gamma.detach(), beta.detach(), False, 0.1, 0.8), TOL_FWD, "bn eval")


@pytest.mark.parametrize("cfg", [(2, 256, 16, 16, 2), (2, 64, 9, 7, 1), (1, 512, 2, 2, 0), (3, 128, 1, 2, 0),
                                 (2, 64, 160, 128, 2)])  # last: HBM-sized
def test_instancenorm(pg, cfg):
    N, C, H, W, act = cfg
    F = pg.functional
    x = (_leaf(N, C, H, W, seed=1) * 3 + 0.5).requires_grad_(True)
    z_ref = TF.instance_norm(x, eps=1e-5)
    y_ref = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu}[act](z_ref)
    gy = _leaf(N, C, H, W, seed=2)
    y_ref.backward(gy)
    xg = x.detach().to(DEV).requires_grad_(True)
    y = F.norm(xg, None, None, None, None, None, True, 0.1, 1e-5, True, act, 0.2)
    y.backward(gy.to(DEV))
    assert_close(y, y_ref, TOL_FWD, "in fwd")
    if H * W > 1:
        keep = (z_ref.detach().abs() > 1e-5).float()  # pre-activations at the kink: see test_batchnorm2d_train
        assert_close(xg.grad.cpu() * keep, x.grad * keep, 5e-5, "in dx")


@pytest.mark.parametrize("cfg", [(1, 512, 2, 2, 2), (1, 512, 8, 8, 1), (2, 256, 16, 16, 2)])
def test_instancenorm_with_fused_dropout(pg, cfg):
    """InstanceNorm2d -> LeakyReLU | ReLU -> Dropout(0.5) of pix2pix/models.py:25-28,41-45 on the small inner U-Net levels: the mask
    multiplies inside the one-launch normalisation (forward and backward) - against torch with the same mask."""
    N, C, H, W, act = cfg
    F = pg.functional
    x = (_leaf(N, C, H, W, seed=1) * 2 + 0.3).requires_grad_(True)
    mask = (torch.rand(N, C, H, W, generator=torch.Generator().manual_seed(5)) > 0.5).float() * 2.0
    z_ref = TF.instance_norm(x, eps=1e-5)
    y_ref = {1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu}[act](z_ref) * mask
    gy = _leaf(N, C, H, W, seed=2)
    y_ref.backward(gy)
    xg = x.detach().to(DEV).requires_grad_(True)
    assert F.norm_small_takes(xg, True)
    with Launches() as n:
        y = F.norm(xg, None, None, None, None, None, True, 0.1, 1e-5, True, act, 0.2, mask=mask.to(DEV))
        y.backward(gy.to(DEV))
        assert n("norm_small_fwd_kernel") == 1 and n("norm_small_bwd_kernel") == 1 and n("norm_") == 2   # nothing else of norm.hip
    assert_close(y, y_ref, TOL_FWD, "in+dropout fwd")
    keep = (z_ref.detach().abs() > 1e-5).float()
    assert_close(xg.grad.cpu() * keep, x.grad * keep, 5e-5, "in+dropout dx")


def test_norm_residual(pg):
    F = pg.functional
    x, r = _leaf(2, 64, 8, 8, seed=1).requires_grad_(True), _leaf(2, 64, 8, 8, seed=2).requires_grad_(True)
    y_ref = r + TF.instance_norm(x)
    gy = _leaf(2, 64, 8, 8, seed=3)
    y_ref.backward(gy)
    xg, rg = x.detach().to(DEV).requires_grad_(True), r.detach().to(DEV).requires_grad_(True)
    y = F.norm(xg, None, None, rg, None, None, True, 0.1, 1e-5, True)
    y.backward(gy.to(DEV))
    assert_close(y, y_ref, TOL_FWD, "in+res fwd")
    assert_close(rg.grad, r.grad, 1e-7, "res grad")
    assert_close(xg.grad, x.grad, 5e-5, "in+res dx")


@pytest.mark.parametrize("act", [1, 2, 3, 4])
def test_activations(pg, act):
    F = pg.functional
    x = (_leaf(3, 5, 7, 9, seed=1) * 3).requires_grad_(True)
    fn = {1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu, 3: torch.tanh, 4: torch.sigmoid}[act]
    y_ref = fn(x)
    gy = _leaf(3, 5, 7, 9, seed=2)
    y_ref.backward(gy)
    xg = x.detach().to(DEV).requires_grad_(True)
    y = F.activation(xg, act, 0.2)
    y.backward(gy.to(DEV))
    assert_close(y, y_ref, 1e-6, "act fwd")
    assert_close(xg.grad, x.grad, 2e-6, "act bwd")


def test_prelu(pg):
    F = pg.functional
    x = _leaf(2, 64, 12, 12, seed=1).requires_grad_(True)
    a = torch.tensor([0.25], requires_grad=True)
    y_ref = TF.prelu(x, a)
    gy = _leaf(2, 64, 12, 12, seed=2)
    y_ref.backward(gy)
    xg, ag = x.detach().to(DEV).requires_grad_(True), a.detach().to(DEV).requires_grad_(True)
    y = F.prelu(xg, ag)
    y.backward(gy.to(DEV))
    assert_close(y, y_ref, 1e-7, "prelu fwd")
    assert_close(xg.grad, x.grad, 1e-7, "prelu dx")
    assert_close(ag.grad, a.grad, TOL_WGRAD, "prelu dslope")


@pytest.mark.parametrize("cfg", [(4, 64, 12, 12, 0.8, False), (2, 256, 9, 7, 1e-5, True), (3, 10, 5, 4, 0.8, False)])
def test_batchnorm_prelu_fused(pg, cfg):
    """nn.BatchNorm2d -> nn.PReLU() in the norm launches (srgan/models.py:23-24,55-57): forward, dx, dgamma, dbeta and the
    slope gradient (a third sum of the norm's statistics pass) against torch; with and without the residual add; and the
    Sequential route BatchNorm2d -> PixelShuffle -> PReLU (PReLU applied before the shuffle) against the unfused modules."""
    N, C, H, W, eps, with_res = cfg
    F = pg.functional
    x = (_leaf(N, C, H, W, seed=1) * 2 + 0.3).requires_grad_(True)
    gamma = (_leaf(C, seed=2) * 0.5 + 1.0).requires_grad_(True)
    beta = (_leaf(C, seed=3) * 0.5).requires_grad_(True)
    a = torch.tensor([0.25], requires_grad=True)
    r = _leaf(N, C, H, W, seed=4).requires_grad_(True) if with_res else None
    y_ref = TF.prelu(TF.batch_norm(x, None, None, gamma, beta, True, 0.1, eps), a)
    if with_res:
        y_ref = y_ref + r
    gy = _leaf(N, C, H, W, seed=5)
    y_ref.backward(gy)
    xg, gg, bg, ag = (t.detach().to(DEV).requires_grad_(True) for t in (x, gamma, beta, a))
    rg = r.detach().to(DEV).requires_grad_(True) if with_res else None
    y = F.norm(xg, gg, bg, rg, None, None, True, 0.1, eps, False, 0, 0.0, None, ag)
    y.backward(gy.to(DEV))
    assert_close(y, y_ref, TOL_FWD, "bn+prelu fwd")
    pre = TF.batch_norm(x.detach(), None, None, gamma.detach(), beta.detach(), True, 0.1, eps)
    keep = (pre.abs() > 1e-5).float()   # the kink: a pre-activation within rounding of 0 may land on the other branch
    assert_close(xg.grad.cpu() * keep, x.grad * keep, 5e-5, "bn+prelu dx")
    assert_close(gg.grad, gamma.grad, 2e-5, "bn+prelu dgamma")
    assert_close(bg.grad, beta.grad, 2e-5, "bn+prelu dbeta")
    assert_close(ag.grad, a.grad, 2e-5, "bn+prelu dslope")
    if with_res:
        assert_close(rg.grad, r.grad, 1e-7, "bn+prelu residual grad")


@pytest.mark.parametrize("cfg", [(3, 16, 5, 7, True), (2, 256, 12, 12, True), (2, 8, 6, 4, False)])
def test_batchnorm_shuffle_prelu_vs_torch(pg, cfg):
    """BatchNorm2d -> PixelShuffle(2) -> PReLU (srgan/models.py:55-57) with the shuffle as the store / gradient-load index map of
    the norm launches, against torch's three ops; ragged H != W; and the shuffle alone (no PReLU)."""
    N, C, H, W, with_prelu = cfg
    F = pg.functional
    x = (_leaf(N, C, H, W, seed=1) * 2 + 0.3).requires_grad_(True)
    gamma = (_leaf(C, seed=2) * 0.5 + 1.0).requires_grad_(True)
    beta = (_leaf(C, seed=3) * 0.5).requires_grad_(True)
    a = torch.tensor([0.25], requires_grad=True)
    y_ref = TF.pixel_shuffle(TF.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5), 2)
    if with_prelu:
        y_ref = TF.prelu(y_ref, a)
    gy = _leaf(N, C // 4, 2 * H, 2 * W, seed=5)
    y_ref.backward(gy)
    xg, gg, bg, ag = (t.detach().to(DEV).requires_grad_(True) for t in (x, gamma, beta, a))
    y = F.norm(xg, gg, bg, None, None, None, True, 0.1, 1e-5, False, 0, 0.0, None, ag if with_prelu else None, 2)
    assert y.shape == y_ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.to(DEV))
    assert_close(y, y_ref, TOL_FWD, "bn+shuffle+prelu fwd")
    pre = TF.batch_norm(x.detach(), None, None, gamma.detach(), beta.detach(), True, 0.1, 1e-5)
    keep = (pre.abs() > 1e-5).float() if with_prelu else torch.ones_like(pre)
    assert_close(xg.grad.cpu() * keep, x.grad * keep, 5e-5, "bn+shuffle+prelu dx")
    assert_close(gg.grad, gamma.grad, 2e-5, "dgamma")
    assert_close(bg.grad, beta.grad, 2e-5, "dbeta")
    if with_prelu:
        assert_close(ag.grad, a.grad, 2e-5, "dslope")


def test_sequential_bn_shuffle_prelu_equals_unfused(pg):
    import copy

    nn = pg.nn
    torch.manual_seed(3)
    seq = nn.Sequential(nn.Conv2d(16, 64, 3, 1, 1), nn.BatchNorm2d(64), nn.PixelShuffle(upscale_factor=2), nn.PReLU()).to(DEV)
    ref = copy.deepcopy(seq)
    x = _leaf(2, 16, 10, 10, seed=7).to(DEV)
    gy = _leaf(2, 16, 20, 20, seed=8).to(DEV)
    outs = []
    for m, fused in ((seq, True), (ref, False)):
        pg.set_fusion(fused)
        try:
            xin = x.clone().requires_grad_(True)
            y = m(xin)
            y.backward(gy)
            outs.append((y.detach(), xin.grad, [p.grad.clone() for p in m.parameters()]))
        finally:
            pg.set_fusion(True)
    assert outs[0][0].shape == (2, 16, 20, 20) and outs[0][0].is_contiguous(memory_format=torch.channels_last)
    assert_close(outs[0][0], outs[1][0], 2e-6, "bn->shuffle->prelu fwd, fused vs modules")
    assert_close(outs[0][1], outs[1][1], 2e-5, "input gradient")
    for ga, gb, (k, _) in zip(outs[0][2], outs[1][2], seq.named_parameters()):
        if k == "0.bias":   # conv bias in front of BatchNorm: exactly-zero true gradient, rounding noise on both sides
            continue
        assert_close(ga, gb, 5e-5, "gradient of " + k)


@pytest.mark.parametrize("shape", [(1, 9, 64), (2, 13, 96)], ids=["1x9x64", "2x13x96"])
def test_trunk_block_batchnorm_prelu_folded_into_the_conv(pg, shape, monkeypatch):
    """srgan/models.py:19-31 ResidualBlock: conv, BatchNorm2d(64, 0.8), PReLU, conv, BatchNorm2d(64, 0.8), + x.  nn.Sequential hands
    BatchNorm -> PReLU -> Conv2d(64, 64, 3, 1, 1) to one Function: the second conv (forward and weight gradient) reads the first conv's
    output through the normalisation and the PReLU (csrc/conv_c64.hip INMAP = 2), no launch stores the activated tensor.  Against torch in
    fp64 on the host: output, input gradient, every parameter gradient, the running statistics and num_batches_tracked; then against
    the same block with the fold switched off (the separate norm launches)."""
    import copy

    N, H, W = shape
    nn, F = pg.nn, pg.functional
    torch.manual_seed(5)
    ref = torch.nn.Sequential(torch.nn.Conv2d(64, 64, 3, 1, 1), torch.nn.BatchNorm2d(64, 0.8), torch.nn.PReLU(), torch.nn.Conv2d(64, 64, 3, 1, 1),
                              torch.nn.BatchNorm2d(64, 0.8))
    with torch.no_grad():
        for bn in (ref[1], ref[4]):
            bn.weight.copy_(_leaf(64, seed=21) * 0.5 + 1.0)
            bn.bias.copy_(_leaf(64, seed=22) * 0.5)
    x = _leaf(N, 64, H, W, seed=7)
    gy = _leaf(N, 64, H, W, seed=8)
    ref64 = copy.deepcopy(ref).double()
    x64 = x.double().requires_grad_(True)
    y64 = x64 + ref64(x64)
    y64.backward(gy.double())
    outs = []
    for fold in (True, False):
        monkeypatch.setattr(F, "_BN_FOLD", fold)
        seq = pg.swap(copy.deepcopy(ref)).to(DEV)
        assert type(seq) is nn.Sequential
        xin = x.to(DEV).clone().requires_grad_(True)
        with Launches() as n:
            y = seq(xin, res=xin)
            y.backward(gy.to(DEV))
            if fold:
                assert n("c64_conv_kernel<2>") == 1 and n("c64_wgrad_kernel<2>") == 1, "the folded kernels did not run"
                assert n("norm_apply") == 1, "the activated tensor was stored"   # (the one launch: the block's last BatchNorm + x)
            else:
                assert n("c64_conv_kernel<2>") == 0 and n("norm_apply") == 2
        outs.append((y.detach(), xin.grad, {k: p.grad.clone() for k, p in seq.named_parameters()},
                     {k: b.clone() for k, b in seq.named_buffers()}))
    for tag, (y, dx, grads, bufs) in zip(("folded", "separate"), outs):
        assert_close(y, y64.float(), 2e-5, tag + " output")
        assert_close(dx, x64.grad.float(), 1e-4, tag + " input gradient")
        for k, p in ref64.named_parameters():
            if k in ("0.bias", "3.bias"):   # a conv bias in front of BatchNorm: zero true gradient, rounding noise on both sides
                assert float(grads[k].abs().max()) <= 1e-3 * float(gy.abs().sum()) / 64, k
                continue
            assert_close(grads[k], p.grad.float(), 1e-4, tag + " gradient of " + k)
        for k, b in ref64.named_buffers():
            if k.endswith("num_batches_tracked"):
                assert int(bufs[k]) == int(b) == 1, k
            else:
                assert_close(bufs[k], b.float(), 1e-5, tag + " " + k)
    assert_close(outs[0][0], outs[1][0], 2e-6, "folded vs separate launches: output")
    assert_close(outs[0][1], outs[1][1], 2e-5, "folded vs separate launches: input gradient")


BN_CONV1_CASES = [(2, 64, 16, 16, "lrelu", True), (3, 64, 13, 20, "lrelu", True), (2, 32, 9, 24, "relu", False), (1, 128, 8, 8, "none", True)]


@pytest.mark.parametrize("cfg", BN_CONV1_CASES, ids=["2x64x16x16", "3x64x13x20-ragged", "2x32x9x24-relu-nobias", "1x128x8x8-noact"])
def test_generator_tail_batchnorm_folded_into_the_image_conv(pg, cfg, monkeypatch):
    """dcgan.py:60-62: BatchNorm2d(64, 0.8), LeakyReLU(0.2), Conv2d(64, 1, 3, 1, 1), Tanh as ONE Function (nn.Sequential's peephole): the thin-N
    conv reads the BatchNorm input through the normalisation (thin_conv_kernel<1, true>), the backward walks x twice (bn_conv1_bwd_sums_kernel:
    the conv's weight gradient + the BatchNorm sums; bn_conv1_bwd_apply_kernel: dx) and recomputes the conv's input gradient from dz - neither
    the normalised tensor nor that gradient is stored.  Against torch in fp64 on the host (output, input gradient, every parameter gradient,
    running statistics) and against the separate launches."""
    import copy

    N, C, H, W, act, bias = cfg
    F = pg.functional
    torch.manual_seed(11)
    mods = [torch.nn.BatchNorm2d(C, 0.8)]
    if act != "none":
        mods.append(torch.nn.LeakyReLU(0.2) if act == "lrelu" else torch.nn.ReLU())
    mods += [torch.nn.Conv2d(C, 1, 3, 1, 1, bias=bias), torch.nn.Tanh()]
    ref = torch.nn.Sequential(*mods)
    with torch.no_grad():
        ref[0].weight.copy_(_leaf(C, seed=21) * 0.5 + 1.0)
        ref[0].bias.copy_(_leaf(C, seed=22) * 0.5)
    x = _leaf(N, C, H, W, seed=7) * 2 + 0.25
    gy = _leaf(N, 1, H, W, seed=8)
    ref64 = copy.deepcopy(ref).double()
    x64 = x.double().requires_grad_(True)
    y64 = ref64(x64)
    y64.backward(gy.double())
    outs = []
    for fold in (True, False):
        monkeypatch.setattr(F, "_BN_FOLD", fold)
        seq = pg.swap(copy.deepcopy(ref)).to(DEV)
        xin = x.to(DEV).clone().requires_grad_(True)
        with Launches() as n:
            y = seq(xin)
            y.backward(gy.to(DEV))
            if fold:
                assert n("thin_conv_kernel<1, true>") == 1 and n("bn_conv1_bwd_sums_kernel") == 1 and n("bn_conv1_bwd_apply_kernel") == 1
                assert n("norm_apply") == 0 and n("norm_bwd_apply") == 0, "the normalised tensor / the conv's input gradient was stored"
            else:
                assert n("bn_conv1") == 0 and n("norm_apply") == 1
        outs.append((y.detach(), xin.grad, {k: p.grad.clone() for k, p in seq.named_parameters()},
                     {k: b.clone() for k, b in seq.named_buffers()}))
    # a pre-activation within rounding of the kink may take the other branch: compare the input gradient away from it
    pre = TF.batch_norm(x, None, None, ref[0].weight.detach(), ref[0].bias.detach(), True, 0.1, ref[0].eps)
    keep = (pre.abs() > 1e-5).float() if act != "none" else torch.ones_like(pre)
    for tag, (y, dx, grads, bufs) in zip(("folded", "separate"), outs):
        assert_close(y, y64.float(), 2e-5, tag + " output")
        assert_close(dx.cpu() * keep, x64.grad.float() * keep, 1e-4, tag + " input gradient")
        for k, p in ref64.named_parameters():
            assert_close(grads[k], p.grad.float(), 1e-4, tag + " gradient of " + k)
        for k, b in ref64.named_buffers():
            if k.endswith("num_batches_tracked"):
                assert int(bufs[k]) == int(b) == 1, k
            else:
                assert_close(bufs[k], b.float(), 1e-5, tag + " " + k)
    assert_close(outs[0][0], outs[1][0], 2e-6, "folded vs separate launches: output")
    assert_close(outs[0][1], outs[1][1], 2e-5, "folded vs separate launches: input gradient")


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 128), (1, 12, 20, 32, 40)], ids=["64-128ch", "k-tail-40ch"])
def test_relu_backward_handed_to_the_consumer(pg, shape):
    """vgg19.features[:18] (srgan/models.py:8-15): conv ReLU conv ReLU MaxPool conv ReLU.  nn.Sequential hands the ReLU backward of a
    conv to its consumer - the epilogue of the next conv's input-gradient launch (migan_conv2d_dgrad_relu_ws) or the pool's backward
    (migan_maxpool2_relu_bwd).  Same input / weight gradients as torch on the host, bit-identical to the un-handed form, and no
    act_bwd_kernel launch for the handed layers."""
    import copy

    import torch.nn as tnn

    from pytorch_gan_amd import nn as gnn

    N, H, W, C1, C2 = shape
    torch.manual_seed(3)
    ref = tnn.Sequential(tnn.Conv2d(C1, C1, 3, 1, 1), tnn.ReLU(inplace=True), tnn.Conv2d(C1, C2, 3, 1, 1), tnn.ReLU(inplace=True),
                         tnn.MaxPool2d(2, 2), tnn.Conv2d(C2, C2, 3, 1, 1), tnn.ReLU(inplace=True))
    x = torch.randn(N, C1, H, W)
    dy = torch.randn(N, C2, H // 2, W // 2)
    xr = x.clone().requires_grad_(True)
    ref(xr).backward(dy)
    outs = {}
    for hand in (True, False):
        m = pg.swap(copy.deepcopy(ref)).to(DEV)
        xg = x.clone().to(DEV).requires_grad_(True)
        old = gnn._RELU_HANDOFF
        gnn._RELU_HANDOFF = hand
        try:
            with Launches() as n:
                y = m(xg)
                y.backward(dy.to(DEV))
                counts = (n("act_bwd"), n("maxpool2_bwd_kernel"))   # act_bwd_kernel | act_bwd_colsum_kernel (trainable convs with a bias)
        finally:
            gnn._RELU_HANDOFF = old
        outs[hand] = (y.detach().cpu().clone(), xg.grad.cpu().clone(), [p.grad.cpu().clone() for p in m.parameters()])
        # three ReLUs: handed -> only the last one (consumed by the loss) runs its own backward pass
        assert counts == ((1, 1) if hand else (3, 1)), counts
    # against torch on the host: a pre-activation within rounding of 0 may take the other side of the ReLU (one element in ~1e5)
    assert_close(outs[True][1], xr.grad, 5e-3, "dx vs torch")
    for g, p in zip(outs[True][2], ref.parameters()):
        assert_close(g, p.grad, 5e-3, "parameter gradient vs torch")
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    for a, b in zip(outs[True][2], outs[False][2]):   # bias gradients: column sums from another kernel (no act_bwd_colsum pass to ride in)
        assert torch.equal(a, b) if a.dim() == 4 else rel_fro(a, b) <= TOL_BIAS



@pytest.mark.parametrize("cfg", [((3, 3, 3, 3), 1), ((1, 1, 1, 1), 1), ((1, 1, 0, 0), 0), ((0, 0, 0, 0), 2), ((2, 2, 1, 1), 2)])
def test_gather2d(pg, cfg):
    pads, mode = cfg
    F = pg.functional
    x = _leaf(2, 6, 9, 8, seed=1).requires_grad_(True)
    y_ref = _ref_gather(x, pads, mode)
    gy = _leaf(*y_ref.shape, seed=2)
    y_ref.backward(gy)
    xg = x.detach().to(DEV).requires_grad_(True)
    y = F.gather2d(xg, pads, mode)
    y.backward(gy.to(DEV))
    assert torch.equal(y.cpu(), y_ref.detach()), "pad/upsample forward must be bit-exact (index remap)"
    assert_close(xg.grad, x.grad, 1e-6, "gather bwd")


def test_pixel_shuffle_maxpool_cat_add(pg):
    F = pg.functional
    x = _leaf(2, 16, 5, 6, seed=1).requires_grad_(True)
    y_ref = TF.pixel_shuffle(x, 2)
    gy = _leaf(*y_ref.shape, seed=2)
    y_ref.backward(gy)
    xg = x.detach().to(DEV).requires_grad_(True)
    y = F.pixel_shuffle(xg, 2)
    y.backward(gy.to(DEV))
    assert torch.equal(y.cpu(), y_ref.detach()) and torch.equal(xg.grad.cpu(), x.grad), "PixelShuffle is an index map"

    x = _leaf(2, 8, 6, 8, seed=3).requires_grad_(True)
    y_ref = TF.max_pool2d(x, 2, 2)
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    xg = x.detach().to(DEV).requires_grad_(True)
    y = F.maxpool2(xg)
    y.backward(gy.to(DEV))
    assert torch.equal(y.cpu(), y_ref.detach()) and torch.equal(xg.grad.cpu(), x.grad), "MaxPool is selection"

    a, b = _leaf(2, 5, 4, 4, seed=5).requires_grad_(True), _leaf(2, 3, 4, 4, seed=6).requires_grad_(True)
    y_ref = torch.cat((a, b), 1)
    gy = _leaf(*y_ref.shape, seed=7)
    y_ref.backward(gy)
    ag, bg = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    y = F.cat_channels(ag, bg)
    y.backward(gy.to(DEV))
    assert torch.equal(y.cpu(), y_ref.detach()) and torch.equal(ag.grad.cpu(), a.grad) and torch.equal(bg.grad.cpu(), b.grad)

    s = F.add(ag.detach(), ag.detach())
    assert torch.equal(s.cpu(), (a + a).detach())


def test_dropout_masks(pg):
    F = pg.functional
    x = _leaf(4, 16, 8, 8, seed=1).requires_grad_(True)
    m = (torch.rand(4, 16) > 0.25).float() / 0.75
    y_ref = x * m[:, :, None, None]
    gy = _leaf(4, 16, 8, 8, seed=2)
    y_ref.backward(gy)
    xg = x.detach().to(DEV).requires_grad_(True)
    y = F.mul_mask(xg, m.to(DEV))
    y.backward(gy.to(DEV))
    assert torch.equal(y.cpu(), y_ref.detach()) and torch.equal(xg.grad.cpu(), x.grad)
    # device Philox stream: keep-probability and scaling
    ctr = torch.zeros(2, dtype=torch.int64, device=DEV)  # {stream position, arrival ticket}
    mk = F.rand_mask((1 << 20,), 0.25, 1234, ctr, DEV)
    vals = torch.unique(mk).cpu().tolist()
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1 / 0.75) < 1e-6
    assert abs((mk > 0).float().mean().item() - 0.75) < 3e-3
    mk2 = F.rand_mask((1 << 20,), 0.25, 1234, ctr, DEV)
    assert not torch.equal(mk, mk2), "counter must advance the stream"
    assert ctr.cpu().tolist() == [2 * (1 << 18), 0], "position advanced by the last block, ticket back at rest"


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_losses(pg, kind):
    F = pg.functional
    n = (64, 1) if kind == 0 else (2, 3, 33, 31)
    x = _leaf(*n, seed=1)
    if kind == 0:
        x = torch.sigmoid(x * 4)
        x[0, 0], x[1, 0] = 0.0, 1.0  # exercises the log clamp at -100
    x = x.requires_grad_(True)
    t = (torch.rand(*n) > 0.5).float() if kind == 0 else _leaf(*n, seed=2)
    ref = [TF.binary_cross_entropy, TF.mse_loss, TF.l1_loss, lambda a, b: a.mean()][kind](x, t)
    (ref * 1.7).backward()
    xg = x.detach().to(DEV).requires_grad_(True)
    out = F.loss(kind, xg, None if kind == 3 else t.to(DEV))
    (F.axpby(out, None, 1.7, 0.0)).backward()
    assert abs(out.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    assert_close(xg.grad, x.grad, 2e-6, "loss grad")


def test_adam_matches_torch(pg):
    ps = [_leaf(300, 70, seed=1), _leaf(5000, seed=2), _leaf(3, 3, 3, 3, seed=3)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    o_ref = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
    o_mine = pg.optim.Adam(mine, lr=2e-4, betas=(0.5, 0.999))
    for step in range(5):
        o_ref.zero_grad()
        o_mine.zero_grad()
        for i, (r, m) in enumerate(zip(ref, mine)):
            g = _leaf(*r.shape, seed=10 * step + i)
            r.grad = g.clone()
            m.grad.add_(g.to(DEV))
        o_ref.step()
        o_mine.step()
    for r, m in zip(ref, mine):
        assert_close(m, r, 1e-7, "adam param")


def test_gantensor_view_add_cat(pg):
    nn = pg.nn
    conv = nn.Conv2d(8, 16, 3, 1, 1).to(DEV)
    x = _leaf(2, 8, 4, 4, seed=1).to(DEV)
    y = conv(x)
    assert isinstance(y, nn.GanTensor)
    flat = y.view(y.shape[0], -1)  # dcgan.py:96 on NHWC storage
    ref = TF.conv2d(x.cpu(), conv.weight.detach().cpu(), conv.bias.detach().cpu(), 1, 1)
    assert_close(flat, ref.view(2, -1), TOL_FWD, "view of NHWC activation")
    s = y + y
    assert isinstance(s, nn.GanTensor)
    assert_close(s, 2 * ref, TOL_FWD, "residual add")
    c = torch.cat((y, y), 1)
    assert_close(c, torch.cat((ref, ref), 1), TOL_FWD, "cat")


def test_cpu_tensor_raises(pg):
    with pytest.raises(RuntimeError):
        pg.functional.activation(torch.zeros(4), 1, 0.2)


@pytest.mark.parametrize("cfg", [(2, 128, 16, 16, 128, 0, True), (2, 128, 8, 12, 64, 1, True), (1, 256, 6, 5, 128, 2, False),
                                 (2, 32, 7, 9, 20, 0, True), (1, 64, 4, 4, 3, 3, True),
                                 (8, 128, 48, 48, 64, 0, True),   # 128x64 tiles, tap-inner K order; weight gradient on 64x256 tiles
                                 (1, 64, 8, 16, 64, 1, True),     # ... 64x256 weight-gradient tiles, one N-tile per class
                                 # the two up-conv layers of the bench line at its exact geometry (dcgan.py:55,59 at --img_size 64, batch 128):
                                 # upconv_wgrad[128x 128->64 @64] is the dominant launch - its split plan at 524 288 pixels runs here
                                 (128, 128, 32, 32, 64, 0, True), (128, 128, 16, 16, 128, 0, True)])
def test_upconv3x3_phase_collapsed(pg, cfg):
    """Upsample(2) -> Conv3x3(p=1) in the phase-collapsed form (dcgan.py:54-55,58-59; cyclegan/models.py:74-75)
    against the dense reference: forward, dgrad, wgrad (collapsed when Co%4==0 and Ci%4==0, dense fallback else)."""
    N, Ci, H, W, Co, act, bias = cfg
    F = pg.functional
    x = _leaf(N, Ci, H, W, seed=1).requires_grad_(True)
    w = _leaf(Co, Ci, 3, 3, seed=2, scale=0.2).requires_grad_(True)
    b = _leaf(Co, seed=3).requires_grad_(True) if bias else None
    y_ref = TF.conv2d(TF.interpolate(x, scale_factor=2, mode="nearest"), w, b, 1, 1)
    y_ref = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu, 3: torch.tanh}[act](y_ref)
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    xg, wg = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
    bg = b.detach().to(DEV).requires_grad_(True) if bias else None
    y = F.upconv3x3(xg, wg, bg, act, 0.2)
    with Launches() as nl:
        y.backward(gy.to(DEV))
        if Co == 64 and Ci % 64 == 0 and W % 8 == 0:   # dcgan.py:59 / cyclegan/models.py:75 shape class: the wide weight-gradient tile
            assert nl("wgrad_dma_kernel<64, 256") == 1
    assert_close(y, y_ref, TOL_FWD, "upconv fwd")
    assert_close(xg.grad, x.grad, TOL_FWD, "upconv dgrad")
    assert_close(wg.grad, w.grad, TOL_WGRAD, "upconv wgrad")
    if bias:
        assert_close(bg.grad, b.grad, TOL_WGRAD, "upconv bias")
    # the dense gathered path must agree too (set_upconv_collapse(False))
    F.set_upconv_collapse(False)
    try:
        y2 = F.upconv3x3(xg.detach(), wg.detach(), bg.detach() if bias else None, act, 0.2)
    finally:
        F.set_upconv_collapse(True)
    assert_close(y2, y_ref, TOL_FWD, "dense up2 conv")


def test_direct_grad_accumulation(pg):
    """With a pre-allocated contiguous `.grad` (the optimiser's flat bucket) the wgrad / bias / norm reductions ADD
    into it in place (functional.set_direct_grad); the values must equal autograd's AccumulateGrad result, twice in a
    row (two backward passes accumulate, as the D step of dcgan.py:170-174 does)."""
    F = pg.functional
    nn = pg.nn
    torch.manual_seed(0)
    net = nn.Sequential(
        nn.Conv2d(8, 16, 3, 1, 1), nn.BatchNorm2d(16), nn.LeakyReLU(0.2),
        nn.Upsample(scale_factor=2), nn.Conv2d(16, 12, 3, stride=1, padding=1), nn.ReLU(),
        nn.ConvTranspose2d(12, 8, 4, 2, 1), nn.Conv2d(8, 3, 3, 1, 1),
    ).to(DEV)
    lin = nn.Linear(3 * 16 * 16, 5).to(DEV)
    params = list(net.parameters()) + list(lin.parameters())
    x = _leaf(3, 8, 4, 4, seed=1).to(DEV)

    def run():
        out = lin(net(x).reshape(3, -1))
        (out * out).mean().backward()

    results = {}
    for direct in (False, True):
        F.set_direct_grad(direct)
        try:
            for i, p in enumerate(params):
                p.grad = torch.full_like(p, 0.25 * (i % 3))  # pre-existing gradient content to accumulate onto
            ptrs = [p.grad.data_ptr() for p in params]
            run()
            run()
            if direct:
                assert [p.grad.data_ptr() for p in params] == ptrs, "direct accumulation must not replace .grad"
            results[direct] = [p.grad.detach().clone() for p in params]
        finally:
            F.set_direct_grad(True)
    for p, a, b in zip(params, results[False], results[True]):
        assert_close(b, a, TOL_WGRAD, "direct grad %s" % (tuple(p.shape),))


@pytest.mark.parametrize("cfg", [(4, 1, 16, 16, 16, 2), (3, 16, 10, 10, 32, 2), (2, 32, 9, 9, 64, 1), (2, 64, 8, 8, 4, 1)])
def test_conv_lrelu_dropout2d_fused(pg, cfg):
    """Conv2d -> LeakyReLU -> Dropout2d (dcgan.py:77-78) with the mask multiply in the conv epilogue and the
    act'+mask product in one backward kernel, against torch with the same injected mask."""
    N, Ci, H, W, Co, stride = cfg
    nn = pg.nn
    x = _leaf(N, Ci, H, W, seed=1).requires_grad_(True)
    w = _leaf(Co, Ci, 3, 3, seed=2, scale=0.3).requires_grad_(True)
    b = _leaf(Co, seed=3).requires_grad_(True)
    mask = (torch.rand(N, Co, generator=torch.Generator().manual_seed(5)) > 0.25).float() / 0.75
    y_ref = TF.leaky_relu(TF.conv2d(x, w, b, stride, 1), 0.2) * mask[:, :, None, None]
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    net = nn.Sequential(nn.Conv2d(Ci, Co, 3, stride, 1), nn.LeakyReLU(0.2, inplace=True), nn.Dropout2d(0.25)).to(DEV)
    with torch.no_grad():
        net[0].weight.copy_(w)
        net[0].bias.copy_(b)
    xg = x.detach().to(DEV).requires_grad_(True)
    with pg.dropout_masks([mask]):
        y = net(xg)
    y.backward(gy.to(DEV))
    assert_close(y, y_ref, TOL_FWD, "fused dropout fwd")
    assert_close(xg.grad, x.grad, TOL_FWD, "fused dropout dgrad")
    assert_close(net[0].weight.grad, w.grad, TOL_WGRAD, "fused dropout wgrad")
    assert_close(net[0].bias.grad, b.grad, TOL_WGRAD, "fused dropout bias")


def test_bias_grad_fused_into_wgrad(pg):
    """Opt-in path (functional._FUSE_BIAS): the bias gradient comes out of the wgrad launches (db argument of
    migan_conv2d_wgrad / migan_upconv3x3_wgrad) instead of migan_colsum; same values, with and without .grad slots."""
    F, nn = pg.functional, pg.nn
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(8, 16, 3, 2, 1), nn.ReLU(), nn.Upsample(scale_factor=2),
                        nn.Conv2d(16, 12, 3, stride=1, padding=1)).to(DEV)
    lin = nn.Linear(12 * 8 * 8, 8).to(DEV)
    params = list(net.parameters()) + list(lin.parameters())
    x = _leaf(5, 8, 8, 8, seed=1).to(DEV)
    res = {}
    for fused in (False, True):
        for slots in (False, True):
            F._FUSE_BIAS = fused
            try:
                for p in params:
                    p.grad = torch.full_like(p, 0.5) if slots else None
                out = lin(net(x).reshape(5, -1))
                (out * out).mean().backward()
                res[(fused, slots)] = [p.grad.detach().clone() for p in params]
            finally:
                F._FUSE_BIAS = False
    for slots in (False, True):
        for p, a, b in zip(params, res[(False, slots)], res[(True, slots)]):
            assert_close(b, a, TOL_WGRAD, "fused bias %s slots=%s" % (tuple(p.shape), slots))


# ------------------------------------------------------------------------------------------------ round-2 kernels
@pytest.mark.parametrize("case", [(2, 8, 5, 4, 16), (1, 256, 12, 12, 256), (2, 64, 20, 18, 32), (3, 32, 4, 7, 40), (1, 256, 64, 64, 256)])
def test_reflect_pad1_dgrad_matches_padded_path(pg, case, monkeypatch):
    """ReflectionPad2d(1)+Conv3x3 input gradient: the direct form (pad-1 dgrad + added ring terms, no padded
    intermediate; cyclegan/models.py:26-35) against torch CPU and against the padded-extent + fold path it replaces.
    The 256-channel cases go through migan_conv2d_dgrad_reflect1_ws: 1 x 256 x 64 x 64 is the residual trunk at one image per GPU, whose
    ring launch - 64 tiles, corner classes 56 K-tiles deep - is cut along K."""
    from util import Launches

    N, Ci, H, W, Co = case
    if DEV == "cpu" and H * W > 1024:
        pytest.skip("the trunk-sized case is for the hardware (minutes on the execution model)")
    F = pg.functional
    x = _leaf(N, Ci, H, W, seed=1).requires_grad_(True)
    w = _leaf(Co, Ci, 3, 3, seed=2, scale=0.2).requires_grad_(True)
    y_ref = TF.conv2d(TF.pad(x, (1, 1, 1, 1), mode="reflect"), w, None, 1)
    gy = _leaf(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    outs = []
    for direct in (True, False):
        monkeypatch.setattr(F, "_REFLECT1", direct)
        xg = x.detach().to(DEV).requires_grad_(True)
        wg = w.detach().to(DEV).requires_grad_(True)
        y = F.conv2d(xg, wg, None, 1, (1, 1, 1, 1), F.GATHER_REFLECT)
        with Launches() as n:
            y.backward(gy.to(DEV))
            if direct and Ci == 256:
                # 12 x 12: both launches under-filled -> both through the ticketed 64 x 64 split-K instantiation; 64 x 64: the ring launch
                assert n("2, 4, true>") >= (2 if H * W <= 1024 else 1), "reflect dgrad of %s was not cut along K" % (case,)
        assert_close(xg.grad, x.grad, TOL_FWD, "reflect dgrad direct=%s" % direct)
        outs.append(xg.grad.clone())
    assert_close(outs[0], outs[1], 2e-6, "direct vs padded+fold")


@pytest.mark.parametrize("cfg", [(1, 128 * 16 * 16, 128, 1, True), (8, 64 * 64, 256, 2, False), (4, 33 * 7, 12, 0, False),
                                 (2, 50, 3, 0, False), (3, 129, 1, 0, False),
                                 (4, 16 * 16, 256, 1, False), (1, 4, 512, 2, False)])   # the one-launch small-tensor route
def test_norm_bwd_column_sum_slabs(pg, cfg):
    """migan_norm_bwd's per-block column sums of dx (the bias gradient of the conv in front of the norm layer is reduced
    from them inside that conv's wgrad launch) equal the column sums of the dx it wrote, for BatchNorm (G=1) and
    InstanceNorm (G>1) views, vector and scalar channel counts."""
    from pytorch_gan_amd._lib import check, lib

    G, P, C, act, affine = cfg
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    x = torch.randn(G, P, C, generator=g).to(DEV)
    dy = torch.randn(G, P, C, generator=g).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV) if affine else None
    beta = torch.randn(C, generator=g).to(DEV) if affine else None
    mean, invstd = torch.empty(G * C, device=DEV), torch.empty(G * C, device=DEV)
    nb = lib.migan_norm_workspace(G, P, C)
    ws = torch.empty(nb // 4 + 1, device=DEV)
    check(lib.migan_norm_stats(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), None, None, None, 0.1, 1e-5, G, P, C,
                               ws.data_ptr(), nb, st))
    dx = torch.empty_like(x)
    nslab = lib.migan_norm_colsum_slabs(G, P, C)
    slabs = torch.full((nslab, C), float("nan"), device=DEV)
    dg, db = (torch.empty(C, device=DEV), torch.empty(C, device=DEV)) if G == 1 else (None, None)
    P_ = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    check(lib.migan_norm_bwd(x.data_ptr(), dy.data_ptr(), mean.data_ptr(), invstd.data_ptr(), P_(gamma), P_(beta),
                             dx.data_ptr(), P_(dg), P_(db), G, P, C, act, 0.2, ws.data_ptr(), nb, 0, slabs.data_ptr(), st))
    got = slabs.double().sum(0).cpu()
    want = dx.double().sum((0, 1)).cpu()
    scale = dx.double().abs().sum((0, 1)).cpu()
    assert torch.isfinite(slabs).all()
    assert ((got - want).abs() <= 2e-6 * scale + 1e-9).all(), (got - want).abs().max()
    # and dx itself is unchanged by asking for the slabs
    dx2 = torch.empty_like(x)
    check(lib.migan_norm_bwd(x.data_ptr(), dy.data_ptr(), mean.data_ptr(), invstd.data_ptr(), P_(gamma), P_(beta),
                             dx2.data_ptr(), P_(dg), P_(db), G, P, C, act, 0.2, ws.data_ptr(), nb, 0, None, st))
    assert torch.equal(dx, dx2)


@pytest.mark.parametrize("cfg", [(4, 16 * 16, 32, 1, True), (128, 64 * 64, 1, 3, False), (2, 9, 12, 0, True), (3, 100, 3, 4, False)])
def test_act_bwd_colsum(pg, cfg):
    """Backward of conv -> act [-> Dropout2d] in one pass: dx bit-identical to the separate kernels, slabs = column sums."""
    from pytorch_gan_amd._lib import check, lib

    N, HW, C, act, masked = cfg
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5)
    y = torch.tanh(torch.randn(N, HW, C, generator=g)).to(DEV) if act in (3, 4) else torch.randn(N, HW, C, generator=g).to(DEV)
    if act == 4:
        y = y * 0.5 + 0.5
    dy = torch.randn(N, HW, C, generator=g).to(DEV)
    mask = ((torch.rand(N, C, generator=g) > 0.25).float() / 0.75).to(DEV) if masked else None
    dx = torch.empty_like(dy)
    nslab = lib.migan_norm_colsum_slabs(N, HW, C)
    slabs = torch.full((nslab, C), float("nan"), device=DEV)
    check(lib.migan_act_bwd_colsum(dy.data_ptr(), y.data_ptr(), None if mask is None else mask.data_ptr(), dx.data_ptr(),
                                   slabs.data_ptr(), N, HW, C, act, 0.2, st))
    ref = torch.empty_like(dy)
    if masked and C % 4 == 0:
        check(lib.migan_act_bwd_nc(dy.data_ptr(), y.data_ptr(), mask.data_ptr(), ref.data_ptr(), N, HW, C, act, 0.2, st))
    else:
        check(lib.migan_act_bwd(dy.data_ptr(), y.data_ptr(), ref.data_ptr(), dy.numel(), act, 0.2, st))
        if masked:
            ref = ref * mask[:, None, :]
    assert torch.equal(dx, ref)
    got, want = slabs.double().sum(0).cpu(), dx.double().sum((0, 1)).cpu()
    assert ((got - want).abs() <= 2e-6 * dx.double().abs().sum((0, 1)).cpu() + 1e-9).all()


def test_conv_bias_grad_via_norm_slabs_equals_colsum(pg, monkeypatch):
    """Conv2d(bias) -> InstanceNorm/BatchNorm: the conv's bias gradient reduced from the norm backward's slabs inside the
    wgrad launch equals the one from the separate column-sum launches (both are rounding noise around an exactly-zero
    true gradient, so they are compared with each other at the scale of sum|dy|)."""
    F = pg.functional
    x = _leaf(4, 32, 16, 16, seed=1).to(DEV)
    w = _leaf(64, 32, 3, 3, seed=2, scale=0.2).to(DEV)
    gy = _leaf(4, 64, 16, 16, seed=4).to(DEV)
    res = {}
    for fuse in (True, False):
        monkeypatch.setattr(F, "_COLSUM_FUSE", fuse)
        for inst in (False, True):
            b = _leaf(64, seed=3).to(DEV).requires_grad_(True)
            wg = w.clone().requires_grad_(True)
            y = F.norm(F.conv2d(x, wg, b, 1, (1, 1, 1, 1)), instance=inst, act=F.ACT_RELU)
            y.backward(gy)
            res[(fuse, inst)] = (b.grad.clone(), wg.grad.clone())
    for inst in (False, True):
        (b1, w1), (b0, w0) = res[(True, inst)], res[(False, inst)]
        assert torch.equal(w1, w0)
        assert (b1 - b0).abs().max().item() <= 1e-5 * gy.abs().sum().item() / 64


# ------------------------------------------------------------------------------------------------ F2: DCGAN-block clones
def test_embedding_mul_softmax_cross_entropy(pg):
    """acgan.py:50,61,100,113: label embedding * noise -> ... -> Linear+Softmax scores -> CrossEntropyLoss, against stock
    torch on the CPU, forward and all gradients (embedding gradient = deterministic scatter-add; label tensors bit-exact)."""
    import pytorch_gan_amd.nn as gnn

    g = torch.Generator().manual_seed(0)
    B, V, D, C = 64, 10, 100, 10
    emb_c = torch.nn.Embedding(V, D)
    lin_c = torch.nn.Linear(D, C)
    labels = torch.randint(0, V, (B,), generator=g)
    targets = torch.randint(0, C, (B,), generator=g)
    noise = torch.randn(B, D, generator=g)
    x_c = torch.mul(emb_c(labels), noise)
    p_c = torch.nn.Softmax(dim=1)(lin_c(x_c))
    loss_c = torch.nn.CrossEntropyLoss()(p_c, targets)   # the reference feeds softmax outputs to CrossEntropyLoss
    loss_c.backward()

    emb_g, lin_g = gnn.Embedding(V, D).to(DEV), gnn.Linear(D, C).to(DEV)
    emb_g.load_state_dict(emb_c.state_dict())
    lin_g.load_state_dict(lin_c.state_dict())
    x_g = torch.mul(emb_g(labels.to(DEV)), noise.to(DEV))
    p_g = gnn.Softmax()(lin_g(x_g))
    loss_g = gnn.CrossEntropyLoss()(p_g, targets.to(DEV))
    loss_g.backward()
    assert_close(x_g, x_c, 1e-7, "embedding*noise")
    assert_close(p_g, p_c, TOL_FWD, "softmax")
    assert abs(float(loss_g) - float(loss_c)) <= 1e-6 * max(1.0, abs(float(loss_c)))
    assert_close(emb_g.weight.grad, emb_c.weight.grad, TOL_WGRAD, "embedding grad")
    assert_close(lin_g.weight.grad, lin_c.weight.grad, TOL_WGRAD, "linear grad through softmax+CE")
    assert_close(lin_g.bias.grad, lin_c.bias.grad, TOL_WGRAD, "bias grad through softmax+CE")
    # rows of the embedding gradient for labels that do not occur are exactly zero
    absent = [v for v in range(V) if v not in set(labels.tolist())]
    assert all(float(emb_g.weight.grad[v].abs().max()) == 0.0 for v in absent)


@pytest.mark.parametrize("n", [64, 40000])
def test_bce_with_logits(pg, n):
    """relativistic_gan.py:95 BCEWithLogitsLoss (mean), incl. large |logits| (no overflow: log_sigmoid form)."""
    import pytorch_gan_amd.nn as gnn

    g = torch.Generator().manual_seed(1)
    x = (torch.randn(n, 1, generator=g) * 6).requires_grad_(True)
    with torch.no_grad():
        x[:4, 0] = torch.tensor([80.0, -80.0, 0.0, 1e-8])
    t = (torch.rand(n, 1, generator=g) > 0.5).float()
    ref = torch.nn.BCEWithLogitsLoss()(x, t)
    ref.backward()
    xg = x.detach().to(DEV).requires_grad_(True)
    out = gnn.BCEWithLogitsLoss()(xg, t.to(DEV))
    out.backward()
    assert abs(float(out) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert_close(xg.grad, x.grad, 2e-6, "bce_with_logits grad")


def test_swap_dcgan_block_clone_layers(pg):
    """swap() re-classes the four extra leaf types of the DCGAN-block clones (acgan / relativistic_gan)."""
    import pytorch_gan_amd.nn as gnn

    m = torch.nn.ModuleDict({"emb": torch.nn.Embedding(10, 8), "aux": torch.nn.Sequential(torch.nn.Linear(8, 10), torch.nn.Softmax()),
                             "ce": torch.nn.CrossEntropyLoss(), "bcel": torch.nn.BCEWithLogitsLoss()})
    pg.swap(m)
    assert type(m["emb"]) is gnn.Embedding and type(m["aux"][1]) is gnn.Softmax
    assert type(m["ce"]) is gnn.CrossEntropyLoss and type(m["bcel"]) is gnn.BCEWithLogitsLoss
    with pytest.raises(ValueError):
        pg.swap(torch.nn.CrossEntropyLoss(label_smoothing=0.1))


# ------------------------------------------------------------------------------------------------ F1: double backward
def _gp_like(y, x, params, seed):
    """A gradient-penalty-shaped scalar: g = d(sum(y*v))/dx with create_graph, loss = sum(g^2 * c) -> backward."""
    gen = torch.Generator().manual_seed(seed)
    v = torch.randn(y.shape, generator=gen).to(y.device)
    (g,) = torch.autograd.grad((y * v).sum(), x, create_graph=True)
    c = torch.rand(g.shape, generator=gen).to(y.device) + 0.5
    loss = (g * g * c).sum()
    for p in params:
        p.grad = None
    loss.backward()
    return g.detach(), loss.detach()


@pytest.mark.parametrize("case", [(4, 8, 16, 16, 16, 3, 2, 1, 1, False), (2, 16, 12, 12, 32, 4, 2, 1, 1, True),
                                  (3, 3, 16, 16, 16, 4, 2, 1, 1, False), (2, 32, 9, 9, 64, 3, 1, 1, 0, False)])
def test_conv_double_backward(pg, case):
    """dgrad-of-dgrad (= conv forward) and wgrad-of-dgrad through Conv2d [+ LeakyReLU + Dropout2d mask]: the second-order
    terms of the conv-critic gradient penalties (dragan.py:77-78, stargan/models.py:92-101), against torch CPU fp64."""
    N, Ci, H, W, Co, k, stride, pad, act, masked = case
    F = pg.functional
    x = _leaf(N, Ci, H, W, seed=1).double().requires_grad_(True)
    w = _leaf(Co, Ci, k, k, seed=2, scale=0.3).double().requires_grad_(True)
    b = _leaf(Co, seed=3).double().requires_grad_(True)
    mask = ((torch.rand(N, Co, generator=torch.Generator().manual_seed(9)) > 0.25).double() / 0.75) if masked else None
    y = TF.conv2d(x, w, b, stride, pad)
    if act:
        y = TF.leaky_relu(y, 0.2)
    if masked:
        y = y * mask[:, :, None, None]
    g_ref, loss_ref = _gp_like(y, x, [w, b], 7)
    xg = x.detach().float().to(DEV).requires_grad_(True)
    wg = w.detach().float().to(DEV).requires_grad_(True)
    bg = b.detach().float().to(DEV).requires_grad_(True)
    yg = F.conv2d(xg, wg, bg, stride, (pad,) * 4, F.GATHER_ZERO, act, 0.2, None if mask is None else mask.float().to(DEV))
    g_gpu, loss_gpu = _gp_like(yg, xg, [wg, bg], 7)
    assert_close(g_gpu, g_ref.float(), TOL_FWD, "first-order input gradient (create_graph)")
    assert abs(float(loss_gpu) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref)), (float(loss_gpu), float(loss_ref))
    assert_close(wg.grad, w.grad.float(), TOL_WGRAD, "d(penalty)/dw through dgrad")
    assert bg.grad is None or float(bg.grad.abs().max()) == 0.0   # the penalty does not depend on the bias
    with F.input_grad_only():   # the hint skips the (discarded) first-order weight-gradient kernels, same result
        xg2 = x.detach().float().to(DEV).requires_grad_(True)
        wg2 = w.detach().float().to(DEV).requires_grad_(True)
        yg2 = F.conv2d(xg2, wg2, b.detach().float().to(DEV), stride, (pad,) * 4, F.GATHER_ZERO, act, 0.2,
                       None if mask is None else mask.float().to(DEV))
        _gp_like(yg2, xg2, [wg2], 7)
    assert_close(wg2.grad, w.grad.float(), TOL_WGRAD, "d(penalty)/dw with input_grad_only")


@pytest.mark.parametrize("cfg", [(False, 8, 32, 7, 7, 0.8, 0), (False, 4, 16, 8, 8, 1e-5, 1), (True, 3, 12, 6, 5, 1e-5, 0),
                                 (True, 2, 64, 4, 4, 1e-5, 2)])
def test_norm_double_backward(pg, cfg):
    """BatchNorm2d(C, 0.8) (dragan.py:80) / InstanceNorm2d (dualgan) under create_graph=True: migan_norm_bwd2 against
    torch CPU fp64 autograd through batch_norm / instance_norm (+ fused LeakyReLU / ReLU)."""
    inst, N, C, H, W, eps, act = cfg
    F = pg.functional
    x = _leaf(N, C, H, W, seed=1).double().requires_grad_(True)
    gamma = None if inst else (_leaf(C, seed=2) * 0.3 + 1.0).double().requires_grad_(True)
    beta = None if inst else _leaf(C, seed=3).double().requires_grad_(True)
    if inst:
        y = TF.instance_norm(x, eps=eps)
    else:
        y = TF.batch_norm(x, None, None, gamma, beta, True, 0.1, eps)
    y = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, 0.2), 2: torch.relu}[act](y)
    params = [] if inst else [gamma, beta]
    g_ref, loss_ref = _gp_like(y, x, params, 11)
    (gx_ref,) = [x.grad.clone()] if x.grad is not None else [None]
    xg = x.detach().float().to(DEV).requires_grad_(True)
    gg = None if inst else gamma.detach().float().to(DEV).requires_grad_(True)
    bgm = None if inst else beta.detach().float().to(DEV).requires_grad_(True)
    yg = F.norm(xg, gg, bgm, None, None, None, True, 0.1, eps, inst, act, 0.2)
    g_gpu, loss_gpu = _gp_like(yg, xg, [] if inst else [gg, bgm], 11)
    assert_close(g_gpu, g_ref.float(), 1e-5, "norm first-order dx (create_graph)")
    assert abs(float(loss_gpu) - float(loss_ref)) <= 2e-5 * abs(float(loss_ref))
    assert_close(xg.grad, x.grad.float(), 2e-5, "d(penalty)/dx through the norm backward")
    if not inst:
        assert_close(gg.grad, gamma.grad.float(), 2e-5, "d(penalty)/dgamma")


@pytest.mark.parametrize("act", [3, 4])
def test_activation_second_derivative(pg, act):
    """Tanh / Sigmoid under create_graph=True (dragan.py:92: the critic ends in Sigmoid)."""
    F = pg.functional
    x = (_leaf(6, 40, seed=1) * 2).double().requires_grad_(True)
    y = torch.tanh(x) if act == 3 else torch.sigmoid(x)
    g_ref, loss_ref = _gp_like(y, x, [], 5)
    xg = x.detach().float().to(DEV).requires_grad_(True)
    g_gpu, loss_gpu = _gp_like(F.activation(xg, act), xg, [], 5)
    assert_close(g_gpu, g_ref.float(), 2e-6, "act first-order")
    assert_close(xg.grad, x.grad.float(), 5e-6, "act second-order")


def test_raw_backward_refuses_create_graph(pg):
    F = pg.functional
    x = _leaf(2, 8, 6, 6, seed=1).to(DEV).requires_grad_(True)
    y = F.maxpool2(x)
    with pytest.raises(NotImplementedError):
        torch.autograd.grad(y.sum(), x, create_graph=True)


# ------------------------------------------------------------------------------------------------ conv-epilogue statistics
@pytest.mark.parametrize("cfg", [
    # N, Ci, H, W, Co, k, stride, pad, inst, masked, up
    (8, 128, 16, 16, 128, 3, 1, 1, False, False, True),    # dcgan.py:54-56 Upsample+Conv+BatchNorm (4 phase classes, 128x128)
    (8, 128, 16, 16, 64, 3, 1, 1, False, False, True),     # dcgan.py:58-60 (128x64 tap-inner kernel)
    (16, 16, 16, 16, 32, 3, 2, 1, False, True, False),     # dcgan.py:78-80 Conv+LeakyReLU+Dropout2d+BN, Ci=16 (K-tail), 64x64 tile
    (5, 32, 9, 7, 48, 3, 1, 1, False, False, False),       # ragged: 315 rows, partial last tile, Co tail
    (2, 64, 16, 16, 64, 3, 1, 1, True, False, False),      # InstanceNorm: groups = images, 256 pixels = 2-4 tiles per image
    (2, 256, 16, 16, 256, 3, 1, 1, True, False, False),    # cyclegan residual conv geometry (reflect handled below)
    (3, 32, 6, 6, 32, 3, 1, 1, True, False, False),        # InstanceNorm with 36 pixels per image: NOT supported -> fallback
])
def test_conv_epilogue_statistics(pg, cfg, monkeypatch):
    """Conv -> [act -> Dropout2d ->] BatchNorm / InstanceNorm with the statistics taken in the conv epilogue (per-tile
    mean / M2 / count + Chan combination) against the norm layer's own statistics pass: outputs, running statistics and
    all gradients agree to fp32 rounding."""
    N, Ci, H, W, Co, k, stride, pad, inst, masked, up = cfg
    F = pg.functional
    x = _leaf(N, Ci, H, W, seed=1).to(DEV)
    w = _leaf(Co, Ci, k, k, seed=2, scale=0.2).to(DEV)
    b = _leaf(Co, seed=3).to(DEV)
    mask = ((torch.rand(N, Co, generator=torch.Generator().manual_seed(4)) > 0.25).float() / 0.75).to(DEV) if masked else None
    res = {}
    for on in (True, False):
        monkeypatch.setattr(F, "_CONV_STATS", on)
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        nbt = torch.zeros((), dtype=torch.int64, device=DEV)
        gamma = None if inst else (torch.ones(Co, device=DEV) * 1.5).requires_grad_(True)
        beta = None if inst else torch.zeros(Co, device=DEV).requires_grad_(True)
        kind = "instance" if inst else "batch"
        if up:
            y = F.upconv3x3(xg, wg, b, F.ACT_NONE, 0.0, kind)
        else:
            gather = F.GATHER_REFLECT if (inst and Ci == 256) else F.GATHER_ZERO
            y = F.conv2d(xg, wg, b, stride, (pad,) * 4, gather, F.ACT_LRELU if masked else F.ACT_NONE, 0.2, mask, kind)
        used = hasattr(y, "_migan_stats")
        z = F.norm(y, gamma, beta, None, None if inst else rm, None if inst else rv, True, 0.1, 0.8 if masked else 1e-5, inst,
                   F.ACT_LRELU, 0.2, None if inst else nbt)
        gz = _leaf(*z.shape, seed=6).to(DEV)
        z.backward(gz)
        res[on] = (z.detach(), xg.grad, wg.grad, rm, rv, int(nbt), used)
    assert res[False][6] is False
    supported = not (inst and (H * W) % 64 != 0)
    assert res[True][6] is supported
    assert_close(res[True][0], res[False][0], 2e-6, "norm output, conv-epilogue statistics vs statistics pass")
    assert_close(res[True][1], res[False][1], 2e-5, "dx")
    assert_close(res[True][2], res[False][2], 2e-5, "dw")
    if not inst:
        assert_close(res[True][3], res[False][3], 2e-6, "running_mean")
        assert_close(res[True][4], res[False][4], 2e-6, "running_var")
        assert res[True][5] == res[False][5] == 1


@pytest.mark.parametrize("dims", [(64, 1024, 512, 1), (64, 128, 256, 0), (33, 512, 1024, 3), (64, 1024, 1024, 0), (16, 48, 32, 0),
                                  (64, 144, 80, 1)])
def test_skinny_linear(pg, dims, monkeypatch):
    """<= 64-row Linear forward / input gradient on the skinny MFMA kernels (wgan_gp.py:46-56,73-77 at batch 64) against
    torch CPU and against the tiled kernels they replace."""
    B, K, Nf, act = dims
    F = pg.functional
    from pytorch_gan_amd._lib import lib

    assert lib.migan_skinny_nt_ok(B, Nf, K) == 1 and (Nf % 16 != 0 or lib.migan_skinny_nn_ok(B, Nf, K) == (1 if K % 32 == 0 else 0))
    x = _leaf(B, K, seed=1).requires_grad_(True)
    w = _leaf(Nf, K, seed=2, scale=0.1).requires_grad_(True)
    b = _leaf(Nf, seed=3).requires_grad_(True)
    y_ref = TF.linear(x, w, b)
    gy = _leaf(B, Nf, seed=4)
    y_ref.backward(gy)
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(F, "_SKINNY", on)
        xg, wg, bg = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
        y = F.linear(xg, wg, bg)
        y.backward(gy.to(DEV))
        assert_close(y, y_ref, TOL_FWD, "skinny=%s linear fwd" % on)
        assert_close(xg.grad, x.grad, TOL_FWD, "skinny=%s linear dgrad" % on)
        assert_close(wg.grad, w.grad, TOL_WGRAD, "linear wgrad")
        assert_close(bg.grad, b.grad, TOL_BIAS, "linear bias (inside the wgrad launch for few rows)")
        outs[on] = (y.detach(), xg.grad)
    assert_close(outs[True][0], outs[False][0], 2e-6, "skinny vs tiled fwd")
    assert_close(outs[True][1], outs[False][1], 2e-6, "skinny vs tiled dgrad")
    # fused activation epilogue of the forward kernel (C ABI)
    if act:
        st = torch.cuda.current_stream().cuda_stream
        xa, wa, ba = x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV)
        out = torch.empty(B, Nf, device=DEV)
        assert lib.migan_skinny_nt(xa.data_ptr(), wa.data_ptr(), ba.data_ptr(), out.data_ptr(), B, Nf, K, act, 0.2, st) == 0
        ref = {1: lambda t: TF.leaky_relu(t, 0.2), 3: torch.tanh}[act](y_ref.detach())
        assert_close(out, ref, TOL_FWD, "skinny fused act")
    # weight + bias gradient in one direct launch (C ABI), overwrite and accumulate-into-bucket modes
    if lib.migan_skinny_tn_ok(B, Nf, K):
        st = torch.cuda.current_stream().cuda_stream
        g_d, x_d = gy.to(DEV), x.detach().to(DEV)
        dw = torch.full((Nf, K), 7.0, device=DEV)
        dbv = torch.full((Nf,), 7.0, device=DEV)
        assert lib.migan_skinny_tn(g_d.data_ptr(), x_d.data_ptr(), dw.data_ptr(), dbv.data_ptr(), B, Nf, K, 0, 0, st) == 0
        assert_close(dw, w.grad, TOL_WGRAD, "skinny_tn dw")
        assert_close(dbv, b.grad, TOL_BIAS, "skinny_tn db")
        assert lib.migan_skinny_tn(g_d.data_ptr(), x_d.data_ptr(), dw.data_ptr(), dbv.data_ptr(), B, Nf, K, 1, 1, st) == 0
        assert_close(dw, 2 * w.grad, TOL_WGRAD, "skinny_tn dw accumulate")
        assert_close(dbv, 2 * b.grad, TOL_BIAS, "skinny_tn db accumulate")
    else:
        assert Nf % 16 != 0 or K % 64 != 0
