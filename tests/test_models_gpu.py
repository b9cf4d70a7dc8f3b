"""Module-level parity: a reference network (oracle restatement, pinned bit-exact against the reference) is
deep-copied, swap()ped onto the HIP path and must reproduce the CPU forward output, every parameter
gradient, input gradients and BatchNorm side effects; checked both with and without peephole fusion and
against the committed golden fixtures generated from the real reference (tests/golden/)."""
import numpy as np
import pytest
import torch

from util import TOL_MODEL_FWD, TOL_MODEL_GRAD, assert_close, digest, gpu_copy, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pg():
    import pytorch_gan_amd as pg

    return pg


def _seed(s):
    import random

    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def _fwd_bwd(model, inputs, w=None, ctx=None):
    for p in model.parameters():
        p.grad = None
    ins = [t.clone().requires_grad_(True) for t in inputs]
    if ctx is not None:
        with ctx:
            out = model(*ins)
    else:
        out = model(*ins)
    if w is None:
        g = torch.Generator().manual_seed(123)
        w = torch.randn(out.shape, generator=g)
    (out * w.to(device=out.device, dtype=out.dtype)).sum().backward()
    return out.detach(), [t.grad for t in ins], w


def _noise_aware(gpu, cpu32, cpu64, tol, what):
    """The HIP result must be as close to the fp64 evaluation of the reference network as the reference's own
    fp32 CPU path is (x8 slack), or within the stated relative tolerance — whichever is larger.  Needed because
    some gradients are ill-conditioned in fp32 (e.g. conv bias in front of a norm layer is exactly 0 in exact
    arithmetic; tiny-batch BatchNorm backward cancels heavily: the CPU fp32 path itself is 2e-3 off fp64)."""
    g, c, t = gpu.detach().double().cpu(), cpu32.detach().double(), cpu64.detach().double()
    err_g, err_c, ref = (g - t).norm().item(), (c - t).norm().item(), t.norm().item()
    bound = tol * ref + 8.0 * err_c + 1e-12
    assert err_g <= bound, "%s: |hip-f64| %.3e > bound %.3e (|cpu32-f64| %.3e, |f64| %.3e)" % (what, err_g, bound, err_c, ref)


def _digest_vs_golden(gmodel, cpu_model, model64, keys, gold_digest, what):
    """Parameter-gradient digests (sum |g|, sum g^2 per tensor) recorded from the REAL reference (tests/golden/*:
    `*_keys`, `*_digest`; same seeded weights, inputs, output weighting and dropout masks as `_compare` uses) against the
    HIP gradients.  Noise-aware like `_noise_aware`: a conv bias in front of a norm layer has an exactly-zero true
    gradient, so its digest is rounding noise on every implementation (the reference's own fp32 run differs from the
    oracle's fp64 evaluation by 100 % there); the noise scale is taken from the two CPU fp32 runs vs fp64."""
    gp, cp, dp = dict(gmodel.named_parameters()), dict(cpu_model.named_parameters()), dict(model64.named_parameters())
    seen = strict = 0
    for k, gd in zip([str(x) for x in keys], gold_digest):
        if gp[k].grad is None:
            continue
        mine, c32, d64 = digest(gp[k].grad), digest(cp[k].grad), digest(dp[k].grad)
        for j, tol in ((1, 5e-3), (2, 1e-2)):
            noise = max(abs(c32[j] - d64[j]), abs(gd[j] - d64[j]))
            bound = tol * abs(gd[j]) + 8.0 * noise + 1e-12
            assert abs(mine[j] - gd[j]) <= bound, "%s %s digest[%d]: hip %.6e golden %.6e f64 %.6e cpu32 %.6e" % (
                what, k, j, mine[j], gd[j], d64[j], c32[j])
            strict += noise <= tol * abs(gd[j])
        seen += 1
    assert seen >= 1 and strict >= seen, "%s: only %d of %d digests were compared at the stated tolerance" % (what, strict, 2 * seen)


def _compare(pg, cpu_model, inputs, fused=True, tol_grad=TOL_MODEL_GRAD, gold=None, masks=None):
    import copy

    from oracle import reference_models as M

    pg.set_fusion(fused)
    try:
        gmodel = gpu_copy(cpu_model)
        model64 = copy.deepcopy(cpu_model).double()
        _seed(7)
        if masks is None:  # the oracle draws (and records) the masks all three evaluations share
            rec = []
            out_c, gin_c, w = _fwd_bwd(cpu_model, inputs, ctx=M.feed_masks(record=rec))
            masks = [m.numpy() for m in rec]
        else:              # masks recorded from the real reference run (golden fixture)
            out_c, gin_c, w = _fwd_bwd(cpu_model, inputs, ctx=M.feed_masks(masks=masks))
        out_d, gin_d, _ = _fwd_bwd(model64, [t.double() for t in inputs], w, ctx=M.feed_masks(masks=masks))
        out_g, gin_g, _ = _fwd_bwd(gmodel, [t.to(DEV) for t in inputs], w, ctx=pg.dropout_masks(masks))
        assert_close(out_g, out_c, TOL_MODEL_FWD, "forward")
        for a, b, d in zip(gin_g, gin_c, gin_d):
            if b is not None:
                _noise_aware(a, b, d, tol_grad, "input grad")
        gp = dict(gmodel.named_parameters())
        p64 = dict(model64.named_parameters())
        for k, p in cpu_model.named_parameters():
            if p.grad is None:
                assert gp[k].grad is None or float(gp[k].grad.abs().max()) == 0.0, k
                continue
            _noise_aware(gp[k].grad, p.grad, p64[k].grad, tol_grad, "grad " + k)
        gb = dict(gmodel.named_buffers())
        for k, b in cpu_model.named_buffers():
            if b.dtype.is_floating_point:
                assert_close(gb[k], b, 1e-5, "buffer " + k)
            else:
                assert torch.equal(gb[k].cpu(), b), "buffer %s (bit-exact)" % k
        assert list(gmodel.state_dict().keys()) == list(cpu_model.state_dict().keys())
        if gold is not None:
            _digest_vs_golden(gmodel, cpu_model, model64, gold[0], gold[1], gold[2])
        return out_g, gmodel
    finally:
        pg.set_fusion(True)


@pytest.mark.parametrize("fused", [True, False])
def test_dcgan(pg, golden_dir, fused):
    from oracle import reference_models as M

    gold = load_golden(golden_dir, "dcgan_32")
    _seed(0)
    G = M.DcganGenerator(32, 100, 1)
    G.apply(M.init_normal_dcgan)
    _seed(0)
    D = M.DcganDiscriminator(32, 1)
    D.apply(M.init_normal_dcgan)
    z, img = torch.from_numpy(gold["z"]), torch.from_numpy(gold["img"])
    out_g, _ = _compare(pg, G, [z], fused=fused, gold=(gold["g_keys"], gold["g_digest"], "dcgan G"))
    assert_close(out_g, torch.from_numpy(gold["gen"]), TOL_MODEL_FWD, "G vs golden (real reference output)")
    masks = [gold["mask_%02d" % i] for i in range(int(gold["n_masks"]))]  # Dropout2d masks the real reference drew
    _compare(pg, D, [img], fused=fused, gold=(gold["d_keys"], gold["d_digest"], "dcgan D"), masks=masks)
    _compare(pg, D, [img], fused=fused)  # and with masks drawn by the oracle
    # D against the reference run recorded in the fixture (its own dropout masks)
    Dg = gpu_copy(D)
    with pg.dropout_masks(masks):
        d_out = Dg(img.to(DEV))
    assert_close(d_out, torch.from_numpy(gold["d_out"]), TOL_MODEL_FWD, "D vs golden")


@pytest.mark.parametrize("fused", [True, False])
def test_mlp_gan_wgan(pg, golden_dir, fused):
    from oracle import reference_models as M

    gold = load_golden(golden_dir, "wgan_gp_32")
    _seed(0)
    G = M.MlpGenerator((1, 32, 32), 100)
    _seed(0)
    D = M.MlpCritic((1, 32, 32))
    out_g, _ = _compare(pg, G, [torch.from_numpy(gold["z"])], fused=fused, gold=(gold["g_keys"], gold["g_digest"], "wgan G"))
    assert_close(out_g, torch.from_numpy(gold["gen"]), TOL_MODEL_FWD, "wgan G vs golden")
    out_d, Dg = _compare(pg, D, [torch.from_numpy(gold["real"])], fused=fused, gold=(gold["d_keys"], gold["d_digest"], "wgan D"))
    assert_close(out_d, torch.from_numpy(gold["d_out"]), TOL_MODEL_FWD, "wgan D vs golden")
    # gradient penalty on the HIP path vs the value/gradients produced by the reference function
    from pytorch_gan_amd import steps

    for p in Dg.parameters():
        p.grad = None
    real, fake, alpha = (torch.from_numpy(gold[k]).to(DEV) for k in ("real", "gen", "alpha"))
    gp = steps.compute_gradient_penalty(Dg, real, fake, alpha)
    gp.backward()
    assert abs(gp.item() - float(gold["gp"])) <= 1e-5 * max(1.0, abs(float(gold["gp"])))
    keys = [str(k) for k in gold["gp_keys"]]
    named = dict(Dg.named_parameters())
    for k, dg in zip(keys, gold["gp_digest"]):
        mine = digest(named[k].grad)
        assert abs(mine[2] - dg[2]) <= 1e-4 * max(dg[2], 1e-12), "GP grad energy of %s" % k


@pytest.mark.parametrize("fused", [True, False])
def test_cyclegan(pg, golden_dir, fused):
    from oracle import reference_models as M

    gold = load_golden(golden_dir, "cyclegan_32")
    shape = (3, 32, 32)
    _seed(0)
    G = M.CycleGenerator(shape, 3)
    G.apply(M.init_normal_cyclegan)
    _seed(0)
    D = M.CycleDiscriminator(shape)
    D.apply(M.init_normal_cyclegan)
    x = torch.from_numpy(gold["x"])
    out_g, _ = _compare(pg, G, [x], fused=fused, gold=(gold["g_keys"], gold["g_digest"], "cyclegan G"))
    assert_close(out_g, torch.from_numpy(gold["gen"]), TOL_MODEL_FWD, "cyclegan G vs golden")
    out_d, _ = _compare(pg, D, [x], fused=fused, gold=(gold["d_keys"], gold["d_digest"], "cyclegan D"))
    assert_close(out_d, torch.from_numpy(gold["d_out"]), TOL_MODEL_FWD, "cyclegan D vs golden")


def test_srgan(pg, golden_dir):
    from oracle import reference_models as M

    gold = load_golden(golden_dir, "srgan_32")
    _seed(0)
    G = M.SrganGenerator()
    _seed(0)
    D = M.SrganDiscriminator((3, 32, 32))
    _seed(0)
    V = M.SrganFeatureExtractor()
    V.eval()
    lr, hr = torch.from_numpy(gold["lr"]), torch.from_numpy(gold["hr"])
    out_g, _ = _compare(pg, G, [lr], gold=(gold["g_keys"], gold["g_digest"], "srgan G"))
    assert_close(out_g, torch.from_numpy(gold["gen"]), TOL_MODEL_FWD, "srgan G vs golden")
    out_d, _ = _compare(pg, D, [hr], gold=(gold["d_keys"], gold["d_digest"], "srgan D"))
    assert_close(out_d, torch.from_numpy(gold["d_out"]), TOL_MODEL_FWD, "srgan D vs golden")
    out_v, _ = _compare(pg, V, [hr])
    assert np.allclose(digest(out_v), gold["vgg_digest"], rtol=1e-4)


def test_esrgan(pg, golden_dir):
    """esrgan/models.py:8-130: RRDB generator (channel concatenations, `out.mul(0.2) + x` through the
    GanTensor handlers, LeakyReLU(0.01), PixelShuffle), discriminator logits and VGG19[:35], swapped from the oracle
    restatement; outputs and gradient digests against the fixture recorded from the reference's own modules."""
    from oracle import reference_models as M

    gold = load_golden(golden_dir, "esrgan_32")
    _seed(0)
    G = M.EsrganGenerator(3, filters=64, num_res_blocks=2)
    _seed(0)
    D = M.EsrganDiscriminator((3, 32, 32))
    _seed(0)
    V = M.EsrganFeatureExtractor()
    V.eval()
    lr, hr = torch.from_numpy(gold["lr"]), torch.from_numpy(gold["hr"])
    out_g, _ = _compare(pg, G, [lr], gold=(gold["g_keys"], gold["g_digest"], "esrgan G"))
    assert_close(out_g, torch.from_numpy(gold["gen"]), TOL_MODEL_FWD, "esrgan G vs golden")
    out_d, _ = _compare(pg, D, [hr])
    assert_close(out_d, torch.from_numpy(gold["d_out"]), TOL_MODEL_FWD, "esrgan D vs golden")
    out_v, _ = _compare(pg, V, [hr])
    assert np.allclose(digest(out_v), gold["vgg_digest"], rtol=1e-4)


def test_product_models_equal_swapped_oracle(pg):
    """pytorch_gan_amd.models (networks written directly on the HIP layer set) against swap(oracle network) with the same
    state_dict: same kernels underneath, so outputs agree to rounding (the ESRGAN blocks fuse `out*0.2 + x` into one
    launch, the swapped reference code runs it as two)."""
    from oracle import reference_models as OM
    from pytorch_gan_amd import models as PM

    _seed(3)
    z = torch.randn(8, 100)
    img1 = torch.rand(8, 1, 32, 32) * 2 - 1
    img3 = torch.rand(2, 3, 32, 32) * 2 - 1
    lr = torch.randn(2, 3, 8, 8)
    cases = [
        (PM.DcganGenerator(32), OM.DcganGenerator(32), [z]),
        (PM.DcganDiscriminator(32), OM.DcganDiscriminator(32), [img1]),
        (PM.MlpGenerator(), OM.MlpGenerator(), [z]),
        (PM.MlpCritic(), OM.MlpCritic(), [img1]),
        (PM.CycleGenerator((3, 32, 32), 2), OM.CycleGenerator((3, 32, 32), 2), [img3]),
        (PM.CycleDiscriminator((3, 32, 32)), OM.CycleDiscriminator((3, 32, 32)), [img3]),
        (PM.Pix2pixDiscriminator(), OM.Pix2pixDiscriminator(), [img3, img3.flip(0)]),
        (PM.SrganGenerator(n_residual_blocks=2), OM.SrganGenerator(n_residual_blocks=2), [lr]),
        (PM.SrganDiscriminator((3, 32, 32)), OM.SrganDiscriminator((3, 32, 32)), [img3]),
        (PM.SrganFeatureExtractor(), OM.SrganFeatureExtractor(), [img3]),
        (PM.EsrganGenerator(3, 64, 2), OM.EsrganGenerator(3, 64, 2), [lr]),
        (PM.EsrganDiscriminator((3, 32, 32)), OM.EsrganDiscriminator((3, 32, 32)), [img3]),
        (PM.EsrganFeatureExtractor(), OM.EsrganFeatureExtractor(), [img3]),
    ]
    for prod, orc, ins in cases:
        prod.load_state_dict(orc.state_dict())
        prod, sw = prod.to(DEV), gpu_copy(orc)
        prod.eval()   # no dropout draws, running statistics: the comparison is of the layer wiring
        sw.eval()
        with torch.no_grad():
            a = prod(*[t.to(DEV) for t in ins])
            b = sw(*[t.to(DEV) for t in ins])
        assert a.shape == b.shape
        assert_close(a, b, 2e-6, type(prod).__name__ + " product vs swapped oracle")
        if any(isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)) for m in prod.modules()):
            continue
        # training mode, forward + backward: the product blocks fuse the residual add into the last norm launch
        # (Sequential(..., res=x)) and the ESRGAN blocks `out*0.2 + x` into one axpby - gradients must not notice
        prod.train()
        sw.train()
        g = torch.Generator().manual_seed(5)
        wgt = None
        grads = []
        for net in (prod, sw):
            for p_ in net.parameters():
                p_.grad = None
            xin = [t.to(DEV).requires_grad_(True) for t in ins]
            out = net(*xin)
            if wgt is None:
                wgt = torch.randn(out.shape, generator=g).to(DEV)
            (out * wgt).sum().backward()
            grads.append(([t.grad for t in xin], [p_.grad for p_ in net.parameters()], out.detach()))
        assert_close(grads[0][2], grads[1][2], 2e-6, type(prod).__name__ + " train-mode forward")
        for ga, gb in zip(grads[0][0], grads[1][0]):
            assert_close(ga, gb, 2e-5, type(prod).__name__ + " input gradient, product vs swapped oracle")
        num = sum(float(((ga - gb).double() ** 2).sum()) for ga, gb in zip(grads[0][1], grads[1][1]) if ga is not None)
        den = sum(float((gb.double() ** 2).sum()) for gb in grads[1][1] if gb is not None)
        assert (num / max(den, 1e-300)) ** 0.5 < 2e-5, type(prod).__name__ + " parameter gradients"


def test_pix2pix(pg, golden_dir):
    from oracle import reference_models as M

    gold = load_golden(golden_dir, "pix2pix_256")
    _seed(0)
    G = M.Pix2pixGenerator()
    G.apply(M.init_normal_dcgan)
    _seed(0)
    D = M.Pix2pixDiscriminator()
    D.apply(M.init_normal_dcgan)
    _seed(5)
    a = torch.rand(1, 3, 256, 256) * 2 - 1
    b = torch.rand(1, 3, 256, 256) * 2 - 1
    # the element-dropout masks the real reference drew (packed keep-bits in the fixture)
    masks = []
    for i, (shp, keep) in enumerate(zip(gold["mask_shapes"], gold["mask_keep"])):
        n = int(np.prod(shp))
        bits = np.unpackbits(gold["mask_bits_%02d" % i])[:n].reshape([int(v) for v in shp])
        masks.append(bits.astype(np.float32) * np.float32(keep))
    out_g, _ = _compare(pg, G, [a], gold=(gold["g_keys"], gold["g_digest"], "pix2pix G"), masks=masks)
    # U-Net output of the REAL reference (54 M parameters, element dropout masks regenerated from the recorded seed)
    assert np.allclose(digest(out_g), gold["gen_digest"], rtol=1e-4), (digest(out_g), gold["gen_digest"])
    assert np.allclose(out_g.flatten()[:64].cpu().numpy(), gold["gen_head"], rtol=1e-3, atol=1e-5)
    out_d, _ = _compare(pg, D, [b, a], gold=(gold["d_keys"], gold["d_digest"], "pix2pix D"))
    assert_close(out_d, torch.from_numpy(gold["d_out"]), TOL_MODEL_FWD, "pix2pix D vs golden")


def test_swap_keeps_tree_and_init(pg):
    """weights_init_normal dispatches on class names and writes .weight.data (dcgan.py:36-42): must still work."""
    from oracle import reference_models as M

    G = M.DcganGenerator(32, 100, 1)
    keys = list(G.state_dict().keys())
    pg.swap(G)
    assert list(G.state_dict().keys()) == keys
    names = [type(m).__name__ for m in G.modules()]
    assert "Conv2d" in names and "BatchNorm2d" in names and "Upsample" in names
    G.apply(M.init_normal_dcgan)
    with pytest.raises(NotImplementedError):
        pg.swap(torch.nn.Sequential(torch.nn.GRU(4, 4)))


def test_dragan_gradient_penalty_vs_reference(pg, golden_dir):
    """dragan.py:144-167 on the HIP path - double backward through Conv2d / LeakyReLU / Dropout2d /
    BatchNorm2d(eps .8) / Linear / Sigmoid - against the value and the discriminator gradients recorded from the REAL
    reference function (tests/golden/dragan_32.npz; same X, alpha, noise and Dropout2d masks) and against the oracle."""
    from oracle import reference_models as M
    from oracle import reference_steps as S
    from pytorch_gan_amd import steps

    gold = load_golden(golden_dir, "dragan_32")
    _seed(0)
    M.DcganGenerator(32, 100, 1).apply(M.init_normal_dcgan)   # the pin script builds G first (same RNG consumption)
    _seed(0)
    D = M.DcganDiscriminator(32, 1)
    D.apply(M.init_normal_dcgan)
    X, alpha, noise = (torch.from_numpy(gold[k]) for k in ("X", "alpha", "noise"))
    masks = [gold["mask_%02d" % i] for i in range(int(gold["n_masks"]))]
    import copy

    D64 = copy.deepcopy(D).double()
    Dg = gpu_copy(D)
    for p in D.parameters():
        p.grad = None
    with M.feed_masks(masks=masks):
        gp_c = S.dragan_gradient_penalty(D, X, alpha, noise, 10)
    gp_c.backward()
    assert abs(float(gp_c) - float(gold["gp"])) <= 1e-6 * abs(float(gold["gp"]))
    with M.feed_masks(masks=masks):
        gp_d = S.dragan_gradient_penalty(D64, X.double(), alpha.double(), noise.double(), 10)
    gp_d.backward()
    for p in Dg.parameters():
        p.grad = None
    with pg.dropout_masks(masks):
        gp_g = steps.compute_gradient_penalty_dragan(Dg, X.to(DEV), alpha.to(DEV), noise.to(DEV), 10)
    gp_g.backward()
    assert abs(float(gp_g) - float(gold["gp"])) <= 2e-5 * abs(float(gold["gp"])), (float(gp_g), float(gold["gp"]))
    gp_, cp_, dp_ = dict(Dg.named_parameters()), dict(D.named_parameters()), dict(D64.named_parameters())
    keys = [str(k) for k in gold["gp_keys"]]
    for k, gd in zip(keys, gold["gp_digest"]):
        assert np.allclose(digest(cp_[k].grad), gd, rtol=1e-3, atol=1e-9 + 1e-3 * abs(gd[1])), k
        _noise_aware(gp_[k].grad, cp_[k].grad, dp_[k].grad, TOL_MODEL_GRAD, "dragan penalty grad " + k)
    # parameters the penalty does not reach have no gradient on either side
    for k, p in cp_.items():
        if p.grad is None:
            assert gp_[k].grad is None or float(gp_[k].grad.abs().max()) == 0.0, k
    # BatchNorm side effects of the penalty's discriminator forward
    for (k, b), (_, c) in zip(D.named_buffers(), Dg.named_buffers()):
        if b.dtype.is_floating_point:
            assert_close(c, b, 1e-5, "dragan buffer " + k)


@pytest.mark.parametrize("name", ["stargan", "dualgan"])
def test_conv_critic_gradient_penalty_vs_reference(pg, golden_dir, name):
    """SURVEY.md 8f F1, the other two scripts: compute_gradient_penalty of stargan.py:142-161 (critic stargan/models.py:87-115:
    Conv4x4 s2 + LeakyReLU(0.01), tuple output) and dualgan.py:116-135 (dualgan/models.py:102-123: BatchNorm2d(C, 0.8) and a
    ZeroPad2d inside the twice-differentiated path) on the HIP path against the value and the critic gradients recorded from
    the REAL reference functions (tests/golden/critic_gp_32.npz) and against the oracle's fp64 evaluation."""
    import copy

    from oracle import reference_models as M
    from oracle import reference_steps as S
    from pytorch_gan_amd import steps

    gold = load_golden(golden_dir, "critic_gp_32")
    # the fixture inputs were drawn so that no LeakyReLU pre-activation lies within 2e-5 (relative to the layer rms) of the kink:
    # an element within rounding of 0 takes the other branch on another implementation and moves these twice-differentiated
    # gradients by ~1 % (measured with the first draw: 0.8 % on model.0.weight from ONE element of 114 k) - DESIGN.md 4
    assert float(gold[name + "_kink_margin"]) > 2e-5
    _seed(0)
    D = M.StarganDiscriminator((3, 32, 32), 5, 4) if name == "stargan" else M.DualganDiscriminator(3)
    real, fake, alpha = (torch.from_numpy(gold["%s_%s" % (name, k)]) for k in ("real", "fake", "alpha"))
    D64 = copy.deepcopy(D).double()
    Dg = gpu_copy(D)
    gp_c = S.critic_gradient_penalty(D, real, fake, alpha)
    gp_c.backward()
    want = float(gold[name + "_gp"])
    assert abs(float(gp_c.detach()) - want) <= 1e-6 * abs(want)
    gp_d = S.critic_gradient_penalty(D64, real.double(), fake.double(), alpha.double())
    gp_d.backward()
    for p in Dg.parameters():
        p.grad = None
    gp_g = steps.compute_gradient_penalty(Dg, real.to(DEV), fake.to(DEV), alpha.to(DEV))
    gp_g.backward()
    # the stated loss tolerance of the step tests (1e-4 * max(1, |loss|), tests/util.py): measured 2.0e-5 relative on the dualgan
    # critic (K = 4096 reductions accumulated in one fp32 MFMA chain per output, four layers forward and twice backward)
    assert abs(float(gp_g.detach()) - want) <= 1e-4 * max(1.0, abs(want)), (float(gp_g.detach()), want)
    gp_, cp_, dp_ = dict(Dg.named_parameters()), dict(D.named_parameters()), dict(D64.named_parameters())
    keys = [str(k) for k in gold[name + "_keys"]]
    assert len(keys) >= 6
    for k, gd in zip(keys, gold[name + "_digest"]):
        assert np.allclose(digest(cp_[k].grad), gd, rtol=1e-3, atol=1e-9 + 1e-3 * abs(gd[1])), k
        if gp_[k].grad is None:
            # the last layer's bias does not enter dD/dx: torch hands back a zero tensor, the HIP Functions no gradient at all
            # (under optim.Adam the parameter's .grad is a zeroed slot of the flat bucket either way)
            assert float(cp_[k].grad.abs().max()) == 0.0, k
            continue
        _noise_aware(gp_[k].grad, cp_[k].grad, dp_[k].grad, TOL_MODEL_GRAD, "%s penalty grad %s" % (name, k))
    for k, p in cp_.items():   # stargan's class head (out2) is not on the penalty's path: no gradient on either side
        if p.grad is None:
            assert gp_[k].grad is None or float(gp_[k].grad.abs().max()) == 0.0, k
    for (k, b), (_, c) in zip(D.named_buffers(), Dg.named_buffers()):
        if b.dtype.is_floating_point:
            assert_close(c, b, 1e-5, "%s buffer %s" % (name, k))



CLONE_NAMES = ["lsgan", "sgan", "infogan", "relativistic_gan", "cogan", "began", "ebgan"]


def _multi_fwd_bwd(model, inputs, ctx):
    for p in model.parameters():
        p.grad = None
    ins = [t.clone().requires_grad_(t.is_floating_point()) for t in inputs]
    with ctx:
        outs = model(*ins)
    outs = outs if isinstance(outs, tuple) else (outs,)
    g = torch.Generator().manual_seed(123)
    loss = 0
    for o in outs:
        w = torch.randn(o.shape, generator=g)
        loss = loss + (o * w.to(device=o.device, dtype=o.dtype)).sum()
    loss.backward()
    return [o.detach() for o in outs], [t.grad for t in ins]


@pytest.mark.parametrize("name", CLONE_NAMES)
def test_clone_models(pg, golden_dir, name):
    """swap() of every DCGAN-block clone (lsgan, sgan, infogan, relativistic_gan, cogan, began, ebgan - their
    generators and single / multi-head / coupled / auto-encoder discriminators): outputs vs the values recorded from the
    reference's own classes, input and parameter gradients vs the fp64-anchored oracle, BatchNorm buffers, state_dict keys."""
    import contextlib
    import copy
    import warnings

    from oracle import reference_models as M

    gold = load_golden(golden_dir, "clone_%s_32" % name)
    _seed(0)
    G, D, init = M.clone_models(name)
    if init is not None:
        G.apply(init)
        D.apply(init)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, net in (("g", G), ("d", D)):
            ins = [torch.from_numpy(gold["%s_in%d" % (tag, i)]) for i in range(8) if "%s_in%d" % (tag, i) in gold.files]
            masks = [gold["mask_%02d" % i] for i in range(int(gold["n_masks"]))] if tag == "d" else []
            gnet, net64 = gpu_copy(net), copy.deepcopy(net).double()
            cctx = (lambda: M.feed_masks(masks=masks)) if masks else contextlib.nullcontext
            out_c, gin_c = _multi_fwd_bwd(net, ins, cctx())
            out_d, gin_d = _multi_fwd_bwd(net64, [t.double() for t in ins], cctx())
            out_g, gin_g = _multi_fwd_bwd(gnet, [t.to(DEV) for t in ins], pg.dropout_masks(masks) if masks else contextlib.nullcontext())
            for i, (a, b) in enumerate(zip(out_g, out_c)):
                assert_close(a, b, TOL_MODEL_FWD, "%s %s output %d vs oracle" % (name, tag, i))
                assert_close(a, torch.from_numpy(gold["%s_out%d" % (tag, i)]), TOL_MODEL_FWD, "%s %s output %d vs the reference" % (name, tag, i))
            if tag == "d":
                for a, b, d in zip(gin_g, gin_c, gin_d):
                    _noise_aware(a, b, d, TOL_MODEL_GRAD, "%s input grad" % name)
            gp, p64 = dict(gnet.named_parameters()), dict(net64.named_parameters())
            for k, p in net.named_parameters():
                if p.grad is None:
                    assert gp[k].grad is None or float(gp[k].grad.abs().max()) == 0.0, k
                    continue
                _noise_aware(gp[k].grad, p.grad, p64[k].grad, TOL_MODEL_GRAD, "%s grad %s" % (name, k))
            gb = dict(gnet.named_buffers())
            for k, b in net.named_buffers():
                if b.dtype.is_floating_point:
                    assert_close(gb[k], b, 1e-5, "buffer " + k)
                else:
                    assert torch.equal(gb[k].cpu(), b), k
            assert list(gnet.state_dict().keys()) == list(net.state_dict().keys())
            _digest_vs_golden(gnet, net, net64, gold[tag + "_keys"], gold[tag + "_digest"], "%s %s" % (name, tag))
