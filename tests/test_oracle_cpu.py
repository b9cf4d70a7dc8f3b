"""CPU suite (no GPU): the oracle restatement against the golden fixtures that oracle/pin_against_reference.py
generated from the REAL reference classes (run in the build container where /root/reference exists).
Weights are not stored: they are re-created from the recorded seeds (same torch build => identical init)."""
import random

import numpy as np
import pytest
import torch

from util import digest, load_golden

from oracle import reference_models as M
from oracle import reference_steps as S

TOL = 2e-5  # same torch CPU build reproduces these bit-exactly; tolerance covers a different oneDNN/thread count


def _seed(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def _close(a, b, tol=TOL):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    den = b.norm().item() or 1.0
    assert (a - b).norm().item() / den <= tol, (a - b).norm().item() / den


def _fwd_bwd(model, inputs, masks=None):
    for p in model.parameters():
        p.grad = None
    ins = [t.clone().requires_grad_(True) for t in inputs]
    with M.feed_masks(masks=masks):
        out = model(*ins)
    g = torch.Generator().manual_seed(123)
    w = torch.randn(out.shape, generator=g)
    (out * w).sum().backward()
    return out.detach(), {k: p.grad for k, p in model.named_parameters() if p.grad is not None}, [t.grad for t in ins]


def _check_grads(grads, keys, digests, tol=5e-3):
    # digest = (sum, sum|.|, sum .^2); sums cancel, so compare the energy and the L1 mass
    emax = max(float(d[2]) for d in digests)
    for k, d in zip([str(k) for k in keys], digests):
        mine = digest(grads[k])
        if d[2] < 1e-9 * emax:
            # rounding noise (e.g. a bias in front of a norm layer has an exactly-zero true gradient): its value
            # depends on the thread count / reduction order, only its smallness is checked
            assert mine[2] < 1e-6 * emax, k
        else:
            assert abs(mine[2] - d[2]) <= tol * d[2], (k, mine, d)
            assert abs(mine[1] - d[1]) <= tol * d[1], (k, mine, d)


def test_dcgan_against_reference_fixture(golden_dir):
    gold = load_golden(golden_dir, "dcgan_32")
    _seed(0)
    G = M.DcganGenerator(32, 100, 1)
    G.apply(M.init_normal_dcgan)
    _seed(0)
    D = M.DcganDiscriminator(32, 1)
    D.apply(M.init_normal_dcgan)
    out, grads, _ = _fwd_bwd(G, [torch.from_numpy(gold["z"])])
    _close(out, gold["gen"])
    _check_grads(grads, gold["g_keys"], gold["g_digest"])
    _close(G.state_dict()["conv_blocks.0.running_mean"], gold["g_bn_rm"])
    _close(G.state_dict()["conv_blocks.3.running_var"], gold["g_bn_rv"])
    masks = [gold["mask_%02d" % i] for i in range(int(gold["n_masks"]))]
    out, grads, gin = _fwd_bwd(D, [torch.from_numpy(gold["img"])], masks)
    _close(out, gold["d_out"])
    _close(gin[0], gold["d_in_grad"], 5e-3)
    _check_grads(grads, gold["d_keys"], gold["d_digest"])
    # labels / index tensors are bit-exact by construction
    assert torch.equal(torch.ones(4, 1), torch.full((4, 1), 1.0))


def test_wgan_gp_and_gan_against_reference_fixture(golden_dir):
    gold = load_golden(golden_dir, "wgan_gp_32")
    _seed(0)
    G = M.MlpGenerator((1, 32, 32), 100)
    _seed(0)
    D = M.MlpCritic((1, 32, 32))
    out, grads, _ = _fwd_bwd(G, [torch.from_numpy(gold["z"])])
    _close(out, gold["gen"])
    _check_grads(grads, gold["g_keys"], gold["g_digest"])
    out, grads, _ = _fwd_bwd(D, [torch.from_numpy(gold["real"])])
    _close(out, gold["d_out"])
    for p in D.parameters():
        p.grad = None
    gp = S.gradient_penalty(D, torch.from_numpy(gold["real"]), torch.from_numpy(gold["gen"]),
                            torch.from_numpy(gold["alpha"]))
    gp.backward()
    assert abs(gp.item() - float(gold["gp"])) <= 1e-6 * max(1.0, abs(float(gold["gp"])))
    _check_grads({k: p.grad for k, p in D.named_parameters() if p.grad is not None}, gold["gp_keys"], gold["gp_digest"])
    assert D.model[4].bias.grad is None  # last-layer bias takes no gradient from the penalty

    gold = load_golden(golden_dir, "gan_28")
    _seed(0)
    G = M.MlpGenerator((1, 28, 28), 100)
    _seed(0)
    D = M.MlpCritic((1, 28, 28), sigmoid=True)
    out, grads, _ = _fwd_bwd(G, [torch.from_numpy(gold["z"])])
    _close(out, gold["gen"])
    out, _, _ = _fwd_bwd(D, [torch.from_numpy(gold["gen"])])
    _close(out, gold["d_out"])


def test_cyclegan_against_reference_fixture(golden_dir):
    gold = load_golden(golden_dir, "cyclegan_32")
    shape = (3, 32, 32)
    _seed(0)
    G = M.CycleGenerator(shape, 3)
    G.apply(M.init_normal_cyclegan)
    _seed(0)
    D = M.CycleDiscriminator(shape)
    D.apply(M.init_normal_cyclegan)
    x = torch.from_numpy(gold["x"])
    out, grads, gin = _fwd_bwd(G, [x])
    _close(out, gold["gen"])
    _close(gin[0], gold["g_in_grad"], 5e-3)
    _check_grads(grads, gold["g_keys"], gold["g_digest"])
    out, grads, gin = _fwd_bwd(D, [x])
    _close(out, gold["d_out"])
    _close(gin[0], gold["d_in_grad"], 5e-3)
    # host-side index logic: bit-exact
    random.seed(11)
    buf = M.ReplayBuffer(max_size=3)
    for i in range(6):
        batch = torch.full((2, 1, 2, 2), float(i)) + torch.tensor([0.0, 0.5]).view(2, 1, 1, 1)
        got = buf.push_and_pop(batch)[:, 0, 0, 0].numpy()
        assert np.array_equal(got, gold["replay_picks"][i])
    lam = M.lambda_lr(200, 0, 100)
    assert [lam(e) for e in (0, 50, 100, 101, 150, 199)] == list(gold["lr_factors"])


def test_srgan_against_reference_fixture(golden_dir):
    gold = load_golden(golden_dir, "srgan_32")
    _seed(0)
    G = M.SrganGenerator()
    _seed(0)
    D = M.SrganDiscriminator((3, 32, 32))
    _seed(0)
    V = M.SrganFeatureExtractor()
    V.eval()
    lr, hr = torch.from_numpy(gold["lr"]), torch.from_numpy(gold["hr"])
    out, grads, _ = _fwd_bwd(G, [lr])
    _close(out, gold["gen"])
    _check_grads(grads, gold["g_keys"], gold["g_digest"], tol=2e-2)
    out, grads, gin = _fwd_bwd(D, [hr])
    _close(out, gold["d_out"])
    out, _, gin = _fwd_bwd(V, [hr])
    assert np.allclose(digest(out), gold["vgg_digest"], rtol=1e-5)
    _close(gin[0], gold["vgg_in_grad"], 5e-3)
    assert D.output_shape == (1, 2, 2)


def test_esrgan_against_reference_fixture(golden_dir):
    """esrgan/models.py:8-130 and the loop body esrgan.py:101-174 (one warm-up step, two relativistic steps) against the
    fixture recorded with the reference's own modules."""
    gold = load_golden(golden_dir, "esrgan_32")
    _seed(0)
    G = M.EsrganGenerator(3, filters=64, num_res_blocks=2)
    _seed(0)
    D = M.EsrganDiscriminator((3, 32, 32))
    _seed(0)
    V = M.EsrganFeatureExtractor()
    V.eval()
    lr, hr = torch.from_numpy(gold["lr"]), torch.from_numpy(gold["hr"])
    out, grads, _ = _fwd_bwd(G, [lr])
    _close(out, gold["gen"])
    _check_grads(grads, gold["g_keys"], gold["g_digest"], tol=2e-2)
    out, grads, gin = _fwd_bwd(D, [hr])
    _close(out, gold["d_out"])
    out, _, gin = _fwd_bwd(V, [hr])
    assert np.allclose(digest(out), gold["vgg_digest"], rtol=1e-5)
    _close(gin[0], gold["vgg_in_grad"], 5e-3)
    assert len(list(V.vgg19_54.children())) == 35 and isinstance(V.vgg19_54[34], torch.nn.Conv2d)  # conv5_4, no ReLU
    _seed(0)
    s = S.make_esrgan((32, 32), n_res=1)
    s.warmup_batches = 1
    lrs, hrs = torch.from_numpy(gold["loop_lr"]), torch.from_numpy(gold["loop_hr"])
    keys = [str(k) for k in gold["loop_keys"]]
    for t in range(3):
        o = S.esrgan_step(s, lrs[t], hrs[t], t)
        assert (len(o) == 1) == (t == 0)   # warm-up iteration reports the pixel loss only
        for j, k in enumerate(keys):
            if k in o:
                assert abs(float(o[k]) - float(gold["loop_trace"][t][j])) <= 1e-5 * max(1.0, abs(float(gold["loop_trace"][t][j]))), (t, k)
            else:
                assert np.isnan(gold["loop_trace"][t][j])


def test_conv_critic_penalties_against_reference_fixture(golden_dir):
    """stargan.py:142-161 / dualgan.py:116-135 restated (oracle.reference_steps.critic_gradient_penalty) on the restated critics
    reproduce the values and critic-gradient digests recorded from the reference's own functions and classes."""
    gold = load_golden(golden_dir, "critic_gp_32")
    for name in ("stargan", "dualgan"):
        _seed(0)
        D = M.StarganDiscriminator((3, 32, 32), 5, 4) if name == "stargan" else M.DualganDiscriminator(3)
        gp = S.critic_gradient_penalty(D, *(torch.from_numpy(gold["%s_%s" % (name, k)]) for k in ("real", "fake", "alpha")))
        gp.backward()
        assert abs(float(gp.detach()) - float(gold[name + "_gp"])) <= 1e-6 * abs(float(gold[name + "_gp"]))
        grads = dict(D.named_parameters())
        for k, gd in zip([str(k) for k in gold[name + "_keys"]], gold[name + "_digest"]):
            assert np.allclose(digest(grads[k].grad), gd, rtol=1e-4, atol=1e-10), (name, k)
    assert dict(M.StarganDiscriminator((3, 32, 32), 5, 4).named_parameters())["out2.weight"].shape == (5, 512, 2, 2)


def test_acgan_loop_against_reference_fixture(golden_dir):
    """acgan.py:46-107,167-222 restated: three iterations reproduce the losses recorded with the reference's own Generator /
    Discriminator classes inside the loop (same images, labels, z, generated labels and Dropout2d masks)."""
    gold = load_golden(golden_dir, "acgan_32_loop")
    _seed(0)
    s = S.make_acgan(32)
    n = int(gold["masks_per_step"])
    for t in range(3):
        masks = [gold["mask_%d_%02d" % (t, i)] for i in range(n)]
        with M.feed_masks(masks=masks):
            o = S.acgan_step(s, torch.from_numpy(gold["imgs"][t]), torch.from_numpy(gold["labels"][t]),
                             torch.from_numpy(gold["zs"][t]), torch.from_numpy(gold["gen_labels"][t]))
        assert abs(float(o["g_loss"]) - gold["trace"][t][0]) <= 1e-6 and abs(float(o["d_loss"]) - gold["trace"][t][1]) <= 1e-6, t


def test_loop_traces_against_reference_fixture(golden_dir):
    """The restated loops, driven from the same seeds, reproduce the traces recorded with the REAL reference
    modules inside the same loop (dcgan 3 steps, wgan_gp 6 critic iterations, cyclegan 3 steps)."""
    gold = load_golden(golden_dir, "dcgan_32_loop")
    _seed(0)
    s = S.make_dcgan(32)
    n = int(gold["masks_per_step"])
    for t in range(3):
        masks = [gold["mask_%d_%02d" % (t, i)] for i in range(n)]
        with M.feed_masks(masks=masks):
            o = S.dcgan_step(s, torch.from_numpy(gold["imgs"][t]), torch.from_numpy(gold["zs"][t]))
        assert abs(o["g_loss"].item() - gold["trace"][t][0]) <= 1e-5
        assert abs(o["d_loss"].item() - gold["trace"][t][1]) <= 1e-5
    for (k, v), d in zip(s.G.state_dict().items(), gold["g_final_digest"]):
        assert np.allclose(digest(v.float()), d, rtol=1e-3, atol=1e-6), k

    gold = load_golden(golden_dir, "wgan_gp_32_loop")
    _seed(0)
    s = S.make_wgan_gp(32)
    for i in range(6):
        o = S.wgan_gp_step(s, torch.from_numpy(gold["reals"][i]), i, torch.from_numpy(gold["zs"][i]),
                           torch.from_numpy(gold["alphas"][i]))
        assert abs(o["d_loss"].item() - gold["trace"][i][0]) <= 1e-5 * max(1, abs(gold["trace"][i][0]))
        assert abs(o["gp"].item() - gold["trace"][i][1]) <= 1e-5 * max(1, abs(gold["trace"][i][1]))
        assert ("g_loss" in o) == (i % 5 == 0)

    gold = load_golden(golden_dir, "cyclegan_32_loop")
    _seed(0)
    s = S.make_cyclegan((3, 32, 32), 2)
    for t in range(3):
        random.seed(50 + t)
        o = S.cyclegan_step(s, torch.from_numpy(gold["A"][t]), torch.from_numpy(gold["B"][t]))
        got = [o[k].item() for k in ("loss_G", "loss_D", "loss_GAN", "loss_cycle", "loss_identity")]
        assert np.allclose(got, gold["trace"][t], rtol=2e-4, atol=1e-5), (got, gold["trace"][t])


def test_batchnorm_positional_eps():
    """nn.BatchNorm2d(C, 0.8) sets eps=0.8; both variants exist in dcgan.py:53,56."""
    G = M.DcganGenerator(32)
    assert G.conv_blocks[0].eps == 1e-5 and G.conv_blocks[3].eps == 0.8 and G.conv_blocks[7].eps == 0.8
    assert M.MlpGenerator().model[3].eps == 0.8


def test_pix2pix_shapes_and_keys():
    G, D = M.Pix2pixGenerator(), M.Pix2pixDiscriminator()
    assert sum(p.numel() for p in G.parameters()) == 54404099
    assert sum(p.numel() for p in D.parameters()) == 2767808
    assert "down1.model.0.weight" in G.state_dict() and "final.2.weight" in G.state_dict()
    assert G.up1.model[0].weight.shape == (512, 512, 4, 4)  # ConvTranspose2d weight is (Cin, Cout, kh, kw)
    assert "model.12.weight" in D.state_dict() and "model.12.bias" not in D.state_dict()


def test_dragan_penalty_against_reference_fixture(golden_dir):
    """the restated dragan.py:144-167 reproduces the value and the discriminator gradients recorded from
    the real reference function (same X, alpha, noise and Dropout2d masks)."""
    gold = load_golden(golden_dir, "dragan_32")
    _seed(0)
    M.DcganGenerator(32, 100, 1).apply(M.init_normal_dcgan)
    _seed(0)
    D = M.DcganDiscriminator(32, 1)
    D.apply(M.init_normal_dcgan)
    masks = [gold["mask_%02d" % i] for i in range(int(gold["n_masks"]))]
    with M.feed_masks(masks=masks):
        gp = S.dragan_gradient_penalty(D, torch.from_numpy(gold["X"]), torch.from_numpy(gold["alpha"]),
                                       torch.from_numpy(gold["noise"]), 10)
    gp.backward()
    assert abs(gp.item() - float(gold["gp"])) <= 1e-6 * abs(float(gold["gp"]))
    named = dict(D.named_parameters())
    for k, d in zip([str(k) for k in gold["gp_keys"]], gold["gp_digest"]):
        assert np.allclose(digest(named[k].grad), d, rtol=1e-3, atol=1e-9 + 1e-3 * abs(d[1])), k


CLONE_NAMES = ["lsgan", "sgan", "infogan", "relativistic_gan", "cogan", "began", "ebgan"]


def _clone_fwd_bwd(model, inputs, ctx, in_grad=False):
    for p in model.parameters():
        p.grad = None
    ins = [x.clone().requires_grad_(in_grad) for x in inputs]
    with ctx:
        outs = model(*ins)
    outs = outs if isinstance(outs, tuple) else (outs,)
    g = torch.Generator().manual_seed(123)
    loss = 0
    for o in outs:
        loss = loss + (o * torch.randn(o.shape, generator=g)).sum()
    loss.backward()
    return [o.detach() for o in outs], [x.grad for x in ins]


@pytest.mark.parametrize("name", CLONE_NAMES)
def test_clone_models_against_reference_fixture(golden_dir, name):
    """the restated generator / discriminator of every DCGAN-block clone (lsgan.py:45,72, sgan.py:46,76,
    infogan.py:58,88, relativistic_gan.py:37,65, cogan.py:51,90, began.py:47,75, ebgan.py:47,74) reproduce the outputs, input
    gradients and parameter-gradient digests recorded from the reference's OWN classes (same seeds, inputs, Dropout2d masks)."""
    import contextlib
    import warnings

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
            ctx = M.feed_masks(masks=masks) if masks else contextlib.nullcontext()
            outs, gins = _clone_fwd_bwd(net, ins, ctx, in_grad=(tag == "d"))
            assert len(outs) == int(gold[tag + "_nout"])
            for i, o in enumerate(outs):
                # (bit-equal in the pinning run; other thread counts / hosts re-order the CPU reductions)
                assert torch.allclose(o, torch.from_numpy(gold["%s_out%d" % (tag, i)]), rtol=1e-5, atol=1e-6), (name, tag, i)
            if tag == "d":
                for i, gi in enumerate(gins):
                    ref_g = torch.from_numpy(gold["d_in_grad%d" % i])
                    assert float((gi - ref_g).norm()) <= 1e-4 * float(ref_g.norm()) + 1e-9, (name, "input grad", i)
            grads = dict(net.named_parameters())
            # a conv bias in front of a BatchNorm has an exactly-zero true gradient: its recorded digest is rounding noise
            # (1e-7 against 1e+1 for the weights), so the absolute floor scales with the largest digest of the network
            scale = np.abs(gold[tag + "_digest"]).max(axis=0)
            for k, gd in zip([str(k) for k in gold[tag + "_keys"]], gold[tag + "_digest"]):
                mine = digest(grads[k].grad)
                assert abs(mine[1] - gd[1]) <= 1e-4 * abs(gd[1]) + 1e-6 * scale[1], (name, k)
                assert abs(mine[2] - gd[2]) <= 1e-4 * abs(gd[2]) + 1e-9 * scale[2], (name, k)


@pytest.mark.parametrize("name,kw", [("relativistic_gan", {}), ("ebgan", {}), ("lsgan", {}),
                                     ("relativistic_gan_avg", {"rel_avg_gan": True})])
def test_clone_loops_against_reference_fixture(golden_dir, name, kw):
    """relativistic_gan.py:126-182 (incl. its overwritten generator loss and the --rel_avg_gan branch), ebgan.py:142-202 and
    lsgan.py:140-180 restated: the traces recorded with the reference's own classes inside the loop are reproduced."""
    gold = load_golden(golden_dir, "clone_%s_32_loop" % name)
    base = name.replace("_avg", "")
    _seed(0)
    s = S.make_clone(base)
    step = getattr(S, base + "_step")
    n = int(gold["masks_per_step"])
    for t in range(len(gold["trace"])):
        masks = [gold["mask_%d_%02d" % (t, i)] for i in range(n)]
        with M.feed_masks(masks=masks):
            o = step(s, torch.from_numpy(gold["imgs"][t]), torch.from_numpy(gold["zs"][t]), **kw)
        # step 0 is a pure forward of pinned weights; later steps carry the Adam trajectory of a CPU run with another thread
        # count (Adam's first steps are sign-like: measured 1e-5 on ebgan's d_loss at step 1)
        tol = 1e-6 if t == 0 else 1e-4
        for j, k in enumerate(("g_loss", "d_loss")):
            assert abs(float(o[k]) - gold["trace"][t][j]) <= tol * max(1.0, abs(gold["trace"][t][j])), (t, k)
