"""The training-loop bodies on the HIP path: one function per reference script.

Each `*_step` performs exactly the work of one iteration of the reference loop it cites — forwards, losses,
backward, Adam — on device tensors, with every layer/loss/optimiser op running in libmigan.so.  Host RNG
draws (z, alpha, replay-buffer picks) stay on the host as in the reference and are passed in / drawn with
`np.random` / `random`, so a seeded run consumes the same random numbers as the reference.

`skip_dead_grads=True` (default) does not compute gradients that the reference computes and then provably
discards: the discriminator weight gradients during the generator step (zeroed by `optimizer_D.zero_grad()`
at dcgan.py:175, cyclegan.py:211,228, srgan.py:135, pix2pix.py:158), the generator gradients inside the
WGAN-GP critic step (wgan_gp.py:167,173 then zeroed at :176) and VGG weight gradients (srgan.py:128, never
read).  Results are identical either way (tests run both).
"""
import contextlib
import itertools
import os
import random
from types import SimpleNamespace

import numpy as np
import torch

from . import functional as F
from . import nn as gnn
from .dp import LocalStepper
from .optim import Adam

ADAM = dict(lr=2e-4, betas=(0.5, 0.999))


def _dev(a, device):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32).to(device)


@contextlib.contextmanager
def frozen(*modules, enabled=True):
    """Temporarily mark parameters as not requiring grad (skips their wgrad kernels)."""
    ps = [p for m in modules for p in m.parameters() if p.requires_grad] if enabled else []
    for p in ps:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in ps:
            p.requires_grad_(True)


def _scoped(fn):
    """Run a step body inside functional.weight_cache_scope(): packed weights are re-used within ONE step only."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **kw):
        # the plans that batch a step's weight packs / dropout masks into one launch belong to the step STATE (first argument)
        with F.weight_cache_scope(owner=a[0] if a else None):
            try:
                return fn(*a, **kw)
            finally:
                F.join_wgrad_streams()   # no weight-gradient launch outlives the step body that issued it (normally joined by dp.step())

    return wrapper


_PAIR_D = os.environ.get("MIGAN_PAIR_D", "1") == "1"  # A/B knob: run D(real), D(fake) as one batch (dcgan_step)


def _sync_bn(s):
    return getattr(s.dp, "sync_bn", None) is not None


_ONES = {}


def _root_grad(loss):
    """The cached ones tensor `_backward` seeds loss.backward() with, or None while a capture is in progress and none is cached yet."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    one = _ONES.get(key)
    if one is None:
        if loss.is_cuda and torch.cuda.is_current_stream_capturing():   # memory made during a capture belongs to the graph's pool: do not keep it
            return None
        one = _ONES[key] = torch.ones(loss.shape, device=loss.device, dtype=loss.dtype)
    return one


def _backward(loss):
    """loss.backward() with the root gradient from a cached ones scalar: autograd otherwise makes it with ones_like - one more fill
    launch (and graph node) per backward call, two to three per training step."""
    one = _root_grad(loss)
    if one is None:
        loss.backward()
    else:
        loss.backward(one)


def half_sum(a, b):
    """(a + b) / 2 as the reference writes it (bit-identical: scaling by 0.5 is exact)."""
    return F.axpby(a, b, 0.5, 0.5)


# ------------------------------------------------------------------------------------------------ dcgan / gan
def make_gan_state(G, D, latent_dim=100, skip_dead_grads=True, dp=None):
    """G, D: swapped modules already on the GPU (dcgan.py:106-116 / gan.py:87-93)."""
    return SimpleNamespace(G=G, D=D, opt_G=Adam(G.parameters(), **ADAM), opt_D=Adam(D.parameters(), **ADAM),
                           bce=gnn.BCELoss(), latent_dim=latent_dim, skip=skip_dead_grads, labels={},
                           dp=dp or LocalStepper())


def _labels(s, shape, device):
    key = (tuple(shape), str(device))
    if key not in s.labels:
        s.labels[key] = (torch.ones(shape, device=device), torch.zeros(shape, device=device))
    return s.labels[key]


# The discriminator update of dcgan.py:170-183 needs from the generator update only `gen_imgs.detach()` (made by the forward) and the
# discriminator as the generator's loss left it (its weights untouched - it is frozen there - and its BatchNorm running statistics after
# the D(gen) forward): nothing of the generator's BACKWARD.  With one process per GPU and no collective in the step, its forward and
# backward - ~65 launches of 5-30 us on 2 x 128 images of at most 32 x 32, which cannot fill the chip - run on a second HIP stream
# UNDERNEATH the generator's backward (six MFMA launches of 150-330 us and the large BatchNorm passes); optimizer_D.step() follows the
# join, so nothing reads a weight that is being updated.  Same kernels in the same order per stream: results are bit-identical to the
# sequential order (tests flip `_OVERLAP_D`).  Inside a captured step the fork / join are edges of the one hipGraph.
_OVERLAP_D = True
_D_STREAMS = {}


def _d_stream(device):
    main = torch.cuda.current_stream(device)
    key = (device.index, main.cuda_stream)
    if key not in _D_STREAMS:
        # (default priority: a high-priority second stream was measured - the captured DCGAN step 2.91 -> 4.33 ms, SRGAN +0.5 %,
        # CycleGAN -0.5 %, profiles/r04_ab.txt call 16)
        _D_STREAMS[key] = torch.cuda.Stream(device)
        F.ensure_splitk_ws(device, _D_STREAMS[key])
    return main, _D_STREAMS[key]


_VREAL_SIDE = True   # srgan_step: VGG features of the real images beside the generator's forward (83.10 -> 82.77 ms, profiles/r04_ab.txt call 29)
_CHAINS = True      # cyclegan_step: the two halves of the generators' forward (and backward) on two streams; tests / bench --no-overlap flip it
_C_STREAMS = {}


def _c_stream(device):
    """(current stream, the stream of cyclegan_step's second forward chain) - not the discriminator halves' stream: the chain's
    backward nodes run on it during the generators' backward, beside the discriminator updates."""
    main = torch.cuda.current_stream(device)
    key = (device.index, main.cuda_stream)
    if key not in _C_STREAMS:
        _C_STREAMS[key] = torch.cuda.Stream(device)
        F.ensure_splitk_ws(device, _C_STREAMS[key])
    return main, _C_STREAMS[key]


def _conv_params_only(s):
    """True when every parameter of the CycleGAN generators belongs to a conv layer (cyclegan/models.py:22-88: InstanceNorm2d without
    affine parameters): their gradients all come from weight-gradient launches, which one_wgrad_stream() serialises.  A norm layer with
    affine parameters would add its gradients on the backward node's own stream."""
    ok = s.__dict__.get("_chains_ok")
    if ok is None:
        ok = all(isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) or not list(m.parameters(recurse=False))
                 for g in (s.G_AB, s.G_BA) for m in g.modules())
        s._chains_ok = ok
    return ok


def _two_streams_ok(s, ref):
    # skip_dead_grads: otherwise the generator's backward also writes (dead) discriminator gradients, which the other stream zeroes;
    # cross-replica BatchNorm puts collectives inside forward / backward: one stream
    return _OVERLAP_D and s.skip and ref.is_cuda and not _sync_bn(s)


# Where the discriminator update runs when there is more than one rank (SURVEY.md 8e "Overlap").
#   "sequential" (default for world > 1): the reference's order - generator backward, dp.step(opt_G), discriminator update,
#       dp.step(opt_D).  The generator bucket's all-reduce + Adam leave for the data-parallel side stream right after the generator's
#       backward and run UNDER the discriminator phase (which reads gen.detach() only: dcgan.py:179, cyclegan.py:216,233, srgan.py:139).
#   "fork": the single-GPU body - the discriminator update on a second stream underneath the generator's backward, every dp.step()
#       behind the join.  Faster compute (the D phase is hidden), but the generator bucket's exchange is then exposed at the end of the step.
# Which one wins at N = 8 depends on exchange time vs the D phase's length and needs the 8-GPU node to decide: bench.py --dp-order.
# world == 1 always takes the fork (there is no exchange to hide).
_DP_ORDER = os.environ.get("MIGAN_DP_ORDER", "sequential")


def set_dp_order(order):
    global _DP_ORDER
    if order not in ("sequential", "fork"):
        raise ValueError("dp order: 'sequential' or 'fork'")
    _DP_ORDER = order


def _d_under_g_ok(s, ref):
    return _two_streams_ok(s, ref) and (getattr(s.dp, "world", 1) == 1 or _DP_ORDER == "fork")


def _fork_join(g_loss, d_half):
    """g_loss.backward() on the current stream with d_half() - forward + backward of the discriminator update(s) - on the second stream
    underneath it; returns d_half()'s value after the join.  The optimiser steps are the caller's, AFTER the join: with data
    parallelism they close a hipGraph segment (graph.StepRunner cuts the captured step at every dp.step()), which must not happen
    while a stream is forked, and no weight is updated while the other stream may still read it."""
    main, side = _d_stream(g_loss.device)
    _root_grad(g_loss)       # both backward calls are seeded from this tensor: it exists before the fork
    side.wait_stream(main)   # fork: everything up to the generator's loss
    try:
        with F.two_streams():
            with torch.cuda.stream(side):
                out = d_half()
            _backward(g_loss)
    finally:
        main.wait_stream(side)   # join - also when a half raised: no stream stays forked (a capture in progress could not end)
    return out


@_scoped
def dcgan_step(s, real_imgs, z):
    """dcgan.py:143-183 (and gan.py:121-161)."""
    valid, fake = _labels(s, (real_imgs.shape[0], 1), real_imgs.device)
    s.dp.begin_step()
    s.opt_G.zero_grad()
    gen = s.G(z)
    with frozen(s.D, enabled=s.skip):
        g_loss = s.bce(s.D(gen), valid)
    if _d_under_g_ok(s, real_imgs):
        d_loss = _fork_join(g_loss, lambda: _dcgan_d_half(s, real_imgs, gen, valid, fake))
        s.dp.step(s.opt_G)
    else:
        _backward(g_loss)
        s.dp.step(s.opt_G)
        d_loss = _dcgan_d_half(s, real_imgs, gen, valid, fake)
    s.dp.step(s.opt_D)
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gen_imgs": gen.detach()}


def _dcgan_d_half(s, real_imgs, gen, valid, fake):
    """dcgan.py:170-181: optimizer_D.zero_grad() ... d_loss.backward() (the optimiser step is the caller's)."""
    s.opt_D.zero_grad()
    if _PAIR_D and real_imgs.shape == gen.shape and not _sync_bn(s):
        # D(real) and D(fake) of dcgan.py:176-177 as ONE pass over cat(real, fake): per-half BatchNorm statistics (and
        # running-statistics updates in the reference's order), one launch per layer instead of two in forward and backward;
        # mean BCE over the 2n rows against (valid | fake) == (real_loss + fake_loss) / 2
        n = real_imgs.shape[0]
        both = torch.empty((2 * n,) + tuple(real_imgs.shape[1:]), device=real_imgs.device, dtype=torch.float32,
                           memory_format=torch.channels_last)
        F.copy_into(both[:n], real_imgs)
        F.copy_into(both[n:], gen.detach())
        key = ("pair", n, str(real_imgs.device))
        if key not in s.labels:
            s.labels[key] = torch.cat([valid, fake])
        with gnn.paired_batches(s.D):
            d_loss = s.bce(s.D(both), s.labels[key])
    else:
        real_loss = s.bce(s.D(real_imgs), valid)
        fake_loss = s.bce(s.D(gen.detach()), fake)
        d_loss = half_sum(real_loss, fake_loss)
    _backward(d_loss)
    return d_loss


gan_step = dcgan_step


# ------------------------------------------------------------------------------------------------ wgan_gp
def compute_gradient_penalty(D, real_samples, fake_samples, alpha=None):
    """wgan_gp.py:119-138 - and the same function of stargan.py:142-161 / dualgan.py:116-135 on their conv critics - on the HIP
    path: interpolation, D forward, differentiable backward (create_graph=True through the Linear / Conv2d / LeakyReLU /
    BatchNorm Functions), per-sample L2 norm and mean((n-1)^2).  A critic that returns a tuple (stargan's (out_adv, out_cls),
    stargan/models.py:110-115) is differentiated through its first output, as the reference does."""
    B = real_samples.size(0)
    dev = real_samples.device
    if alpha is None:
        alpha = _dev(np.random.random((B, 1, 1, 1)), dev)
    a = alpha.reshape(B)
    if real_samples.dim() == 4:  # one memory layout for both operands of the row-scaled sum
        real_samples, fake_samples = F.canon(real_samples), F.canon(fake_samples)
    mix = F.axpby(F.rowscale(real_samples, a), F.rowscale(fake_samples, 1 - a), 1.0, 1.0)
    mix = mix.view(real_samples.shape).requires_grad_(True)
    d_mix = D(mix)
    if isinstance(d_mix, tuple):
        d_mix = d_mix[0]
    ones = torch.ones(d_mix.shape, device=dev)
    conv_critic = d_mix.dim() == 4
    with F.input_grad_only() if conv_critic else contextlib.nullcontext():
        grads = torch.autograd.grad(outputs=d_mix, inputs=mix, grad_outputs=ones, create_graph=True, retain_graph=True,
                                    only_inputs=True)[0]
    if grads.dim() == 4 and not grads.is_contiguous():
        # the per-sample norm runs over all of C*H*W: any order of the sample's elements will do, so the NHWC storage is
        # viewed as rows directly (no re-layout launch inside the differentiated graph)
        rows = grads.permute(0, 2, 3, 1).reshape(B, -1)
    else:
        rows = grads.view(B, -1)
    norms = F.rownorm(rows)
    return F.loss(F.LOSS_MSE, norms, None, 1.0)


def make_wgan_gp_state(G, D, latent_dim=100, skip_dead_grads=True, dp=None):
    return SimpleNamespace(G=G, D=D, opt_G=Adam(G.parameters(), **ADAM), opt_D=Adam(D.parameters(), **ADAM),
                           latent_dim=latent_dim, lambda_gp=10.0, n_critic=5, skip=skip_dead_grads,
                           dp=dp or LocalStepper())


# K7: the critic half of the iteration (D(real), D(fake), gradient penalty with its double backward, d_loss.backward()) on the
# fused kernels of csrc/critic_fused.hip, the generator's no_grad forward and the generator iteration on csrc/mlp_fused.hip.
# MIGAN_K7=0 keeps the op-by-op path (autograd over the skinny GEMM Functions): the A/B of bench.py and the parity tests.
_K7 = os.environ.get("MIGAN_K7", "1") == "1"


def _k7():
    return _K7


class _CriticFusedPlan:
    """The critic of wgan_gp.py:68-83 - Sequential(Linear, LeakyReLU, Linear, LeakyReLU, Linear(., 1)) behind a flatten - as the
    operands of migan_critic_fused; `ok = False` when the model, the batch or the optimiser layout is anything else."""

    def __init__(self, s, real):
        from ._lib import lib

        self.ok = False
        seq = getattr(s.D, "model", None)
        mods = list(seq) if isinstance(seq, torch.nn.Sequential) else []
        if len(mods) != 5 or not all(isinstance(mods[k], torch.nn.Linear) for k in (0, 2, 4)) \
                or not all(isinstance(mods[k], torch.nn.LeakyReLU) for k in (1, 3)):
            return
        l1, a1, l2, a2, l3 = mods
        if a1.negative_slope != a2.negative_slope or l3.out_features != 1 or any(l.bias is None for l in (l1, l2, l3)) \
                or l2.in_features != l1.out_features or l3.in_features != l2.out_features:
            return
        B = real.shape[0]
        if real[0].numel() != l1.in_features or not lib.migan_critic_fused_ok(B, l1.in_features, l1.out_features, l2.out_features):
            return
        self.B, self.dims, self.slope = B, (l1.in_features, l1.out_features, l2.out_features), float(a1.negative_slope)
        self.params = [l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias]
        if set(map(id, self.params)) != set(map(id, s.opt_D.params)):
            return
        dev = real.device
        self.lambda_gp = float(s.lambda_gp)
        self.ws_bytes = lib.migan_critic_fused_workspace(B, *self.dims)
        self.ws = torch.empty(self.ws_bytes // 4, device=dev, dtype=torch.float32)
        self.ok = True

    def usable(self, real, fake):
        return self.ok and real.shape[0] == self.B and real.is_contiguous() and fake.is_contiguous() \
            and real.dtype == fake.dtype == torch.float32

    def run(self, real, fake, alpha, grads):
        """Six launches: the gradient of d_loss is WRITTEN into `grads` (six contiguous tensors); -> (d_loss, gp)."""
        from ._lib import check, lib

        a = alpha.reshape(self.B).contiguous()
        w = [p.detach() for p in self.params]
        o = torch.empty(4, device=real.device, dtype=torch.float32)   # a fresh slot per call: callers keep their losses
        check(lib.migan_critic_fused(real.data_ptr(), fake.data_ptr(), a.data_ptr(), *[t.data_ptr() for t in w],
                                     *[g.data_ptr() for g in grads], o.data_ptr(), self.ws.data_ptr(), self.ws_bytes,
                                     self.B, *self.dims, self.slope, self.lambda_gp, 0, 0,
                                     torch.cuda.current_stream().cuda_stream), "critic_fused")
        return o[0], o[1]


class _GeneratorFusedPlan:
    """The MLP generator of wgan_gp.py:42-65 - Sequential of Linear [-> BatchNorm1d] [-> LeakyReLU | Tanh] groups behind a view to
    img_shape - as the operands of migan_mlp_fused_fwd (four launches for the no_grad forward of a critic iteration: one per layer, the
    first layer inside the launch of the second)."""

    def __init__(self, G, z, out_shape=None):
        import ctypes

        from ._lib import lib

        self.ok = False
        seq, shape = getattr(G, "model", None), (out_shape if out_shape is not None else getattr(G, "img_shape", None))
        if not isinstance(seq, torch.nn.Sequential) or shape is None or z.dim() != 2:
            return
        mods, groups, i = list(seq), [], 0
        while i < len(mods):
            if not isinstance(mods[i], torch.nn.Linear):
                return
            lin, bn, act, slope = mods[i], None, F.ACT_NONE, 0.0
            i += 1
            if i < len(mods) and isinstance(mods[i], torch.nn.BatchNorm1d):
                bn = mods[i]
                i += 1
                if bn.momentum is None or not bn.track_running_stats:   # cumulative averaging / no buffers: not on the reference path
                    return
            if i < len(mods) and isinstance(mods[i], torch.nn.LeakyReLU):
                act, slope = F.ACT_LRELU, float(mods[i].negative_slope)
                i += 1
            elif i < len(mods) and isinstance(mods[i], torch.nn.Tanh):
                act = F.ACT_TANH
                i += 1
            groups.append((lin, bn, act, slope))
        B, n = z.shape[0], len(groups)
        if not groups or groups[0][0].in_features != z.shape[1] or int(np.prod(shape)) != groups[-1][0].out_features:
            return
        self.dims = (ctypes.c_int * (4 * n))(*[v for (l, bn, a, _) in groups for v in (l.in_features, l.out_features, int(bn is not None), a)])
        self.fpar = (ctypes.c_float * (3 * n))(*[v for (_, bn, _, sl) in groups for v in (sl, bn.eps if bn else 0.0, bn.momentum if bn else 0.0)])
        if not lib.migan_mlp_fused_ok(B, n, self.dims):
            return
        self.B, self.n, self.groups, self.shape, self.G = B, n, groups, tuple(shape), G
        self.ws_bytes = lib.migan_mlp_fused_workspace(B, n, self.dims, 0)
        self.ws = torch.empty(self.ws_bytes // 4, device=z.device, dtype=torch.float32)
        self.tickets = torch.zeros(1024, device=z.device, dtype=torch.int32)   # zero at rest: the BatchNorm1d column tiles' arrival counters
        self.save = self.bws = None   # buffers of the differentiated form, made on first use
        self.ok = True

    def _train_buffers(self, dev):
        from ._lib import lib

        if self.save is None:
            self.save_bytes = lib.migan_mlp_fused_workspace(self.B, self.n, self.dims, 1)
            self.save = torch.empty(self.save_bytes // 4, device=dev, dtype=torch.float32)
            self.bws_bytes = lib.migan_mlp_fused_bwd_workspace(self.B, self.n, self.dims)
            self.bws = torch.empty(self.bws_bytes // 4, device=dev, dtype=torch.float32)

    def forward_saved(self, x, buffers=None):
        """Forward that keeps what backward() needs (one launch per layer); BatchNorm side effects as in run()."""
        import ctypes

        from ._lib import check, lib

        self._train_buffers(x.device)
        ts = self.tensors(buffers)
        ptrs = (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
        y = torch.empty(self.B, self.groups[-1][0].out_features, device=x.device, dtype=torch.float32)
        check(lib.migan_mlp_fused_fwd(x.data_ptr(), y.data_ptr(), self.B, self.n, self.dims, self.fpar, ptrs, self.save.data_ptr(),
                                      self.save_bytes, 1, self.tickets.data_ptr(), 0, torch.cuda.current_stream().cuda_stream), "mlp_fused_fwd")
        return y

    def backward(self, x, y, dy, grads=None, want_dx=False, accumulate=False):
        """One launch per phase: parameter gradients written into (accumulate: added to) `grads` (per group [dW, db, dgamma, dbeta],
        None = not wanted) and / or the input gradient (returned) of the forward_saved() call that produced y."""
        import ctypes

        from ._lib import check, lib

        ts = self.tensors()
        ptrs = (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
        flat = [g for grp in (grads or [[None] * 4] * self.n) for g in grp]
        gptrs = (ctypes.c_void_p * len(flat))(*[None if g is None else g.data_ptr() for g in flat])
        dx = torch.empty_like(x) if want_dx else None
        check(lib.migan_mlp_fused_bwd(x.data_ptr(), y.data_ptr(), dy.data_ptr(), self.save.data_ptr(), None if dx is None else dx.data_ptr(),
                                      self.B, self.n, self.dims, self.fpar, ptrs, gptrs, self.bws.data_ptr(), self.bws_bytes,
                                      int(bool(accumulate)), 0, torch.cuda.current_stream().cuda_stream), "mlp_fused_bwd")
        return dx

    def param_grads(self):
        """The parameters' .grad tensors in the order backward() wants them, or None if one is missing / not contiguous."""
        out = []
        for lin, bn, _, _ in self.groups:
            grp = [lin.weight.grad, lin.bias.grad] + ([bn.weight.grad, bn.bias.grad] if bn is not None else [None, None])
            if any(g is None or not g.is_contiguous() for g in grp[:2 if bn is None else 4]):
                return None
            out.append(grp)
        return out

    def tensors(self, buffers=None):
        """Device tensors in the order of the C entry's pointer table; `buffers` replaces the BatchNorm buffers (verification)."""
        out, k = [], 0
        for lin, bn, _, _ in self.groups:
            out += [lin.weight, lin.bias]
            if bn is None:
                out += [None] * 5
            else:
                bufs = buffers[k:k + 3] if buffers is not None else [bn.running_mean, bn.running_var, bn.num_batches_tracked]
                k += 3
                out += [bn.weight, bn.bias] + list(bufs)
        return out

    def usable(self, z):
        # the fused kernels take BatchNorm1d statistics over the rows they are given: with cross-replica BatchNorm on
        # (dp.enable_sync_batchnorm, world > 1) a generator with BatchNorm layers goes through the modules, whose norm calls gather
        if F._SYNC_BN is not None and F._SYNC_BN.world > 1 and self.ok and any(bn is not None for _, bn, _, _ in self.groups):
            return False
        return self.ok and self.G.training and z.shape[0] == self.B and z.is_contiguous() and z.dtype == torch.float32

    def run(self, z, buffers=None):
        import ctypes

        from ._lib import check, lib

        ts = self.tensors(buffers)
        ptrs = (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
        y = torch.empty(self.B, self.groups[-1][0].out_features, device=z.device, dtype=torch.float32)
        check(lib.migan_mlp_fused_fwd(z.data_ptr(), y.data_ptr(), self.B, self.n, self.dims, self.fpar, ptrs, self.ws.data_ptr(),
                                      self.ws_bytes, 0, self.tickets.data_ptr(), 0, torch.cuda.current_stream().cuda_stream), "mlp_fused_fwd")
        return y.view(self.B, *self.shape)


def _generator_nograd(s, z):
    """fake_imgs = generator(z) without a graph (wgan_gp.py:163 when its gradients are dead): the fused forward when the
    generator is the MLP of wgan_gp.py:42-65, else the modules."""
    if _k7():
        plan = getattr(s, "_k7_gen_plan", None)
        if plan is None or (plan.ok and plan.B != z.shape[0]):
            if _capturing(z):   # see _capturing: no plan is built inside a recording
                return s.G(z)
            plan = s._k7_gen_plan = _GeneratorFusedPlan(s.G, z)
        if plan.usable(z):
            return plan.run(z)
    return s.G(z)


def _capturing(t):
    """A fused-kernel plan owns buffers that must be zero AT REST (ticket counters) and that every later step - eager or another
    recorded graph - reuses: built inside a capture they would live in that graph's private pool and be zeroed by a captured memset
    only.  While a recording is in progress no plan is built; the step takes the module path, whose result is the same."""
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


def _generator_iteration_plans(s, z):
    """(generator plan, critic-as-MLP plan) for the fused generator iteration, or None"""
    gp = getattr(s, "_k7_gen_plan", None)
    if gp is None or (gp.ok and gp.B != z.shape[0]):
        if _capturing(z):
            return None
        gp = s._k7_gen_plan = _GeneratorFusedPlan(s.G, z)
    if not gp.usable(z):
        return None
    dpn = getattr(s, "_k7_dmlp_plan", None)
    nin = gp.groups[-1][0].out_features
    if dpn is None or (dpn.ok and dpn.B != z.shape[0]):
        if _capturing(z):
            return None
        probe = torch.empty(z.shape[0], nin, device=z.device, dtype=torch.float32)
        dpn = s._k7_dmlp_plan = _GeneratorFusedPlan(s.D, probe, out_shape=(1,))
        if dpn.ok and (dpn.groups[-1][0].out_features != 1 or any(bn is not None for _, bn, _, _ in dpn.groups) or nin % 32 != 0):
            dpn.ok = False
    if not dpn.ok:
        return None
    if gp.save is None or dpn.save is None or getattr(dpn, "_dval", None) is None:   # buffers of the differentiated form: made outside a capture
        if _capturing(z):
            return None
        gp._train_buffers(z.device)
        dpn._train_buffers(z.device)
        dpn._dval = torch.full((gp.B, 1), -1.0 / gp.B, device=z.device, dtype=torch.float32)
    if not hasattr(gp, "_covers_opt"):   # the fused backward WRITES gradients: it must own every parameter of the generator's optimiser
        mine = {id(t) for lin, bn, _, _ in gp.groups for t in ((lin.weight, lin.bias) + ((bn.weight, bn.bias) if bn is not None else ()))
                if t is not None}
        gp._covers_opt = mine == {id(q) for q in s.opt_G.params}
    return (gp, dpn) if gp._covers_opt else None


def _fused_generator_pass(gp, dpn, z, buffers, grads):
    """generator(z) -> frozen critic -> g_loss = -mean(validity) -> gradients of the generator's parameters WRITTEN into `grads`:
    two forwards that keep their activations, two backwards (one launch per layer / phase) + the mean."""
    B = gp.B
    fake = gp.forward_saved(z, buffers)
    val = dpn.forward_saved(fake)
    g_loss = F.axpby(F.mean(val), None, -1.0, 0.0)
    if getattr(dpn, "_dval", None) is None:
        dpn._dval = torch.full((B, 1), -1.0 / B, device=z.device, dtype=torch.float32)
    dfake = dpn.backward(fake, val, dpn._dval, None, want_dx=True)
    gp.backward(z, fake, dfake, grads, want_dx=False)
    return g_loss


def _generator_iteration_fused(s, z):
    """wgan_gp.py:179-193 (fake_imgs = generator(z); g_loss = -mean(discriminator(fake_imgs)); g_loss.backward()) when both
    networks are the MLPs of wgan_gp.py:42-83; else None."""
    plans = _generator_iteration_plans(s, z)
    if plans is None:
        return None
    grads = plans[0].param_grads()
    if grads is None:
        return None
    return _fused_generator_pass(plans[0], plans[1], z, None, grads)


def _critic_plan(s, real, fake):
    plan = getattr(s, "_k7_plan", None)
    if plan is None or (plan.ok and plan.B != real.shape[0]):
        if _capturing(real):
            return None
        plan = s._k7_plan = _CriticFusedPlan(s, real)
    return plan if plan.usable(real, fake) else None


@_scoped
def wgan_gp_step(s, real_imgs, i, z, alpha=None):
    """wgan_gp.py:146-193: critic iteration i, generator update when i % n_critic == 0."""
    s.dp.begin_step()
    if s.skip:
        with torch.no_grad():  # G grads from d_loss are discarded at wgan_gp.py:176
            fake_imgs = _generator_nograd(s, z)
    else:
        fake_imgs = s.G(z)
    plan = _critic_plan(s, real_imgs, fake_imgs) if (_k7() and s.skip) else None
    if plan is not None and alpha is None:  # the host draw of wgan_gp.py:122, where the reference makes it
        alpha = _dev(np.random.random((real_imgs.shape[0], 1, 1, 1)), real_imgs.device)
    if plan is not None:
        s.opt_D.attach_grads()   # optimizer_D.zero_grad() of wgan_gp.py:157 without the fill: the fused kernels WRITE every gradient
    else:
        s.opt_D.zero_grad()
    grads = [p.grad for p in plan.params] if plan is not None else []
    if plan is not None and all(g is not None and g.is_contiguous() for g in grads):
        d_loss, gp = plan.run(real_imgs, fake_imgs, alpha, grads)
    else:
        if plan is not None:
            s.opt_D.zero_grad()
        real_v = s.D(real_imgs)
        fake_v = s.D(fake_imgs)
        gp = compute_gradient_penalty(s.D, real_imgs.data, fake_imgs.data, alpha)
        # d_loss = -mean(real) + mean(fake) + lambda_gp * gp
        d_loss = F.axpby(F.axpby(F.mean(fake_v), F.mean(real_v), 1.0, -1.0), gp, 1.0, s.lambda_gp)
        _backward(d_loss)
    s.dp.step(s.opt_D)
    out = {"d_loss": d_loss.detach(), "gp": gp.detach()}
    gen_fused = _k7() and s.skip and _generator_iteration_plans(s, z) is not None
    if gen_fused:
        s.opt_G.attach_grads()   # optimizer_G.zero_grad() of wgan_gp.py:178: the fused generator backward writes every gradient
    else:
        s.opt_G.zero_grad()
    if i % s.n_critic == 0:
        s.dp.wait(s.opt_D)  # the generator step reads the critic that was just updated (wgan_gp.py:186)
        g_loss = _generator_iteration_fused(s, z) if gen_fused else None
        if g_loss is None:
            if gen_fused:
                s.opt_G.zero_grad()
            fake_imgs = s.G(z)
            with frozen(s.D, enabled=s.skip):
                g_loss = F.axpby(F.mean(s.D(fake_imgs)), None, -1.0, 0.0)
            _backward(g_loss)
        s.dp.step(s.opt_G)
        out["g_loss"] = g_loss.detach()
    return out


# ------------------------------------------------------------------------------------------------ dragan (8f F1)
def compute_gradient_penalty_dragan(D, X, alpha=None, noise=None, lambda_gp=10.0):
    """dragan.py:144-167 on the HIP path: the conv-critic gradient penalty.  `autograd.grad(..., create_graph=True)` runs
    the differentiable backward of Conv2d / LeakyReLU / Dropout2d / BatchNorm2d / Linear / Sigmoid (functional.py), the
    gradient norm is taken over the channel dimension as the reference writes it (`norm(2, dim=1)` of a (B, C, H, W)
    tensor), and the interpolation uses the unbiased std of all of X without a host sync.
    Host draws when not given: alpha ~ np.random.random(X.shape), then noise ~ torch.rand(X.size()) (reference order)."""
    X = F.canon(X)
    dev = X.device
    B, C, H, W = X.shape
    if alpha is None:
        alpha = _dev(np.random.random(size=tuple(X.shape)), dev)
    if noise is None:
        noise = torch.rand(X.size()).to(dev)
    interp = F.dragan_interpolate(X, alpha, noise).requires_grad_(True)
    with F.input_grad_only():
        d_interp = D(interp)
        ones = torch.ones(B, 1, device=dev)
        grads = torch.autograd.grad(outputs=d_interp, inputs=interp, grad_outputs=ones, create_graph=True,
                                    retain_graph=True, only_inputs=True)[0]
    # norm over dim 1 (channels): in NHWC memory the channel vector of a pixel is contiguous -> rows of a (B*H*W, C) view
    norms = F.rownorm(F.relayout(grads, True).permute(0, 2, 3, 1).reshape(B * H * W, C))
    return F.axpby(F.loss(F.LOSS_MSE, norms, None, 1.0), None, float(lambda_gp), 0.0)


@_scoped
def dragan_step(s, real_imgs, z, alpha=None, noise=None):
    """dragan.py:176-217 (state from make_gan_state; s.lambda_gp defaults to 10).  Quirk kept: d_loss is computed - its
    discriminator forwards update BatchNorm running statistics and consume Dropout2d draws - but only
    gradient_penalty.backward() feeds optimizer_D.step()."""
    valid, fake = _labels(s, (real_imgs.shape[0], 1), real_imgs.device)
    s.dp.begin_step()
    s.opt_G.zero_grad()
    gen = s.G(z)
    with frozen(s.D, enabled=s.skip):
        g_loss = s.bce(s.D(gen), valid)
    _backward(g_loss)
    s.dp.step(s.opt_G)
    s.opt_D.zero_grad()
    with torch.no_grad() if s.skip else contextlib.nullcontext():  # d_loss is never back-propagated (dragan.py:211-217)
        real_loss = s.bce(s.D(real_imgs), valid)
        fake_loss = s.bce(s.D(gen.detach()), fake)
        d_loss = half_sum(real_loss, fake_loss)
    gp = compute_gradient_penalty_dragan(s.D, real_imgs.data, alpha, noise, getattr(s, "lambda_gp", 10.0))
    _backward(gp)
    s.dp.step(s.opt_D)
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gp": gp.detach(), "gen_imgs": gen.detach()}


class WganGpRunner:
    """`wgan_gp_step` replayed as hipGraphs.  The loop body has two shapes — critic only, and critic + generator when
    `i % n_critic == 0` (wgan_gp.py:179) — and a capture freezes host control flow, so both shapes are captured once
    over static input buffers and `run(i, ...)` replays the one iteration i needs.  The step is ~100 dependent small
    launches (Linear / LeakyReLU / double-backward chain at batch 64): replay removes the host launch cost.
    `prepare()` executes `2 * warmup` real iterations (they update the networks like any other iteration)."""

    def __init__(self, s, batch, img_shape, use_graph=True, warmup=2):
        from .graph import StepRunner

        dev = next(s.G.parameters()).device
        self.s = s
        self.real = torch.zeros(batch, *img_shape, device=dev)
        # z and alpha share ONE staging buffer, so a caller that keeps its draws packed [z | alpha] feeds an iteration with one copy
        self.inp = torch.zeros(batch * s.latent_dim + batch, device=dev)
        self.z = self.inp[:batch * s.latent_dim].view(batch, s.latent_dim)
        self.alpha = self.inp[batch * s.latent_dim:].view(batch, 1, 1, 1)
        self.runners = {
            True: StepRunner(lambda: wgan_gp_step(s, self.real, 0, self.z, self.alpha), s.dp, use_graph, warmup),
            False: StepRunner(lambda: wgan_gp_step(s, self.real, 1, self.z, self.alpha), s.dp, use_graph, warmup),
        }

    def prepare(self, real, z, alpha):
        self._load(real, z, alpha)
        for r in self.runners.values():
            r.prepare()
        return self

    @property
    def graphed(self):
        return all(r.graphed for r in self.runners.values())

    @property
    def capture_error(self):
        return next((r.capture_error for r in self.runners.values() if r.capture_error), None)

    def _load(self, real, z, alpha, packed=None):
        if real is not None:
            self.real.copy_(real)
        if packed is not None:
            self.inp.copy_(packed.reshape(self.inp.shape))
        else:
            self.z.copy_(z)
            self.alpha.copy_(alpha.reshape(self.alpha.shape))

    def run(self, i, real, z, alpha, packed=None):
        """Iteration i on (real, z, alpha); pass real=None to keep the batch already in the static buffer; `packed` = the flat
        [z | alpha] of this iteration (batch * latent + batch floats) instead of z and alpha."""
        self._load(real, z, alpha, packed)
        return self.runners[i % self.s.n_critic == 0].run()


# ------------------------------------------------------------------------------------------------ cyclegan
class ReplayBuffer:
    """cyclegan/utils.py:13-33 with a DEVICE-resident history: the index logic and the python `random`
    draws are the reference's, bit for bit; the samples live in one pool tensor [max_size][C][H][W] on the GPU and a call
    is two kernel launches - one gathers the returned batch from old pool entries and new samples, one writes the new
    samples into the pool - instead of per-sample clones and a torch.cat.  CPU tensors (the oracle comparison of the index
    logic) take the reference's list path.  The draws need nothing from the device, so a step recorded into a hipGraph keeps
    them on the host: plan(B) draws before each replay and refreshes a static device table the recorded launches read."""

    def __init__(self, max_size=50):
        if max_size <= 0:
            raise AssertionError("Empty buffer or trying to create a black hole. Be careful.")
        self.max_size, self.data = max_size, []
        self.pool, self.count = None, 0
        self.table, self._planned = None, None   # static pick table of the recorded step (plan()); batch size plan() drew for

    def __len__(self):
        return self.count if self.pool is not None else len(self.data)

    def _draw(self, B):
        """The reference loop (cyclegan/utils.py:17-32) on indices, for a batch of B new samples: -> (out_src, slot_src) where
        out_src[k] is the pool slot j >= 0 whose CURRENT content is returned as element k, or -1-k' for new sample k', and
        slot_src maps every pool slot written by this call to the new sample it holds afterwards (last write wins).  Consumes
        python `random` exactly as the reference does and advances the fill count."""
        out_src, slot_src = [], {}
        for k in range(B):
            if self.count < self.max_size:
                slot_src[self.count] = k
                self.count += 1
                out_src.append(-1 - k)
            elif random.uniform(0, 1) > 0.5:
                j = random.randint(0, self.max_size - 1)
                # the slot may already have been replaced earlier in this call: then its content is that new sample
                out_src.append(-1 - slot_src[j] if j in slot_src else j)
                slot_src[j] = k
            else:
                out_src.append(-1 - k)
        return out_src, slot_src

    def reserve(self, B):
        """Allocate the static pick table for B-sample batches (no draw): what a recording of the step needs to exist."""
        if self.pool is None:
            raise RuntimeError("ReplayBuffer.reserve(): no device pool yet (run one eager push_and_pop first)")
        if self.table is None or self.table.numel() != 3 * B:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ReplayBuffer: the static table must exist before a capture")
            self.table = torch.full((3 * B,), -1, dtype=torch.int32, device=self.pool.device)

    def plan(self, B):
        """Draw the picks of the NEXT push_and_pop of a B-sample batch now and put them into the buffer's static device table
        ([returned-batch sources | pool-update sources | pool-update slots, -1 = no update], 3*B int32).  A recorded hipGraph
        (CycleGanRunner) launches the two row-selection kernels over this table every replay: the host RNG draws - the
        reference's, in the reference's order - happen here, before the replay, and only the table's contents change."""
        self.reserve(B)
        self.table.copy_(torch.tensor(self.plan_rows(B), dtype=torch.int32))
        self._planned = B

    def plan_rows(self, B):
        """The 3*B table entries of plan() as a python list (draws the picks; the caller owns the copy to the device table and
        sets `_planned` - CycleGanRunner puts both buffers' rows into ONE pinned staging tensor and one asynchronous copy)."""
        out_src, slot_src = self._draw(B)
        slots = sorted(slot_src)
        pad = B - len(slots)
        return out_src + [-1 - slot_src[j] for j in slots] + [-1] * pad + slots + [-1] * pad

    def push_and_pop(self, batch):
        if not F.on_device(batch):
            return self._push_and_pop_host(batch)
        x = F.canon(batch.data)
        B = x.shape[0]
        D = x[0].numel()
        if self.pool is not None and self.pool.shape[1:] != x.shape[1:]:
            raise ValueError("ReplayBuffer: sample shape changed from %s to %s (the history cannot be mixed; use a new buffer)"
                             % (tuple(self.pool.shape[1:]), tuple(x.shape[1:])))
        if D % 4 != 0:
            raise ValueError("ReplayBuffer: C*H*W = %d must be a multiple of 4 on the device path (16-byte row copies)" % D)
        capturing = torch.cuda.is_current_stream_capturing()
        if self.pool is None:
            if capturing:
                raise RuntimeError("ReplayBuffer: the device pool must exist before the step is recorded into a hipGraph (run an "
                                   "eager step first - CycleGanRunner.prepare() does)")
            self.pool = torch.empty((self.max_size, *x.shape[1:]), device=x.device, dtype=torch.float32,
                                    memory_format=torch.channels_last if x.dim() == 4 else torch.contiguous_format)
            self.count = 0
        out = torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream
        if capturing or self._planned is not None:
            # the picks were drawn by plan() (or will be, before every replay): both launches run over the static table, B rows each
            if self.table is None or self.table.numel() != 3 * B or (not capturing and self._planned != B):
                raise RuntimeError("ReplayBuffer: push_and_pop(%d samples) under a capture / after plan() needs plan(%d) first" % (B, B))
            self._planned = None
            t = self.table.data_ptr()
            F.check(F.lib.migan_select_rows(self.pool.data_ptr(), x.data_ptr(), out.data_ptr(), t, None, B, D, st), "select_rows")
            F.check(F.lib.migan_select_rows(self.pool.data_ptr(), x.data_ptr(), self.pool.data_ptr(), t + 4 * B, t + 8 * B, B, D, st),
                    "select_rows")
            return out
        out_src, slot_src = self._draw(B)
        # both index tables in ONE host-to-device copy.  (A pinned staging buffer + non_blocking copy was tried: the
        # two-ranks-on-one-GPU gloo test hung with it in the tree; the cause was not isolated, so the pageable copy stays.)
        slots = sorted(slot_src)
        n = len(slots)
        table = out_src + [-1 - slot_src[j] for j in slots] + slots
        dev = torch.tensor(table, dtype=torch.int32).to(x.device)
        F.check(F.lib.migan_select_rows(self.pool.data_ptr(), x.data_ptr(), out.data_ptr(), dev.data_ptr(), None, B, D, st),
                "select_rows")
        if n:
            F.check(F.lib.migan_select_rows(self.pool.data_ptr(), x.data_ptr(), self.pool.data_ptr(), dev.data_ptr() + 4 * B,
                                            dev.data_ptr() + 4 * (B + n), n, D, st), "select_rows")
        return out

    def _push_and_pop_host(self, batch):
        out = []
        for k in range(batch.shape[0]):
            sample = batch.data[k:k + 1]
            if len(self.data) < self.max_size:
                self.data.append(sample)
                out.append(sample)
            elif random.uniform(0, 1) > 0.5:
                j = random.randint(0, self.max_size - 1)
                out.append(self.data[j].clone())
                self.data[j] = sample
            else:
                out.append(sample)
        return torch.cat(out)

    def samples(self):
        """The stored history as a list of (1, C, H, W) tensors, oldest slot first (tests)."""
        if self.pool is not None:
            return [self.pool[i:i + 1] for i in range(self.count)]
        return list(self.data)


class LambdaLR:
    """cyclegan/utils.py:36-44."""

    def __init__(self, n_epochs, offset, decay_start_epoch):
        if n_epochs - decay_start_epoch <= 0:
            raise AssertionError("Decay must start before the training session ends!")
        self.n_epochs, self.offset, self.decay_start_epoch = n_epochs, offset, decay_start_epoch

    def step(self, epoch):
        return 1.0 - max(0, epoch + self.offset - self.decay_start_epoch) / (self.n_epochs - self.decay_start_epoch)


def make_cyclegan_state(G_AB, G_BA, D_A, D_B, skip_dead_grads=True, dp=None):
    return SimpleNamespace(
        G_AB=G_AB, G_BA=G_BA, D_A=D_A, D_B=D_B,
        opt_G=Adam(itertools.chain(G_AB.parameters(), G_BA.parameters()), **ADAM),
        opt_D_A=Adam(D_A.parameters(), **ADAM), opt_D_B=Adam(D_B.parameters(), **ADAM), mse=gnn.MSELoss(),
        l1=gnn.L1Loss(), buf_A=ReplayBuffer(), buf_B=ReplayBuffer(), lambda_cyc=10.0, lambda_id=5.0,
        skip=skip_dead_grads, labels={}, dp=dp or LocalStepper())


@_scoped
def cyclegan_step(s, real_A, real_B):
    """cyclegan.py:159-239."""
    B = real_A.size(0)
    dev = real_A.device
    valid, fake = _labels(s, (B, *s.D_A.output_shape), dev)
    s.G_AB.train()
    s.G_BA.train()
    s.dp.begin_step()
    s.opt_G.zero_grad()
    chains = _CHAINS and _two_streams_ok(s, real_A) and _conv_params_only(s)
    under = _d_under_g_ok(s, real_A)

    def d_half(opt, D, real, buf, fake_img):   # cyclegan.py:203-233 up to loss_D_X.backward()
        opt.zero_grad()
        loss_real = s.mse(D(real), valid)
        fake_ = buf.push_and_pop(fake_img)
        loss_D = half_sum(loss_real, s.mse(D(fake_.detach()), fake))
        _backward(loss_D)
        return loss_D

    with contextlib.ExitStack() as region:   # everything that forks a stream; left with every stream joined, also on an exception
        if chains:
            # The generators' step is two independent halves (cyclegan.py:170-190): real_A -> G_AB -> fake_B -> {D_B, G_BA -> recov_A} with
            # the identity pass G_BA(real_A), and the mirror image from real_B.  They share nothing but the (read-only) weights until the
            # losses are added, so the second half's forward runs on a second stream beside the first: one half's InstanceNorm passes (HBM)
            # and its MFMA launches' stalls are filled by the other half's launches.  Autograd runs each backward node on its forward's
            # stream, so the backward is two-stream as well; every parameter-gradient launch of both halves goes to ONE stream
            # (functional.one_wgrad_stream) in the engine's node order, so the additions into the shared parameters' gradients are serial
            # and in the order of the one-stream step: bit-identical results.
            main, side = _c_stream(dev)
            region.enter_context(F.one_wgrad_stream())   # until the generators' backward is over
            # the backward nodes of the second chain run on its stream: whoever leaves the region (the optimiser steps) waits for them
            region.callback(lambda: torch.cuda.current_stream(dev).wait_stream(side))
            F.prefill_packs(dev)   # the step's planned weight packs exist before either half asks for one
            side.wait_stream(main)
            try:
                with F.two_streams(), frozen(s.D_A, s.D_B, enabled=True):
                    # host order = the one-stream body's order of calls: autograd numbers its nodes as they are made and walks them
                    # backward by that number, so the parameter-gradient launches reach their one stream in the same order as without
                    # the second stream
                    id_A = s.l1(s.G_BA(real_A), real_A)
                    with torch.cuda.stream(side):
                        id_B = s.l1(s.G_AB(real_B), real_B)
                    # (fake_B / fake_A feed a discriminator AND the other generator: functional.fork2 - their two gradients meet in the
                    # library's add, not in autograd's ATen accumulation)
                    fake_B, fake_B2 = F.fork2(s.G_AB(real_A))
                    loss_GAN_AB = s.mse(s.D_B(fake_B), valid)
                    with torch.cuda.stream(side):
                        fake_A, fake_A2 = F.fork2(s.G_BA(real_B))
                        loss_GAN_BA = s.mse(s.D_A(fake_A), valid)
                    cyc_A = s.l1(s.G_BA(fake_B2), real_A)
                    with torch.cuda.stream(side):
                        cyc_B = s.l1(s.G_AB(fake_A2), real_B)
            finally:
                main.wait_stream(side)
            loss_id = half_sum(id_A, id_B)
            loss_GAN = half_sum(loss_GAN_AB, loss_GAN_BA)
            loss_cycle = half_sum(cyc_A, cyc_B)
        else:
            loss_id = half_sum(s.l1(s.G_BA(real_A), real_A), s.l1(s.G_AB(real_B), real_B))
            with frozen(s.D_A, s.D_B, enabled=s.skip):
                fake_B, fake_B2 = F.fork2(s.G_AB(real_A))
                loss_GAN_AB = s.mse(s.D_B(fake_B), valid)
                fake_A, fake_A2 = F.fork2(s.G_BA(real_B))
                loss_GAN_BA = s.mse(s.D_A(fake_A), valid)
            loss_GAN = half_sum(loss_GAN_AB, loss_GAN_BA)
            loss_cycle = half_sum(s.l1(s.G_BA(fake_B2), real_A), s.l1(s.G_AB(fake_A2), real_B))
        loss_G = F.axpby(F.axpby(loss_GAN, loss_cycle, 1.0, s.lambda_cyc), loss_id, 1.0, s.lambda_id)
        if under:
            # both discriminator updates underneath the generators' backward (see dcgan_step): they need fake_A / fake_B of the forward
            # only, and the step is ~1900 launches whose dependent-launch gaps the second stream's kernels fill
            loss_D_A, loss_D_B = _fork_join(loss_G, lambda: (d_half(s.opt_D_A, s.D_A, real_A, s.buf_A, fake_A),
                                                             d_half(s.opt_D_B, s.D_B, real_B, s.buf_B, fake_B)))
        else:
            _backward(loss_G)
    if under:
        s.dp.step(s.opt_G)
        s.dp.step(s.opt_D_A)
        s.dp.step(s.opt_D_B)
    else:
        # the reference's order (and the data-parallel default, see set_dp_order): the generators' gradient exchange + Adam leave for the
        # data-parallel side stream here and run underneath the two discriminator updates
        s.dp.step(s.opt_G)
        loss_D_A = d_half(s.opt_D_A, s.D_A, real_A, s.buf_A, fake_A)
        s.dp.step(s.opt_D_A)
        loss_D_B = d_half(s.opt_D_B, s.D_B, real_B, s.buf_B, fake_B)
        s.dp.step(s.opt_D_B)
    return {"loss_G": loss_G.detach(), "loss_D": half_sum(loss_D_A, loss_D_B).detach(), "loss_GAN": loss_GAN.detach(),
            "loss_cycle": loss_cycle.detach(), "loss_identity": loss_id.detach()}


class CycleGanRunner:
    """`cyclegan_step` on static device batches, replayed as hipGraph(s).  What kept the step out of a capture was the replay
    buffers' host RNG (cyclegan/utils.py:21-30, called between the generator and discriminator phases, cyclegan.py:216,233); the draws
    need nothing from the device, so they move IN FRONT of the replay: `run()` draws buffer A's picks, then buffer B's - the order
    of the reference's calls - into each buffer's static index table (ReplayBuffer.plan) and replays the recorded launches, which
    read the tables.  Same random numbers, same picks, same arithmetic as the eager step (tests compare them bit for bit).  At one
    image per GPU - the reference's default batch (cyclegan.py:28) and the per-GPU shard of the 8-GPU configuration - the step is
    ~2000 launches of 5-60 us: host-bound when launched one by one.  `prepare()` runs `warmup` real steps."""

    def __init__(self, s, real_A, real_B, use_graph=True, warmup=2):
        from .graph import StepRunner

        self.s = s
        self.a, self.b = real_A.clone(), real_B.clone()
        self.runner = StepRunner(lambda: cyclegan_step(s, self.a, self.b), s.dp, use_graph, max(1, warmup), before_capture=self._reserve)

    def _reserve(self):
        # both buffers' static pick tables are the two halves of ONE device tensor, filled by one host-to-device copy per replay
        B = self.a.shape[0]
        for buf in (self.s.buf_A, self.s.buf_B):
            if buf.pool is None:
                raise RuntimeError("ReplayBuffer.reserve(): no device pool yet (run one eager push_and_pop first)")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ReplayBuffer: the static table must exist before a capture")
        self._tab = torch.full((6 * B,), -1, dtype=torch.int32, device=self.a.device)
        self.s.buf_A.table, self.s.buf_B.table = self._tab[:3 * B], self._tab[3 * B:]
        # two pinned staging tensors in rotation: the copy of replay i+1 is enqueued while replay i runs (a pageable source made
        # every plan() a host synchronisation with all queued GPU work - the launch-bound batch-1 regime the recording is for)
        self._stage = [torch.empty(6 * B, dtype=torch.int32).pin_memory() for _ in range(2)]
        self._stage_ev = [None, None]
        self._stage_k = 0

    def prepare(self):
        self.runner.prepare()
        return self

    graphed = property(lambda self: self.runner.graphed)
    capture_error = property(lambda self: self.runner.capture_error)

    def run(self, real_A=None, real_B=None):
        if real_A is not None:
            self.a.copy_(real_A)
        if real_B is not None:
            self.b.copy_(real_B)
        if self.runner.graphed:
            B = self.a.shape[0]
            # cyclegan.py:216 fake_A_buffer.push_and_pop(fake_A) draws first, cyclegan.py:233 fake_B_buffer.push_and_pop(fake_B) second
            rows = self.s.buf_A.plan_rows(B) + self.s.buf_B.plan_rows(B)
            k = self._stage_k
            self._stage_k ^= 1
            if self._stage_ev[k] is not None:
                self._stage_ev[k].synchronize()   # the copy that last read this staging tensor (two replays ago) is over
            self._stage[k].copy_(torch.tensor(rows, dtype=torch.int32))
            self._tab.copy_(self._stage[k], non_blocking=True)   # on the stream the replay is launched on
            ev = torch.cuda.Event()
            ev.record()
            self._stage_ev[k] = ev
            out = self.runner.run()
            # the replay consumed the tables (its push_and_pop launches are recorded, no Python ran): a later EAGER push_and_pop on this
            # state must draw its own picks instead of finding a stale plan
            self.s.buf_A._planned = self.s.buf_B._planned = None
            return out
        return self.runner.run()


# ------------------------------------------------------------------------------------------------ pix2pix
def make_pix2pix_state(G, D, img_size=256, skip_dead_grads=True, dp=None):
    return SimpleNamespace(G=G, D=D, opt_G=Adam(G.parameters(), **ADAM), opt_D=Adam(D.parameters(), **ADAM),
                           mse=gnn.MSELoss(), l1=gnn.L1Loss(), lambda_pixel=100.0,
                           patch=(1, img_size // 16, img_size // 16), skip=skip_dead_grads, labels={},
                           dp=dp or LocalStepper())


@_scoped
def pix2pix_step(s, real_A, real_B):
    """pix2pix.py:123-172 (real_A = condition image, real_B = target)."""
    # the batch as activations of this package (NHWC, GanTensor): `torch.cat((img_A, img_B), 1)` of pix2pix/models.py:132 then runs on the
    # library's concat kernel for the real pair as well (plain tensors took ATen's), and the condition image is re-laid once, not per pass
    if F.on_device(real_A) and real_A.dim() == 4:
        real_A, real_B = gnn._wrap(F.to_nhwc(real_A)), gnn._wrap(F.to_nhwc(real_B))
    valid, fake = _labels(s, (real_A.size(0), *s.patch), real_A.device)
    s.dp.begin_step()
    s.opt_G.zero_grad()
    fake_B, fake_B2 = F.fork2(s.G(real_A))   # two consumers (discriminator, pixel loss)
    with frozen(s.D, enabled=s.skip):
        loss_GAN = s.mse(s.D(fake_B, real_A), valid)
    loss_pixel = s.l1(fake_B2, real_B)
    loss_G = F.axpby(loss_GAN, loss_pixel, 1.0, s.lambda_pixel)

    def d_half():   # pix2pix.py:151-165 up to loss_D.backward()
        s.opt_D.zero_grad()
        loss_real = s.mse(s.D(real_B, real_A), valid)
        loss_fake = s.mse(s.D(fake_B.detach(), real_A), fake)
        loss_D = half_sum(loss_real, loss_fake)
        _backward(loss_D)
        return loss_D

    if _d_under_g_ok(s, real_A):   # the discriminator update underneath the generator's backward (see dcgan_step)
        loss_D = _fork_join(loss_G, d_half)
        s.dp.step(s.opt_G)
    else:
        _backward(loss_G)
        s.dp.step(s.opt_G)
        loss_D = d_half()
    s.dp.step(s.opt_D)
    return {"loss_G": loss_G.detach(), "loss_D": loss_D.detach(), "loss_pixel": loss_pixel.detach(),
            "loss_GAN": loss_GAN.detach()}


# ------------------------------------------------------------------------------------------------ srgan
def make_srgan_state(G, D, V, skip_dead_grads=True, dp=None):
    V.eval()
    return SimpleNamespace(G=G, D=D, V=V, opt_G=Adam(G.parameters(), **ADAM), opt_D=Adam(D.parameters(), **ADAM),
                           mse=gnn.MSELoss(), l1=gnn.L1Loss(), skip=skip_dead_grads, labels={},
                           dp=dp or LocalStepper())


@_scoped
def srgan_step(s, imgs_lr, imgs_hr):
    """srgan.py:97-145."""
    valid, fake = _labels(s, (imgs_lr.size(0), *s.D.output_shape), imgs_lr.device)
    s.dp.begin_step()
    s.opt_G.zero_grad()
    vside = _VREAL_SIDE and _two_streams_ok(s, imgs_lr)
    if vside:
        # feature_extractor(imgs_hr) of srgan.py:114 needs nothing from the generator and nobody differentiates it: it runs on the
        # second stream beside the generator's forward (whose 33 trunk convs are single-round launches with a latency-bound tail)
        main, side = _d_stream(imgs_lr.device)
        # the step's planned weight packs (G's, D's and V's) are written by ONE launch that rides on the first pack request: it runs
        # here, on the main stream, before the fork - otherwise the side stream's first VGG conv would issue it and the generator's
        # convs on the main stream would read their arena views with no dependency on that launch
        F.prefill_packs(imgs_lr.device)
        side.wait_stream(main)
    try:
        # two_streams(): a pack made inside the forked region carries the event of its launch, a hit from the other stream waits for it
        with F.two_streams() if vside else contextlib.nullcontext():
            if vside:
                with torch.cuda.stream(side), torch.no_grad():
                    real_features = s.V(imgs_hr)
            gen_hr, gen_hr2 = F.fork2(s.G(imgs_lr))   # two consumers (discriminator, VGG): their gradients meet in the library's add
            with frozen(s.D, s.V, enabled=s.skip):
                loss_GAN = s.mse(s.D(gen_hr), valid)
                gen_features = s.V(gen_hr2)
    finally:
        if vside:
            main.wait_stream(side)   # join - also when a forward raised: no stream stays forked
    if not vside:
        if s.skip:
            with torch.no_grad():
                real_features = s.V(imgs_hr)
        else:
            real_features = s.V(imgs_hr)
    loss_content = s.l1(gen_features, real_features.detach())
    # (The frozen VGG19 passes on a third stream - real features beside the generator's forward, gen features and their backward beside
    # the discriminator's pass over gen_hr - were measured: 83.97 vs 83.15-83.41 ms without, profiles/r04_ab.txt call 13; removed.)
    loss_G = F.axpby(loss_content, loss_GAN, 1.0, 1e-3)

    def d_half():   # srgan.py:129-141 up to loss_D.backward()
        s.opt_D.zero_grad()
        loss_real = s.mse(s.D(imgs_hr), valid)
        loss_fake = s.mse(s.D(gen_hr.detach()), fake)
        loss_D = half_sum(loss_real, loss_fake)
        _backward(loss_D)
        return loss_D

    if _d_under_g_ok(s, imgs_lr):   # the discriminator update underneath the generator's backward (see dcgan_step)
        loss_D = _fork_join(loss_G, d_half)
        s.dp.step(s.opt_G)
    else:
        _backward(loss_G)
        s.dp.step(s.opt_G)
        loss_D = d_half()
    s.dp.step(s.opt_D)
    return {"loss_G": loss_G.detach(), "loss_D": loss_D.detach(), "loss_content": loss_content.detach(),
            "loss_GAN": loss_GAN.detach()}


# ------------------------------------------------------------------------------------------------ esrgan
def make_esrgan_state(G, D, V, skip_dead_grads=True, dp=None, warmup_batches=500, lambda_adv=5e-3, lambda_pixel=1e-2):
    """esrgan.py:60-83; Adam betas (0.9, 0.999) are this script's defaults (esrgan.py:40-41)."""
    V.eval()
    adam = dict(lr=2e-4, betas=(0.9, 0.999))
    return SimpleNamespace(G=G, D=D, V=V, opt_G=Adam(G.parameters(), **adam), opt_D=Adam(D.parameters(), **adam),
                           bce_logits=gnn.BCEWithLogitsLoss(), l1_content=gnn.L1Loss(), l1_pixel=gnn.L1Loss(),
                           warmup_batches=warmup_batches, lambda_adv=lambda_adv, lambda_pixel=lambda_pixel,
                           skip=skip_dead_grads, labels={}, dp=dp or LocalStepper())


@_scoped
def esrgan_step(s, imgs_lr, imgs_hr, batches_done):
    """esrgan.py:101-174: pixel-loss warm-up, then the relativistic average GAN step.  `pred - other.mean(0, keepdim=True)`
    is one launch (functional.sub_batch_mean); with skip_dead_grads the D / VGG weight gradients of the G step and the
    graph of D(real) (detached by the reference, esrgan.py:133) are not computed."""
    valid, fake = _labels(s, (imgs_lr.size(0), *s.D.output_shape), imgs_lr.device)
    s.dp.begin_step()
    s.opt_G.zero_grad()
    gen_hr = s.G(imgs_lr)
    loss_pixel = s.l1_pixel(gen_hr, imgs_hr)
    if batches_done < s.warmup_batches:
        _backward(loss_pixel)
        s.dp.step(s.opt_G)
        return {"loss_pixel": loss_pixel.detach()}
    with frozen(s.D, s.V, enabled=s.skip):
        if s.skip:
            with torch.no_grad():
                pred_real = s.D(imgs_hr)
                real_features = s.V(imgs_hr)
        else:
            pred_real = s.D(imgs_hr).detach()
            real_features = s.V(imgs_hr).detach()
        pred_fake = s.D(gen_hr)
        loss_GAN = s.bce_logits(F.sub_batch_mean(pred_fake, pred_real), valid)
        loss_content = s.l1_content(s.V(gen_hr), real_features)
    loss_G = F.axpby(F.axpby(loss_content, loss_GAN, 1.0, s.lambda_adv), loss_pixel, 1.0, s.lambda_pixel)
    _backward(loss_G)
    s.dp.step(s.opt_G)
    s.opt_D.zero_grad()
    pred_real = s.D(imgs_hr)
    pred_fake = s.D(gen_hr.detach())
    loss_real = s.bce_logits(F.sub_batch_mean(pred_real, pred_fake), valid)
    loss_fake = s.bce_logits(F.sub_batch_mean(pred_fake, pred_real), fake)
    loss_D = half_sum(loss_real, loss_fake)
    _backward(loss_D)
    s.dp.step(s.opt_D)
    return {"loss_G": loss_G.detach(), "loss_D": loss_D.detach(), "loss_content": loss_content.detach(),
            "loss_GAN": loss_GAN.detach(), "loss_pixel": loss_pixel.detach()}


@torch.no_grad()
def esrgan_upscale(G, image):
    """test_on_image.py:24-37, the reference's only inference entry point: generator.eval(); sr = generator(image) under
    no_grad (de-normalisation and the PNG write stay with the caller)."""
    G.eval()
    return G(image)


# ------------------------------------------------------------------------------------------------ acgan
def make_acgan_state(G, D, latent_dim=100, n_classes=10, skip_dead_grads=True, dp=None):
    """acgan.py:110-152: BCELoss for validity, CrossEntropyLoss for the auxiliary class head, Adam(2e-4, (0.5, 0.999))."""
    return SimpleNamespace(G=G, D=D, opt_G=Adam(G.parameters(), **ADAM), opt_D=Adam(D.parameters(), **ADAM), bce=gnn.BCELoss(),
                           ce=gnn.CrossEntropyLoss(), latent_dim=latent_dim, n_classes=n_classes, skip=skip_dead_grads,
                           labels={}, dp=dp or LocalStepper())


@_scoped
def acgan_step(s, real_imgs, labels, z, gen_labels):
    """acgan.py:167-222.  labels / gen_labels: int64 class indices on the device; z: (B, latent_dim).  The reference's quirk is
    kept: CrossEntropyLoss consumes the Softmax output of the class head."""
    valid, fake = _labels(s, (real_imgs.shape[0], 1), real_imgs.device)
    s.dp.begin_step()
    s.opt_G.zero_grad()
    gen_imgs = s.G(z, gen_labels)
    with frozen(s.D, enabled=s.skip):
        validity, pred_label = s.D(gen_imgs)
        g_loss = half_sum(s.bce(validity, valid), s.ce(pred_label, gen_labels))
    _backward(g_loss)
    s.dp.step(s.opt_G)
    s.opt_D.zero_grad()
    real_pred, real_aux = s.D(real_imgs)
    d_real_loss = half_sum(s.bce(real_pred, valid), s.ce(real_aux, labels))
    fake_pred, fake_aux = s.D(gen_imgs.detach())
    d_fake_loss = half_sum(s.bce(fake_pred, fake), s.ce(fake_aux, gen_labels))
    d_loss = half_sum(d_real_loss, d_fake_loss)
    _backward(d_loss)
    s.dp.step(s.opt_D)
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gen_imgs": gen_imgs.detach()}



# ------------------------------------------------------------------------------------------------ DCGAN-block clones (8f F2)
def make_clone_state(G, D, skip_dead_grads=True, dp=None):
    """lsgan.py:107-135, relativistic_gan.py:95-118, ebgan.py:105-137: swapped networks on the GPU + Adam(2e-4, (0.5, 0.999))."""
    return SimpleNamespace(G=G, D=D, opt_G=Adam(G.parameters(), **ADAM), opt_D=Adam(D.parameters(), **ADAM), mse=gnn.MSELoss(),
                           bce_logits=gnn.BCEWithLogitsLoss(), skip=skip_dead_grads, labels={}, dp=dp or LocalStepper())


@_scoped
def lsgan_step(s, real_imgs, z):
    """lsgan.py:140-180: the dcgan.py loop with MSELoss on the unbounded validity."""
    valid, fake = _labels(s, (real_imgs.shape[0], 1), real_imgs.device)
    s.dp.begin_step()
    s.opt_G.zero_grad()
    gen = s.G(z)
    with frozen(s.D, enabled=s.skip):
        g_loss = s.mse(s.D(gen), valid)
    _backward(g_loss)
    s.dp.step(s.opt_G)
    s.opt_D.zero_grad()
    d_loss = half_sum(s.mse(s.D(real_imgs), valid), s.mse(s.D(gen.detach()), fake))
    _backward(d_loss)
    s.dp.step(s.opt_D)
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gen_imgs": gen.detach()}


@_scoped
def relativistic_gan_step(s, real_imgs, z, rel_avg_gan=False):
    """relativistic_gan.py:126-182.  The generator step's first two discriminator forwards (relativistic_gan.py:148-149) feed a
    loss that line 157 overwrites; they are still run (without building a graph): they draw Dropout2d masks and move the
    BatchNorm running statistics exactly as in the reference."""
    valid, fake = _labels(s, (real_imgs.shape[0], 1), real_imgs.device)
    s.dp.begin_step()
    s.opt_G.zero_grad()
    gen = s.G(z)
    with torch.no_grad():  # side effects only (the reference discards this loss)
        s.D(real_imgs)
        s.D(gen.detach())
    with frozen(s.D, enabled=s.skip):
        g_loss = s.bce_logits(s.D(gen), valid)
    _backward(g_loss)
    s.dp.step(s.opt_G)
    s.opt_D.zero_grad()
    real_pred = s.D(real_imgs)
    fake_pred = s.D(gen.detach())
    if rel_avg_gan:
        real_loss = s.bce_logits(F.sub_batch_mean(real_pred, fake_pred), valid)
        fake_loss = s.bce_logits(F.sub_batch_mean(fake_pred, real_pred), fake)
    else:
        real_loss = s.bce_logits(F.axpby(real_pred, fake_pred, 1.0, -1.0), valid)
        fake_loss = s.bce_logits(F.axpby(fake_pred, real_pred, 1.0, -1.0), fake)
    d_loss = half_sum(real_loss, fake_loss)
    _backward(d_loss)
    s.dp.step(s.opt_D)
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gen_imgs": gen.detach()}


@_scoped
def ebgan_step(s, real_imgs, z, opt_batch_size=64, lambda_pt=0.1):
    """ebgan.py:159-202: auto-encoder discriminator; the hinge on the fake reconstruction error is decided on the host from
    .item(), as the reference does (one sync per step; this loop is therefore not graph-captured)."""
    margin = max(1, opt_batch_size / 64.0)
    s.dp.begin_step()
    s.opt_G.zero_grad()
    gen = s.G(z)
    with frozen(s.D, enabled=s.skip):
        recon, emb = s.D(gen)
        g_loss = F.axpby(s.mse(recon, gen.detach()), F.pullaway_loss(emb), 1.0, lambda_pt)
    _backward(g_loss)
    s.dp.step(s.opt_G)
    s.opt_D.zero_grad()
    real_recon, _ = s.D(real_imgs)
    fake_recon, _ = s.D(gen.detach())
    d_loss_real = s.mse(real_recon, real_imgs)
    d_loss_fake = s.mse(fake_recon, gen.detach())
    d_loss = d_loss_real
    if margin - float(d_loss_fake.detach()) > 0:
        key = ("margin", str(real_imgs.device))
        if key not in s.labels:
            s.labels[key] = torch.ones((), device=real_imgs.device)
        # d_loss_real + (margin - d_loss_fake); the constant rides on a device scalar so the value matches the reference's
        d_loss = F.axpby(F.axpby(d_loss_real, d_loss_fake, 1.0, -1.0), s.labels[key], 1.0, float(margin))
    _backward(d_loss)
    s.dp.step(s.opt_D)
    return {"g_loss": g_loss.detach(), "d_loss": d_loss.detach(), "gen_imgs": gen.detach()}
