#!/usr/bin/env python
"""Headline benchmark: training images/sec of the GAN hot path on MI355X (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W [--workload dcgan|cyclegan|srgan|wgan_gp|pix2pix|esrgan]
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Default workload = BASELINE.json configs[1]: DCGAN 64x64, batch 128 per GPU, fp32.  One "step" = one full iteration of
the reference loop (dcgan.py:143-183: G forward/backward/Adam, three D forwards, D backward/Adam) on synthetic data
resident in HBM, every op in libmigan.so.  Launch-bound steps are captured once as hipGraphs and replayed (N>1: graph
segments with the RCCL all-reduce + fused Adam between them on a side stream).

Timing: W warm-up steps, then R >= 1 blocks of EXACTLY K steps, each block bracketed by barrier + synchronize on both
sides and reduced with MAX over ranks; R is chosen so that the blocks cover >= --min-seconds (default 2 s).  `value` is
computed from the MEDIAN block (`ms_per_step`); the minimum and every block time are reported next to it.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant conv kernel of the timed workload, measured live with
HIP events on the launch stream: `frac` = EXECUTED FLOPs / time / peak (what the matrix pipe did), `frac_dense` = the
reference's dense 2*M*N*K / time / peak (exceeds 1 where the phase-collapsed Upsample+Conv kernels skip
20/36 of the dense work).  At N=1 the other BASELINE configs (cyclegan 256x256 bs 8, srgan 96->384 bs 16, wgan_gp bs 64)
are measured briefly after the headline run and reported under `extra`; `cpu_baseline` is the oracle loop timed on this
box's host cores (bounded sample).  See DESIGN.md §Measurement.
"""
import argparse
import contextlib
import copy
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

IMG, BATCH, LATENT, CH = 64, 128, 100, 1
PEAK_TFLOPS = 157.3  # fp32-input MFMA, MI355X_MICROARCH.md chip table
UP_EXEC = 16.0 / 36.0  # executed share of the dense FLOPs in the phase-collapsed Upsample(2)+Conv3x3 kernels
PROFILE_ROUND = "r06"  # profiles/r06_pmc_kernels.json: the rocprofv3 --pmc passes over `bench.py --pmc-log` (tools/gpu_r06.sh pmcstep)


def dcgan_flops_per_image(ch=None):
    """Algorithmic FLOPs/img of one training step (step = 3*G_fwd + 8*D_fwd)."""
    ch = CH if ch is None else ch
    s = IMG // 4
    g = 2 * LATENT * 128 * s * s + 2 * (2 * s) ** 2 * 128 * 128 * 9 + 2 * (4 * s) ** 2 * 64 * 128 * 9 \
        + 2 * (4 * s) ** 2 * ch * 64 * 9
    d, h, cin = 0, IMG, ch
    for cout in (16, 32, 64, 128):
        h //= 2
        d += 2 * h * h * cout * cin * 9
        cin = cout
    d += 2 * 128 * h * h
    return 3 * g + 8 * d


def dcgan_upconv_flops_per_image():
    """Dense FLOPs/img per step spent in the two Upsample+Conv3x3 layers (fwd + dgrad + wgrad)."""
    s = IMG // 4
    return 3 * (2 * (2 * s) ** 2 * 128 * 128 * 9 + 2 * (4 * s) ** 2 * 64 * 128 * 9)


def esrgan_flops_per_image(hr=256, blocks=23):
    """Algorithmic FLOPs per image of one full (post warm-up) ESRGAN step, esrgan.py:101-174 at its defaults (hr 256, 23 RRDB):
    3 G (fwd + dgrad + wgrad) + 9 D (detached fwd; fwd + dgrad in the G step; 2 x (fwd + dgrad + wgrad)) + 3 VGG19[:35]."""
    lr = hr // 4
    conv = lambda hw, ci, co: 2.0 * hw * hw * ci * co * 9  # noqa: E731
    dense = sum(conv(lr, 64 * i, 64) for i in range(1, 6))
    g = conv(lr, 3, 64) + 3 * blocks * dense + conv(lr, 64, 64) + conv(lr, 64, 256) + conv(2 * lr, 64, 256) \
        + conv(hr, 64, 64) + conv(hr, 64, 3)
    d, c, h = 0.0, 3, hr
    for co in (64, 128, 256, 512):
        d += conv(h, c, co) + conv(h // 2, co, co)
        c, h = co, h // 2
    d += conv(h, 512, 1)
    v, c, h = 0.0, 3, hr
    for stage, (co, n) in enumerate(((64, 2), (128, 2), (256, 4), (512, 4), (512, 4))):
        for _ in range(n):
            v += conv(h, c, co)
            c = co
        h //= 2
    return 3 * g + 9 * d + 3 * v


# Per-image algorithmic GFLOP of one training step and the share of it inside Upsample+Conv3x3 layers
# (fwd+dgrad+wgrad; CycleGAN: u128 + u64 = 154.62 GFLOP per G forward at bs 8, 18 G-forward equivalents per step).
GFLOP_PER_IMG = {"dcgan": dcgan_flops_per_image() / 1e9, "dcgan_ch3": dcgan_flops_per_image(3) / 1e9, "cyclegan": 2097.99, "srgan": 541.43, "pix2pix": 65.52,
                 "wgan_gp": 0.0219, "esrgan": esrgan_flops_per_image() / 1e9}
UPCONV_GFLOP_PER_IMG = {"dcgan": dcgan_upconv_flops_per_image() / 1e9, "dcgan_ch3": dcgan_upconv_flops_per_image() / 1e9, "cyclegan": 18 * 154.62 / 8, "srgan": 0.0,
                        "pix2pix": 0.0, "wgan_gp": 0.0, "esrgan": 0.0}
WORKLOAD_NAME = {
    "dcgan": "implementations/dcgan 64x64 bs=128 per GPU fp32 (dcgan.py:143-183 full step)",
    "dcgan_ch3": "implementations/dcgan 64x64 bs=128 per GPU fp32 with --channels 3 (SURVEY.md 8d: \"also report ch 3\")",
    "cyclegan": "implementations/cyclegan 256x256 bs=8 per GPU fp32, ResNet-9 G + PatchGAN D (cyclegan.py:159-239 full step)",
    "srgan": "implementations/srgan 96->384 bs=16 per GPU fp32 (srgan.py:97-145 full step)",
    "wgan_gp": "implementations/wgan_gp 32x32 bs=64 per GPU fp32, one critic iteration incl. gradient penalty; generator "
               "update every 5th (wgan_gp.py:146-193)",
    "pix2pix": "implementations/pix2pix 256x256 bs=1 per GPU fp32 (pix2pix.py:123-172 full step)",
    "esrgan": "implementations/esrgan 64->256 bs=4 per GPU fp32, 23 RRDB, relativistic step after warm-up (esrgan.py:101-174)",
}


def executed_gflop_per_image(w):
    return GFLOP_PER_IMG[w] - UPCONV_GFLOP_PER_IMG[w] * (1.0 - UP_EXEC)


# ------------------------------------------------------------------------------------------------ workloads
class Workload:
    """name, per-GPU batch, `run(i)` = one training step on static device inputs, `init` = initial CPU state dicts."""

    def __init__(self, name, batch, run, state, init=None, graphed=False, capture_error=None, nets=()):
        self.name, self.batch, self.run, self.state, self.init = name, batch, run, state, init
        self.graphed, self.capture_error, self.nets = graphed, capture_error, nets


def build_dcgan(dp, rank, dev, args, nsteps, ch=None):
    from pytorch_gan_amd import graph as gmod
    from pytorch_gan_amd import models, steps

    ch = CH if ch is None else ch   # dcgan.py:28 --channels (default 1; 3 is reported under extra.dcgan_ch3)
    torch.manual_seed(0)
    G = models.DcganGenerator(IMG, LATENT, ch)
    D = models.DcganDiscriminator(IMG, ch)
    G.apply(models.init_normal_dcgan)   # dcgan.py:115-116
    D.apply(models.init_normal_dcgan)
    init = (copy.deepcopy(G.state_dict()), copy.deepcopy(D.state_dict()))
    G, D = G.cuda(), D.cuda()
    if dp.world > 1:
        dp.broadcast_parameters(G, D)
    state = steps.make_gan_state(G, D, LATENT, skip_dead_grads=True, dp=dp)
    batch = args.batch or BATCH
    # SURVEY.md 8e "Partitioning": ONE globally seeded draw of the global batch (and of every step's z), sliced by rank - an N-rank
    # run consumes the random numbers the single-process run of the global batch consumes
    rng = np.random.RandomState(1234)
    lo, hi = rank * batch, (rank + 1) * batch
    real = torch.from_numpy(rng.uniform(-1, 1, (dp.world * batch, ch, IMG, IMG)).astype(np.float32)[lo:hi]).to(dev)
    nz = min(nsteps, 64) + 8
    zs = torch.from_numpy(rng.normal(0, 1, (nz, dp.world * batch, LATENT)).astype(np.float32)[:, lo:hi].copy()).to(dev)
    z_static = zs[0].clone()
    runner = gmod.StepRunner(lambda: steps.dcgan_step(state, real, z_static), dp, use_graph=not args.no_graph)
    runner.prepare()

    def run(i):
        z_static.copy_(zs[i % nz])
        return runner.run()

    w = Workload("dcgan" if ch == CH else "dcgan_ch%d" % ch, batch, run, state, init, runner.graphed, runner.capture_error, (G, D))
    w.eager = lambda: steps.dcgan_step(state, real, z_static)
    return w


def build_dcgan_ch3(dp, rank, dev, args, nsteps):
    return build_dcgan(dp, rank, dev, args, nsteps, ch=3)


def build_cyclegan(dp, rank, dev, args, nsteps):
    from pytorch_gan_amd import models, steps

    torch.manual_seed(0)
    random.seed(0)
    shape = (3, 256, 256)
    nets = [models.CycleGenerator(shape, 9), models.CycleGenerator(shape, 9), models.CycleDiscriminator(shape),
            models.CycleDiscriminator(shape)]
    for n in nets:
        n.apply(models.init_normal_cyclegan)   # cyclegan.py:79-83
    nets = [n.to(dev) for n in nets]
    if dp.world > 1:
        dp.broadcast_parameters(*nets)
    state = steps.make_cyclegan_state(*nets, dp=dp)
    batch = args.batch or 8
    rng = np.random.RandomState(4321)   # one global draw, sliced by rank
    lo, hi = rank * batch, (rank + 1) * batch
    a = torch.from_numpy(rng.uniform(-1, 1, (dp.world * batch, *shape)).astype(np.float32)[lo:hi]).to(dev)
    b = torch.from_numpy(rng.uniform(-1, 1, (dp.world * batch, *shape)).astype(np.float32)[lo:hi]).to(dev)
    # replay buffers warm (>= 50 entries) before timing, so the picks and clones of the timed steps are
    # those of a run in steady state.  The histories are filled with generator outputs under no_grad - no training step, no
    # collective: at --global-batch 2 the former 50 warm-up STEPS each moved 113 MB of gradients through gloo (the two-rank
    # single-GPU test mode: 2-3 s per step), which is what made that launch take > 150 s in round 3.
    with torch.no_grad():
        while len(state.buf_A) < state.buf_A.max_size:
            state.buf_A.push_and_pop(state.G_BA(b))
            state.buf_B.push_and_pop(state.G_AB(a))
    # the recorded step: the replay buffers' host draws (python `random`, the reference's order) happen in front of every replay
    # into static device tables (steps.CycleGanRunner); --no-graph launches the same step eagerly
    # Recorded where the step is launch-bound (<= 2 images per GPU: 33.0 ms recorded against 45 ms eager at one image), eager at batch 8.
    # Round 6 removed the 238 per-step pack launches the batch-8 recording held (graph.StepRunner builds the pack plan's tables in front of
    # the capture) and re-measured on one box: 142.3 ms recorded against 136.1 / 136.5 ms eager (profiles/r06_ab.txt calls 38-39) - with
    # ~2200 kernel nodes of 60 us on three forked streams the replay's node-to-node hand-over costs more than the CPU launches it saves,
    # which the GPU hides at this size anyway.  --graph-always records at every batch size, --no-graph never.
    use_graph = (not args.no_graph) and (batch <= 2 or args.graph_always)
    runner = steps.CycleGanRunner(state, a, b, use_graph=use_graph, warmup=1).prepare()
    w = Workload("cyclegan", batch, lambda i: runner.run(), state, None, runner.graphed, runner.capture_error, tuple(nets))
    w.eager = lambda: steps.cyclegan_step(state, a, b)
    return w


def build_srgan(dp, rank, dev, args, nsteps):
    from pytorch_gan_amd import models, steps

    torch.manual_seed(0)
    G, D, V = models.SrganGenerator(), models.SrganDiscriminator((3, 384, 384)), models.SrganFeatureExtractor()
    G, D, V = G.to(dev), D.to(dev), V.to(dev)
    if dp.world > 1:
        dp.broadcast_parameters(G, D, V)
    state = steps.make_srgan_state(G, D, V, dp=dp)
    batch = args.batch or 16
    g = torch.Generator().manual_seed(99)   # one global draw, sliced by rank
    lo, hi = rank * batch, (rank + 1) * batch
    lr = torch.randn(dp.world * batch, 3, 96, 96, generator=g)[lo:hi].to(dev)
    hr = torch.randn(dp.world * batch, 3, 384, 384, generator=g)[lo:hi].to(dev)

    def run(i):
        dp.begin_step()
        out = steps.srgan_step(state, lr, hr)
        dp.end_step()
        return out

    w = Workload("srgan", batch, run, state, None, False, None, (G, D))
    w.eager = lambda: steps.srgan_step(state, lr, hr)
    return w


def build_esrgan(dp, rank, dev, args, nsteps):
    from pytorch_gan_amd import models, steps

    torch.manual_seed(0)
    G, D, V = models.EsrganGenerator(3, 64, 23), models.EsrganDiscriminator((3, 256, 256)), models.EsrganFeatureExtractor()
    G, D, V = G.to(dev), D.to(dev), V.to(dev)
    if dp.world > 1:
        dp.broadcast_parameters(G, D, V)
    state = steps.make_esrgan_state(G, D, V, dp=dp, warmup_batches=0)  # timed steps are the full relativistic step
    batch = args.batch or 4
    g = torch.Generator().manual_seed(99)
    lo, hi = rank * batch, (rank + 1) * batch
    lr = torch.randn(dp.world * batch, 3, 64, 64, generator=g)[lo:hi].to(dev)
    hr = torch.randn(dp.world * batch, 3, 256, 256, generator=g)[lo:hi].to(dev)

    def run(i):
        dp.begin_step()
        out = steps.esrgan_step(state, lr, hr, i)
        dp.end_step()
        return out

    w = Workload("esrgan", batch, run, state, None, False, None, (G, D))
    w.eager = lambda: steps.esrgan_step(state, lr, hr, 0)
    return w


def build_wgan_gp(dp, rank, dev, args, nsteps):
    from pytorch_gan_amd import models, steps

    torch.manual_seed(0)
    G, D = models.MlpGenerator((1, 32, 32), 100).to(dev), models.MlpCritic((1, 32, 32)).to(dev)
    if dp.world > 1:
        dp.broadcast_parameters(G, D)
    state = steps.make_wgan_gp_state(G, D, dp=dp)
    batch = args.batch or 64
    rng = np.random.RandomState(777)
    lo, hi = rank * batch, (rank + 1) * batch
    real = torch.from_numpy(rng.uniform(-1, 1, (dp.world * batch, 1, 32, 32)).astype(np.float32)[lo:hi]).to(dev)
    zs = torch.from_numpy(rng.normal(0, 1, (64, dp.world * batch, 100)).astype(np.float32)[:, lo:hi].copy()).to(dev)
    alphas = torch.from_numpy(rng.random_sample((64, dp.world * batch, 1, 1, 1)).astype(np.float32)[:, lo:hi].copy()).to(dev)
    runner = steps.WganGpRunner(state, batch, (1, 32, 32), use_graph=not args.no_graph).prepare(real, zs[0], alphas[0])
    za = torch.cat([zs.reshape(64, -1), alphas.reshape(64, -1)], 1).contiguous()   # the draws of an iteration, packed: one staging copy

    def run(i):
        return runner.run(i, None, None, None, packed=za[i % 64])

    w = Workload("wgan_gp", batch, run, state, None, runner.graphed, runner.capture_error, (G, D))
    w.eager = lambda: steps.wgan_gp_step(state, real, 1, zs[0], alphas[0])
    return w


def build_pix2pix(dp, rank, dev, args, nsteps):
    from pytorch_gan_amd import graph as gmod
    from pytorch_gan_amd import models, steps

    torch.manual_seed(0)
    G, D = models.Pix2pixGenerator(), models.Pix2pixDiscriminator()
    G.apply(models.init_normal_dcgan)
    D.apply(models.init_normal_dcgan)
    G, D = G.to(dev), D.to(dev)
    if dp.world > 1:
        dp.broadcast_parameters(G, D)
    state = steps.make_pix2pix_state(G, D, 256, dp=dp)
    batch = args.batch or 1
    rng = np.random.RandomState(555)
    lo, hi = rank * batch, (rank + 1) * batch
    a = torch.from_numpy(rng.uniform(-1, 1, (dp.world * batch, 3, 256, 256)).astype(np.float32)[lo:hi]).to(dev)
    b = torch.from_numpy(rng.uniform(-1, 1, (dp.world * batch, 3, 256, 256)).astype(np.float32)[lo:hi]).to(dev)
    runner = gmod.StepRunner(lambda: steps.pix2pix_step(state, a, b), dp, use_graph=not args.no_graph).prepare()
    w = Workload("pix2pix", batch, lambda i: runner.run(), state, None, runner.graphed, runner.capture_error, (G, D))
    w.eager = lambda: steps.pix2pix_step(state, a, b)
    return w


BUILDERS = {"dcgan": build_dcgan, "dcgan_ch3": build_dcgan_ch3, "cyclegan": build_cyclegan, "srgan": build_srgan, "wgan_gp": build_wgan_gp,
            "pix2pix": build_pix2pix, "esrgan": build_esrgan}


# ------------------------------------------------------------------------------------------------ timing
def timed_blocks(w, world, dev, steps, warmup, min_seconds, max_blocks=200):
    """W warm-up steps, then blocks of exactly `steps` steps (barrier + synchronize on both sides, MAX over ranks) until
    `min_seconds` of timed work; returns the block times in seconds and the last step's outputs."""
    import torch.distributed as dist

    for i in range(warmup):
        out = w.run(i)
    done, blocks, out = warmup, [], None
    while True:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            out = w.run(done + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        done += steps
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        blocks.append(el)
        total = sum(blocks)
        more = total < min_seconds and len(blocks) < max_blocks
        if world > 1:   # every rank must take the same decision
            flag = torch.tensor([1.0 if more else 0.0], device=dev)
            dist.broadcast(flag, src=0)
            more = bool(flag.item() > 0.5)
        if not more:
            return blocks, out


def summarise(name, batch, world, steps, blocks):
    med = float(np.median(blocks))
    ips = world * batch * steps / med
    return {
        "images_per_s": round(ips, 3), "ms_per_step": round(1e3 * med / steps, 4),
        "ms_per_step_min": round(1e3 * min(blocks) / steps, 4), "ms_per_step_max": round(1e3 * max(blocks) / steps, 4),
        "blocks": len(blocks), "timed_seconds": round(sum(blocks), 3),
        "step_executed_frac": round(ips * executed_gflop_per_image(name) * 1e9 / (PEAK_TFLOPS * 1e12 * world), 4),
        "step_dense_frac": round(ips * GFLOP_PER_IMG[name] * 1e9 / (PEAK_TFLOPS * 1e12 * world), 4),
    }


# ------------------------------------------------------------------------------------------------ per-kernel roofline
class ConvProfiler:
    """Times every conv-family launch with HIP events on the launch stream and attributes it to the kernel the library
    picks (migan_igemm_tile_code), with its dense (reference) and executed FLOPs."""

    def __init__(self, segments=None):
        from pytorch_gan_amd._lib import lib

        self.lib, self.records, self.orig = lib, [], {}
        # --pmc-log: [group, first, last) ordinals of the library's launches (migan_debug_launch_count, counted from process start)
        # that each wrapped call issued - a rocprofv3 --pmc pass over the same deterministic command lists the same launches in the
        # same order, so tools/pmc_step.py attributes its counter rows to the roofline groups without marker kernels
        self.segments = segments

    def _wrap(self, name, describe):
        fn = getattr(self.lib, name)
        self.orig[name] = fn

        count = self.lib.migan_debug_launch_count

        def wrapper(*a):
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            i0 = count(None)
            e0.record(st)
            rc = fn(*a)
            e1.record(st)
            d = describe(a)
            self.records.append((d, e0, e1))
            if self.segments is not None:
                self.segments.append({"group": d[0], "first": i0, "last": count(None), "dense": d[1], "executed": d[2]})
            return rc

        setattr(self.lib, name, wrapper)

    def __enter__(self):
        tc = self.lib.migan_igemm_tile_code

        def shape(N, Ci, Co, Ho, R, s):
            return "[%dx %d->%d k%d%s @%d]" % (N, Ci, Co, R, "s2" if s == 2 else "", Ho)

        def fwd(a):
            N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride = a[4:14]
            code = tc(N * Ho * Wo, Co, Ci, 1)
            f = 2.0 * N * Ho * Wo * Co * Ci * R * S
            return ("fwd_igemm_%d%s" % (code, shape(N, Ci, Co, Ho, R, stride)), f, f)

        def dgrad(a):
            N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride = a[4:14]
            ncls = stride * stride
            maxm = N * ((Hi + stride - 1) // stride) * ((Wi + stride - 1) // stride)
            code = tc(maxm, Ci, Co, ncls)
            f = 2.0 * N * Ho * Wo * Co * Ci * R * S
            return ("dgrad_igemm_%d%s" % (code, shape(N, Ci, Co, Ho, R, stride)), f, f)

        def wgrad(a):
            N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride = a[5:15]
            f = 2.0 * N * Ho * Wo * Co * Ci * R * S
            return ("wgrad%s" % shape(N, Ci, Co, Ho, R, stride), f, f)

        # phase-collapsed Upsample(2)->Conv3x3: dense FLOPs are the reference's 2*M*N*K on the upsampled grid (SURVEY.md
        # 8d); the kernels execute 16/36 of them
        def up_fwd(a):
            N, H, W, Ci, Co = a[4:9]
            f = 2.0 * N * 4 * H * W * Co * Ci * 9
            return ("upconv_fwd[%dx %d->%d @%d]" % (N, Ci, Co, 2 * H), f, f * UP_EXEC)

        def up_dgrad(a):
            N, H, W, Ci, Co = a[3:8]
            f = 2.0 * N * 4 * H * W * Co * Ci * 9
            return ("upconv_dgrad[%dx %d->%d @%d]" % (N, Ci, Co, 2 * H), f, f * UP_EXEC)

        def up_wgrad(a):
            N, H, W, Ci, Co = a[5:10]
            f = 2.0 * N * 4 * H * W * Co * Ci * 9
            return ("upconv_wgrad[%dx %d->%d @%d]" % (N, Ci, Co, 2 * H), f, f * UP_EXEC)

        # width-Toeplitz thin-N path (csrc/thin_toeplitz.hip) and the reflect-1 input gradient: FLOPs of the layer as the
        # reference specifies it (the expansion executes S*Co' / (S*Co) of them on the matrix pipe; reported as dense = executed)
        def toep(kind, off):
            def describe(a):
                N, Hi, Wi, Ci, Ho = a[off:off + 5]
                if kind == "fwd":
                    Wo, Co, R, S = a[off + 5:off + 9]
                else:
                    Co, R, S = a[off + 5:off + 8]
                    Wo = Wi
                f = 2.0 * N * Ho * Wo * Co * Ci * R * S
                return ("toeplitz_%s%s" % (kind, shape(N, Ci, Co, Ho, R, 1)), f, f)
            return describe

        def reflect1(a):
            N, H, W, Ci, Co = a[3:8]
            f = 2.0 * N * H * W * Co * Ci * 9
            return ("dgrad_reflect1" + shape(N, Ci, Co, H, 3, 1), f, f)

        # image-input layers (csrc/rgb_conv.hip): FLOPs of the layer as the reference specifies it
        def rgb_fwd(a):
            N, H, W, Ci, Ho, Wo, Co, R, S = a[4:13]
            f = 2.0 * N * Ho * Wo * Co * Ci * R * S
            return ("rgb_fwd" + shape(N, Ci, Co, Ho, R, 1), f, f)

        def rgb_wgrad(a):
            N, H, W, Ho, Wo, Co, R, S = a[7:15]
            f = 2.0 * N * Ho * Wo * Co * 3 * R * S
            return ("rgb_wgrad_act_bias" + shape(N, 3, Co, Ho, R, 1), f, f)

        self._wrap("migan_rgb_conv_fwd", rgb_fwd)
        self._wrap("migan_rgb_conv_wgrad", rgb_wgrad)
        self._wrap("migan_conv2d_dgrad_reflect1_ws", reflect1)
        self._wrap("migan_thin_toeplitz_fwd", toep("fwd", 6))
        self._wrap("migan_thin_toeplitz_wgrad", toep("wgrad", 5))
        self._wrap("migan_thin_toeplitz_dgrad", toep("dgrad", 5))
        self._wrap("migan_conv2d_dgrad_reflect1", reflect1)
        self._wrap("migan_conv2d_fwd", fwd)
        self._wrap("migan_conv2d_dropout_fwd", lambda a: fwd(a[:3] + a[4:]))
        self._wrap("migan_conv2d_fwd_ws", lambda a: fwd(a[:3] + a[4:]))   # (x, w, bias, mask, y, N, ...): geometry one slot later
        self._wrap("migan_conv2d_dgrad", dgrad)
        self._wrap("migan_conv2d_dgrad_ws", dgrad)

        # HBM-bound kernels (bytes = 4 * (elements read + written) of the layer's activation): the normalisation
        # layers.  stats = one read; apply = read + write; backward = statistics pass (x, dy) + apply pass (x, dy -> dx)
        def norm(kind, off, passes):
            def describe(a):
                G, P, C = a[off:off + 3]
                return ("norm_%s[%dx%dx%d]" % (kind, G, P, C), 4.0 * G * P * C * passes, -1.0)
            return describe

        self._wrap("migan_norm_stats", norm("stats", 8, 1))
        self._wrap("migan_norm_apply", norm("apply", 7, 2))
        self._wrap("migan_norm_apply_prelu", norm("apply", 8, 2))
        self._wrap("migan_norm_bwd", norm("bwd", 9, 5))
        self._wrap("migan_norm_bwd_prelu", norm("bwd", 11, 5))
        # the generator tail of dcgan.py:60-62 (BatchNorm read by the image-output conv): the forward is the thin-N conv launch; the backward
        # (csrc/norm.hip bn_conv1_bwd_*) is HBM-bound - x read by both walks, dx written: 12 B per element
        self._wrap("migan_conv2d_fwd_normed", lambda a: ("fwd_normed" + fwd(a)[0][3:],) + fwd(a)[1:])

        def bn_conv1_bwd(a):
            N, H, W, C = a[20:24]
            return ("bn_conv1_bwd[1x%dx%d]" % (N * H * W, C), 4.0 * N * H * W * C * 3, -1.0)

        self._wrap("migan_bn_conv1_bwd", bn_conv1_bwd)

        # weight-stationary Conv2d(64, 64, 3, 1, 1) (csrc/conv_c64.hip): forward and (flipped pack) input gradient are the same entry
        def c64(kind, off):
            def describe(a):
                N, H, W = a[off:off + 3]
                f = 2.0 * N * H * W * 64 * 64 * 9
                return ("c64_%s[%dx 64->64 k3 @%d]" % (kind, N, H), f, f)
            return describe

        self._wrap("migan_c64_conv_fwd", c64("conv", 4))
        self._wrap("migan_c64_conv_wgrad", c64("wgrad", 5))
        self._wrap("migan_conv2d_wgrad", wgrad)
        self._wrap("migan_upconv3x3_fwd", up_fwd)
        self._wrap("migan_upconv3x3_dgrad", up_dgrad)
        self._wrap("migan_upconv3x3_wgrad", up_wgrad)
        return self

    def __exit__(self, *exc):
        for name, fn in self.orig.items():
            setattr(self.lib, name, fn)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        self.hbm = {}
        for (sym, dense, execd), e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            if execd < 0:  # HBM-bound group: `dense` holds the algorithmic bytes
                d = self.hbm.setdefault(sym, {"launches": 0, "ms": 0.0, "bytes": 0.0})
                d["launches"] += 1
                d["ms"] += ms
                d["bytes"] += dense
                continue
            d = agg.setdefault(sym, {"launches": 0, "ms": 0.0, "dense": 0.0, "exec": 0.0})
            d["launches"] += 1
            d["ms"] += ms
            d["dense"] += dense
            d["exec"] += execd
        return agg


def pmc_table():
    """Committed rocprofv3 --pmc results for this build (tools/collect_profiles.py): HBM-side bytes per launch
    (FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE, separate passes) and MFMA-busy cycles,
    keyed by the roofline group names used here.  The PMC passes cannot run inside bench.py (counter collection
    serialises the launches), so they are taken on `bench.py --no-graph` by tools/round_measure.sh and committed."""
    for rnd in (PROFILE_ROUND, "r05"):   # the newest committed pass; the previous round's until this round's has been taken
        path = os.path.join(ROOT, "profiles", "%s_pmc_kernels.json" % rnd)
        try:
            with open(path) as f:
                return json.load(f), os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return {}, None


def lib_digest():
    """Digest of the kernel sources + flags the loaded libmigan.so was built from (csrc/build.py writes it next to the library)."""
    try:
        with open(os.path.join(ROOT, "pytorch-gan_amd", "csrc", ".libmigan.stamp")) as f:
            return f.read().strip()
    except OSError:
        return None


def pmc_stale(tab):
    """True when the committed counter table was NOT taken on the library this process runs (or does not say which library it was
    taken on): its traffic / MFMA-busy columns then describe other device code than the one being timed."""
    src = tab.get("_source") if isinstance(tab.get("_source"), dict) else {}
    have, mine = src.get("lib_digests") or ([src["lib_digest"]] if src.get("lib_digest") else []), lib_digest()
    return not (mine is not None and have == [mine])


def roofline(w, rank, nprof, segments=None):
    """Eager runs of the timed step with HIP events (on the launch stream) around every conv-family launch; every rank
    runs the steps (collectives), rank 0 records."""
    from pytorch_gan_amd import steps as _steps

    agg, hbm = {}, {}
    # per-launch times are a property of a kernel alone on the chip: the step bodies' second stream (the discriminator update underneath the
    # generator's backward, steps._fork_join) and the weight-gradient stream (functional._Fork) are off while the launches are timed -
    # the events sit on ONE launch stream
    from pytorch_gan_amd import functional as _F

    overlap, _steps._OVERLAP_D = _steps._OVERLAP_D, False
    wstream, _F._WGRAD_STREAM = _F._WGRAD_STREAM, False
    try:
        with (ConvProfiler(segments) if rank == 0 else contextlib.nullcontext()) as prof:
            for i in range(nprof):
                w.state.dp.begin_step()
                w.eager()
                w.state.dp.end_step()
            torch.cuda.synchronize()
            if rank == 0:
                agg = prof.summary()
                hbm = prof.hbm
    finally:
        _steps._OVERLAP_D, _F._WGRAD_STREAM = overlap, wstream
    if not agg:
        return None
    dom = max(agg, key=lambda k: agg[k]["ms"])
    d = agg[dom]
    sec = d["ms"] * 1e-3
    ach = d["exec"] / sec / 1e12
    dense = d["dense"] / sec / 1e12
    tab, src = pmc_table()
    ent = tab.get(dom, {})
    out = {
        "bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(ach / PEAK_TFLOPS, 4), "achieved_dense": round(dense, 2), "frac_dense": round(dense / PEAK_TFLOPS, 4),
        "traffic": ent.get("hbm_bytes_per_launch"), "traffic_source": src if ent else None,
        "counters_stale": pmc_stale(tab) if ent else None, "lib_digest": (lib_digest() or "")[:16],
        "symbol": ent.get("symbol"), "mfma_busy_frac": ent.get("mfma_busy_frac"),
        "avg_launch_ms": round(d["ms"] / d["launches"], 4), "launches_per_step": d["launches"] // nprof,
        "executed_gflop_per_launch": round(d["exec"] / d["launches"] / 1e9, 3),
        "dense_gflop_per_launch": round(d["dense"] / d["launches"] / 1e9, 3),
        "note": "achieved/frac = FLOPs the kernel EXECUTES / launch time (HIP events on the launch stream, eager run of "
                "the timed step; a wgrad launch = main kernel + split-K reduction); *_dense = the reference's dense "
                "2*M*N*K (SURVEY 8d), above 1 where the phase-collapsed Upsample+Conv3x3 kernels skip 20/36 of it",
        "conv_kernels": {k: {"ms_per_step": round(v["ms"] / nprof, 4), "executed_tflops": round(v["exec"] / (v["ms"] * 1e-3) / 1e12, 2),
                             "frac": round(v["exec"] / (v["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS, 3),
                             "launches_per_step": v["launches"] // nprof}
                         for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:24]},
        "conv_ms_per_step": round(sum(v["ms"] for v in agg.values()) / nprof, 4),
    }
    if hbm:
        # the HBM side of the roofline (BASELINE metric: "HBM GB/s ... vs the chip's roofline"): the normalisation call that takes
        # the most time; algorithmic bytes = 4 * (elements read + written) / time of the call's launches
        hk = max(hbm, key=lambda k: hbm[k]["ms"])
        h = hbm[hk]
        gbs = h["bytes"] / (h["ms"] * 1e-3) / 1e9
        hent = tab.get(hk, {})   # the same call under the PMC passes (tools/pmc_step.py): counted HBM-side bytes per call
        out["hbm"] = {
            "bound": "hbm", "kernel": hk, "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
            "frac_of_achievable_6290": round(gbs / 6290.0, 4), "traffic": hent.get("hbm_bytes_per_launch"),
            "traffic_source": src if hent else None, "counters_stale": pmc_stale(tab) if hent else None,
            "avg_call_ms": round(h["ms"] / h["launches"], 4), "calls_per_step": h["launches"] // nprof,
            "algorithmic_mb_per_call": round(h["bytes"] / h["launches"] / 1e6, 2),
            "note": "one call = the launches of that C entry point (stats: partial + finalize; bwd: partial + finalize + apply)",
            "norm_calls": {k: {"ms_per_step": round(v["ms"] / nprof, 4), "gb_s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                               "calls_per_step": v["launches"] // nprof}
                           for k, v in sorted(hbm.items(), key=lambda kv: -kv[1]["ms"])[:16]},
            "norm_ms_per_step": round(sum(v["ms"] for v in hbm.values()) / nprof, 4),
        }
    return out


def cpu_baseline(init, seconds_budget=25.0):
    """The oracle (CPU restatement of dcgan.py:143-183 on stock torch) timed on this box's host cores,
    starting from the same initial weights as the GPU run."""
    from oracle import reference_models as M
    from oracle import reference_steps as S

    G, D = M.DcganGenerator(IMG, LATENT, CH), M.DcganDiscriminator(IMG, CH)
    G.load_state_dict(init[0])
    D.load_state_dict(init[1])
    s = S.SimpleNamespace(G=G, D=D, opt_G=S._adam(G.parameters()), opt_D=S._adam(D.parameters()),
                          bce=torch.nn.BCELoss(), latent_dim=LATENT)
    threads = torch.get_num_threads()
    torch.manual_seed(1)
    np.random.seed(1)
    imgs = torch.rand(BATCH, CH, IMG, IMG) * 2 - 1
    S.dcgan_step(s, imgs)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        S.dcgan_step(s, imgs)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 20:
            break
    return {"value": round(BATCH * n / el, 3), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "%d full dcgan steps (bs %d, %dx%d) after 1 warm-up, %.1f s, torch CPU fp32, %d threads"
                      % (n, BATCH, IMG, IMG, el, threads)}


def cpu_baseline_extra(name, seconds_budget=12.0):
    """The oracle's step of one of the other BASELINE configs on this box's host cores, on a BOUNDED sample: a reduced batch where a
    full-batch step alone would take minutes (stated in `sample`); images/s scales with the batch on a CPU, so the figure is
    comparable per image.  A reported baseline, not a target."""
    import random as _random

    from oracle import reference_steps as S

    torch.manual_seed(1)
    np.random.seed(1)
    _random.seed(1)
    threads = torch.get_num_threads()
    if name == "cyclegan":
        bs, s = 1, S.make_cyclegan((3, 256, 256), 9)
        a, b = torch.rand(bs, 3, 256, 256) * 2 - 1, torch.rand(bs, 3, 256, 256) * 2 - 1
        step, what, warm = (lambda: S.cyclegan_step(s, a, b)), "cyclegan 256x256, 9 blocks, batch 1 of the config's 8", 0
    elif name == "srgan":
        bs, s = 2, S.make_srgan((384, 384), 16)
        lr, hr = torch.randn(bs, 3, 96, 96), torch.randn(bs, 3, 384, 384)
        step, what, warm = (lambda: S.srgan_step(s, lr, hr)), "srgan 96->384 incl. VGG19[:18], batch 2 of the config's 16", 0
    elif name == "pix2pix":
        bs, s = 1, S.make_pix2pix(256)
        a, b = torch.rand(bs, 3, 256, 256) * 2 - 1, torch.rand(bs, 3, 256, 256) * 2 - 1
        step, what, warm = (lambda: S.pix2pix_step(s, a, b)), "pix2pix 256x256 batch 1 (the config's)", 1
    elif name == "wgan_gp":
        bs, s = 64, S.make_wgan_gp(32)
        real = torch.rand(bs, 1, 32, 32) * 2 - 1
        it = [0]

        def step():
            it[0] += 1
            return S.wgan_gp_step(s, real, it[0])

        what, warm = "wgan_gp 32x32 batch 64 critic iterations (generator update every 5th)", 5
    else:
        return None
    for _ in range(warm):
        step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 200:
            break
    return {"value": round(bs * n / el, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "%d oracle step(s) of %s after %d warm-up, %.1f s, torch CPU fp32, %d threads" % (n, what, warm, el, threads)}


def _safe_cpu_baseline_extra(name):
    try:
        return cpu_baseline_extra(name)
    except Exception as ex:  # noqa: BLE001
        return {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}


def _safe_cpu_baseline(init):
    try:
        return cpu_baseline(init)
    except Exception as ex:  # reported baseline only: never at the price of the measured line
        return {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}


def run_strong(other, global_batch, k, wu, dp, rank, dev, args, world):
    """N > 1: `other` at a FIXED global batch sharded over the ranks (strong scaling) beside the headline's weak-scaling line, so that one
    `bench.py --gpus N` launch yields both figures.  Every rank runs it (the step contains the bucket all-reduces); rank 0 reports."""
    w = None
    try:
        if global_batch % world:
            return {"skipped": "global batch %d is not divisible by %d ranks" % (global_batch, world)}
        ns = argparse.Namespace(**vars(args))
        ns.batch = global_batch // world
        w = BUILDERS[other](dp, rank, dev, ns, wu + k)
        blocks, out = timed_blocks(w, world, dev, k, wu, 1.0, max_blocks=6)
        summ = summarise(other, w.batch, world, k, blocks)
        return {"images_per_s": summ["images_per_s"], "ms_per_step": summ["ms_per_step"], "scaling": "strong", "global_batch": global_batch,
                "per_gpu_batch": w.batch, "n_gpus": world, "workload": WORKLOAD_NAME[other], "steps": k, "warmup": wu, "hipgraph": w.graphed,
                "blocks": summ["blocks"], "losses": {kk: float(v) for kk, v in out.items() if "loss" in kk}}
    except Exception as ex:  # noqa: BLE001 - deterministic failures (shapes, memory) hit every rank alike
        return {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
    finally:
        del w
        torch.cuda.empty_cache()


def run_extra(other, k, wu, dp, rank, dev, args, batch=0, eager_too=False):
    """One of the other BASELINE configs, briefly (headline section only), after the headline workload has been released.
    `batch`: per-GPU batch other than the config's (cyclegan at 1 image per GPU = the shard of the 8-GPU configuration);
    `eager_too`: also time the same step launched one kernel at a time, and count its launches."""
    w = None
    try:
        ns = argparse.Namespace(**vars(args))
        ns.batch = batch
        w = BUILDERS[other](dp, rank, dev, ns, wu + k)
        torch.cuda.reset_peak_memory_stats()
        blocks, out = timed_blocks(w, 1, dev, k, wu, 2.0, max_blocks=20)
        summ = summarise(other, w.batch, 1, k, blocks)
        res = {"images_per_s": summ["images_per_s"], "ms_per_step": summ["ms_per_step"], "ms_per_step_min": summ["ms_per_step_min"],
               "ms_per_step_max": summ["ms_per_step_max"], "blocks": summ["blocks"], "timed_seconds": summ["timed_seconds"],
               "step_executed_frac": summ["step_executed_frac"], "step_dense_frac": summ["step_dense_frac"],
               "workload": WORKLOAD_NAME[other] if not batch else "%s - at batch %d per GPU" % (WORKLOAD_NAME[other], batch),
               "steps": k, "warmup": wu, "hipgraph": w.graphed,
               "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
               "losses": {kk: float(v) for kk, v in out.items() if "loss" in kk}}
        if w.capture_error:
            res["hipgraph_error"] = w.capture_error[:200]
        if eager_too:
            from pytorch_gan_amd._lib import lib

            def eager_step(i):
                dp.begin_step()
                o = w.eager()
                dp.end_step()
                return o

            eager_step(0)
            torch.cuda.synchronize()
            n0 = lib.migan_debug_launch_count(None)
            eager_step(1)
            res["library_launches_per_step"] = int(lib.migan_debug_launch_count(None) - n0)
            we = Workload(other, w.batch, eager_step, w.state)
            eb, _ = timed_blocks(we, 1, dev, k, 1, 1.0, max_blocks=10)
            res["eager_ms_per_step"] = round(1e3 * float(np.median(eb)) / k, 4)
        return res
    except Exception as ex:  # noqa: BLE001 - the headline line must survive a failure of an extra
        return {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
    finally:
        del w
        torch.cuda.empty_cache()


def replicas_identical(w, world, dev):
    """Same initial weights + summed gradients -> same updates on every rank."""
    chk = torch.stack([torch.cat([p.detach().double().flatten() for p in m.parameters()]).abs().sum()
                       for m in w.nets]).to(dev)
    allc = [torch.zeros_like(chk) for _ in range(world)]
    torch.distributed.all_gather(allc, chk)
    if not all(torch.equal(allc[0], c) for c in allc):
        raise SystemExit("data-parallel replicas diverged: %s" % [c.tolist() for c in allc])
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="dcgan", choices=sorted(BUILDERS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the BASELINE config's)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: total batch, sharded over the ranks (must be divisible by --gpus)")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the K-step block until this much is timed")
    ap.add_argument("--max-blocks", type=int, default=200, help="upper bound on the number of timed K-step blocks")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of hipGraph replay")
    ap.add_argument("--graph-always", action="store_true", help="cyclegan: record the step at every batch size (default: batch <= 2 only)")
    ap.add_argument("--pmc-log", default="", help="target mode of the rocprofv3 --pmc passes: run --steps eager steps of the workload "
                    "with the per-launch accounting of `roofline`, write which library launches (by ordinal) belong to which "
                    "roofline group to this file (tools/pmc_step.py joins it with the pass's counter CSV) and exit")
    ap.add_argument("--no-overlap", action="store_true", help="A/B: one stream - the discriminator update after the generator's backward "
                                                             "(reference order) and every weight gradient in line with its layer's backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the brief runs of the other BASELINE configs (N=1)")
    ap.add_argument("--dp-order", default="", choices=["", "sequential", "fork"],
                    help="N>1: 'sequential' (default) = the reference's order, the generator bucket's all-reduce + Adam run on the side "
                         "stream UNDER the discriminator phase; 'fork' = the single-GPU body (discriminator update underneath the "
                         "generator's backward, exchanges behind the join) - A/B on the 8-GPU node")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N>1: BatchNorm statistics over the global batch (the reference's single-process semantics)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU
        # (the contract's launch line); rank 0's JSON line is this process's output, its status our exit status
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    from pytorch_gan_amd import dp as dpmod

    if os.environ.get("MIGAN_HANG_DUMP_S"):
        # diagnosis aid: after S seconds every rank prints the Python stack of each of its threads to stderr and carries on
        # (a launch that is still alive then is stuck or far too slow - the stacks say where)
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["MIGAN_HANG_DUMP_S"]), repeat=False, file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dp = dpmod.init_from_env()
    world = dp.world
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d (launch with torch.distributed.run)" % (args.gpus, world))
    rank = dp.rank
    dev = torch.device("cuda", torch.cuda.current_device())
    scaling = "weak"
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch %d is not divisible by %d ranks" % (args.global_batch, world))
        args.batch = args.global_batch // world
        scaling = "strong"
    if args.sync_bn and world > 1:
        dp.enable_sync_batchnorm()

    name = args.workload
    if args.dp_order:
        from pytorch_gan_amd import steps as _steps

        _steps.set_dp_order(args.dp_order)
    if args.no_overlap:
        from pytorch_gan_amd import functional as _F
        from pytorch_gan_amd import steps as _steps

        _steps._OVERLAP_D = False
        _F._WGRAD_STREAM = False
    if args.pmc_log:
        args.no_graph = True
    w = BUILDERS[name](dp, rank, dev, args, args.warmup + args.steps)
    if args.pmc_log:
        from pytorch_gan_amd._lib import lib

        for i in range(args.warmup):
            w.run(i)
        segs = []
        rf = roofline(w, rank, args.steps, segs)
        torch.cuda.synchronize()
        with open(args.pmc_log, "w") as fh:
            json.dump({"workload": name, "steps": args.steps, "total_launches": lib.migan_debug_launch_count(None), "segments": segs,
                       "conv_kernels": rf.get("conv_kernels"), "norm_calls": (rf.get("hbm") or {}).get("norm_calls")}, fh)
        print(json.dumps({"pmc_log": args.pmc_log, "segments": len(segs), "total_launches": lib.migan_debug_launch_count(None)}))
        return
    blocks, out = timed_blocks(w, world, dev, args.steps, args.warmup, args.min_seconds, max_blocks=args.max_blocks)
    exchange = None
    if world > 1 and hasattr(dp, "start_timing"):
        # what the exchange costs and how much of it the overlap hides: HIP events around every bucket all-reduce (side stream) and
        # around every wait of the main stream for an update in flight, over one more block of --steps steps (not the timed ones)
        dp.start_timing()
        for i in range(args.steps):
            w.run(i)
        exchange = dp.timing_report(args.steps)
    losses = {k: float(v) for k, v in out.items() if "loss" in k}
    if not all(np.isfinite(v) for v in losses.values()):
        raise SystemExit("non-finite loss in the timed region: %s" % losses)
    summ = summarise(name, w.batch, world, args.steps, blocks)
    result = {
        "metric": "training images/sec", "value": round(summ["images_per_s"], 2), "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": summ["ms_per_step"], "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOAD_NAME[name], "global_batch": world * w.batch, "parallelism": "dp%d" % world,
                   "hipgraph": w.graphed, "gflop_per_image": round(GFLOP_PER_IMG[name], 4),
                   "executed_gflop_per_image": round(executed_gflop_per_image(name), 4),
                   "sync_batchnorm": bool(args.sync_bn and world > 1),
                   "streams": ("one" if args.no_overlap or (args.sync_bn and world > 1) or name in ("wgan_gp", "esrgan") else
                               "discriminator update on a second HIP stream underneath the generator's backward"
                               if world == 1 or args.dp_order == "fork" or os.environ.get("MIGAN_DP_ORDER") == "fork" else
                               "reference order: generator exchange + Adam on the data-parallel side stream under the discriminator phase")},
        "timing": {"rule": "median over blocks of exactly --steps steps, each bracketed by barrier+synchronize, MAX over ranks",
                   "blocks": summ["blocks"], "timed_seconds": summ["timed_seconds"],
                   "ms_per_step_min": summ["ms_per_step_min"], "ms_per_step_max": summ["ms_per_step_max"],
                   "block_ms_per_step": [round(1e3 * b / args.steps, 4) for b in blocks[:64]]},
        "step_executed_frac": summ["step_executed_frac"], "step_dense_frac": summ["step_dense_frac"],
        "losses": losses,
        "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
    }
    if w.capture_error:
        result["config"]["hipgraph_error"] = w.capture_error[:200]
    if not args.no_roofline:
        try:
            rf = roofline(w, rank, 5 if name in ("dcgan", "wgan_gp", "pix2pix") else 2)
        except Exception as ex:  # the headline line must survive a failure of the per-kernel accounting
            if world > 1:
                raise          # ranks must stay in lock step (the eager profiling steps contain collectives)
            rf = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
        if rf:
            if isinstance(rf, dict) and "hbm" in rf:
                result["roofline_hbm"] = rf.pop("hbm")
            result["roofline"] = rf
    if world > 1:
        from pytorch_gan_amd import steps as _steps

        result["config"]["replicas_identical"] = replicas_identical(w, world, dev)
        result["config"]["rccl_ranks"] = world
        result["config"]["backend"] = torch.distributed.get_backend()
        result["config"]["dp_order"] = _steps._DP_ORDER
        result["config"]["bucket_bytes"] = {k: int(getattr(w.state, k).flat_grad.numel() * 4) for k in sorted(vars(w.state))
                                            if k.startswith("opt_") and hasattr(getattr(w.state, k), "flat_grad")}
        result["config"]["batch_slicing"] = "one globally seeded draw of the global batch, sliced by rank"
        if exchange:
            result["exchange"] = exchange
        if name == "dcgan" and not args.no_extra and not args.global_batch:
            # BASELINE.json configs[3] as written - CycleGAN 256x256 at a GLOBAL batch of 8 over the N GPUs (one image per GPU at N = 8) - next
            # to the weak-scaling headline: strong scaling from the same launch
            del w, out
            torch.cuda.empty_cache()
            result["extra"] = {"cyclegan_global_batch_8": run_strong("cyclegan", 8, 4, 2, dp, rank, dev, args, world)}
    if rank == 0 and world == 1 and name == "dcgan" and not args.no_extra:
        # the other GPU configs of BASELINE.json, briefly (north_star names CycleGAN 256x256 bs 8 as the second target)
        init = w.init
        del w, out
        torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = _safe_cpu_baseline(init)
        extra = result["extra"] = {}
        for other, k, wu in (("cyclegan", 4, 3), ("srgan", 4, 3), ("wgan_gp", 100, 10), ("dcgan_ch3", 50, 5), ("pix2pix", 50, 5)):
            extra[other] = run_extra(other, k, wu, dp, rank, dev, args)
            if not args.no_cpu_baseline and other != "dcgan_ch3" and "error" not in extra[other]:
                extra[other]["cpu_baseline"] = _safe_cpu_baseline_extra(other)
        # config 4's per-GPU shard at N = 8: one image per GPU (= the reference's default batch, cyclegan.py:28) - the recorded step
        # beside the same step launched kernel by kernel
        extra["cyclegan_bs1"] = run_extra("cyclegan", 10, 3, dp, rank, dev, args, batch=1, eager_too=True)
        # config 5's per-GPU shard at N = 8: two images per GPU (row N3; with cross-replica BatchNorm at N > 1: --sync-bn)
        extra["srgan_bs2"] = run_extra("srgan", 10, 3, dp, rank, dev, args, batch=2)
    elif rank == 0 and world == 1 and name == "dcgan" and not args.no_cpu_baseline:
        result["cpu_baseline"] = _safe_cpu_baseline(w.init)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
