"""Worker processes of tests/test_dp_gpu.py (one per rank, started with torch.distributed.run or directly).

    python tests/dp_worker.py nccl1 <out.json>     world_size 1, backend nccl (= RCCL): side-stream all-reduce + graph segments
    python tests/dp_worker.py syncbn <out.pt>      N ranks (gloo, all on cuda:0): cross-replica BatchNorm DCGAN step
                                                   MIGAN_TEST_STEPS=k: k steps;  MIGAN_TEST_GRAPH=1: step 1 eager, the rest replays of
                                                   the recorded step (hipGraph segments cut at every BatchNorm collective);
                                                   MIGAN_TEST_GRAPH=cuts: the recording protocol without a capture (CPU)
    python tests/dp_worker.py srgan <out.pt>       N ranks (gloo, all on cuda:0): one SRGAN step, the batch sharded, cross-replica BatchNorm
    python tests/dp_worker.py order <out.pt>       N ranks (gloo, all on cuda:0): CycleGAN steps, one image pair per rank, under
                                                   MIGAN_TEST_ORDER=sequential|fork (steps.set_dp_order)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def _seed(s):
    torch.manual_seed(s)
    np.random.seed(s)


def nccl1(out):
    """K12 with world size 1: the RCCL process group, the side-stream all-reduce + fused Adam, and the segmented hipGraph
    replay all execute; results must equal the LocalStepper (no collective) run."""
    import pytorch_gan_amd as pg
    from oracle import reference_steps as S
    from pytorch_gan_amd import dp as dpmod
    from pytorch_gan_amd import graph, steps
    from util import gpu_copy

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dp = dpmod.DataParallel()
    assert dp.world == 1 and dp.side is not None and dist.get_backend() == "nccl"
    _seed(0)
    base = S.make_dcgan(32)
    for m in base.D.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    s_dp = steps.make_gan_state(gpu_copy(base.G), gpu_copy(base.D), dp=dp)
    s_lo = steps.make_gan_state(gpu_copy(base.G), gpu_copy(base.D))
    _seed(2)
    imgs = (torch.rand(8, 1, 32, 32) * 2 - 1).cuda()
    zs = torch.randn(10, 8, 100).cuda()
    res = {"eager": [], "graph": []}
    for i in range(3):   # eager: bucket all-reduce + Adam on the side stream, begin_step/wait ordering
        a = steps.dcgan_step(s_dp, imgs, zs[i])
        b = steps.dcgan_step(s_lo, imgs, zs[i])
        torch.cuda.synchronize()
        res["eager"].append([float(a["g_loss"]), float(b["g_loss"]), float(a["d_loss"]), float(b["d_loss"])])
    z_static = zs[3].clone()
    runner = graph.StepRunner(lambda: steps.dcgan_step(s_dp, imgs, z_static), dp, use_graph=True, warmup=1).prepare()
    steps.dcgan_step(s_lo, imgs, zs[3])  # mirror the runner's warm-up step
    res["graphed"], res["segments"] = runner.graphed, len(runner._segments or [])
    res["capture_error"] = runner.capture_error
    for i in range(4, 8):
        z_static.copy_(zs[i])
        a = runner.run()
        b = steps.dcgan_step(s_lo, imgs, zs[i])
        torch.cuda.synchronize()
        res["graph"].append([float(a["g_loss"]), float(b["g_loss"]), float(a["d_loss"]), float(b["d_loss"])])
    res["max_param_diff"] = max(float((p - q).abs().max()) for p, q in zip(s_dp.G.parameters(), s_lo.G.parameters()))
    dist.destroy_process_group()
    json.dump(res, open(out, "w"))
    del pg


def syncbn(out):
    """Every rank runs HALF of a DCGAN batch with cross-replica BatchNorm on; rank 0 stores losses, G/D gradients and the
    BatchNorm running statistics for comparison with the single-process full-batch step."""
    import pytorch_gan_amd as pg
    from oracle import reference_steps as S
    from pytorch_gan_amd import dp as dpmod
    from pytorch_gan_amd import steps
    from util import gpu_copy

    dp = dpmod.init_from_env()
    sync = os.environ.get("MIGAN_TEST_SYNCBN", "1") == "1"
    if sync:
        dp.enable_sync_batchnorm()
    _seed(0)
    base = S.make_dcgan(32)
    for m in base.D.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0   # device dropout streams would differ between the sharded and the full-batch run
    s = steps.make_gan_state(gpu_copy(base.G), gpu_copy(base.D), dp=dp)
    _seed(9)
    imgs = (torch.rand(16, 1, 32, 32) * 2 - 1).cuda()
    z = torch.randn(16, 100).cuda()
    nsteps, mode = int(os.environ.get("MIGAN_TEST_STEPS", "1")), os.environ.get("MIGAN_TEST_GRAPH", "0")
    xs, zs = dp.shard(imgs).clone(), dp.shard(z).clone()
    info = {}
    if mode == "1":
        from pytorch_gan_amd import graph

        runner = graph.StepRunner(lambda: steps.dcgan_step(s, xs, zs), dp, use_graph=True, warmup=1).prepare()
        for _ in range(nsteps - 1):
            o = runner.run()
        kinds = [k for k, _ in runner._segments or []]
        info = {"graphed": runner.graphed, "capture_error": runner.capture_error, "graphs": kinds.count("graph"),
                "eager": kinds.count("eager"), "cuts": dp.sync_bn.cuts if sync else 0}
    elif mode == "cuts":
        # what a recording does to the step, minus the capture: every BatchNorm collective and every dp.step() goes through
        # _Segmenter.cut() on the thread that runs the step (autograd worker threads off, as graph.StepRunner._capture sets it)
        import threading

        class Cuts:
            def __init__(self):
                self.items, self.threads = [], set()

            def cut(self, fn):
                self.threads.add(threading.get_ident())
                self.items.append(fn)
                fn()

        for _ in range(nsteps):
            seg = dp._segmenter = Cuts()
            if sync:
                dp.sync_bn.cuts = 0
            with torch.autograd.set_multithreading_enabled(False):
                o = steps.dcgan_step(s, xs, zs)
            dp._segmenter = None
            dp.end_step()
        info = {"items": len(seg.items), "cuts": dp.sync_bn.cuts if sync else 0,
                "own_thread": seg.threads == {threading.get_ident()}}
    else:
        for _ in range(nsteps):
            dp.begin_step()
            o = steps.dcgan_step(s, xs, zs)
    dp.end_step()
    torch.cuda.synchronize()
    # the logged loss of a rank is the mean over its shard: average over ranks = the full-batch mean
    losses = torch.stack([o["g_loss"], o["d_loss"]]).clone()
    dist.all_reduce(losses)
    losses /= dp.world
    if dp.rank == 0:
        torch.save({"losses": losses.cpu(), "info": info,
                    "G": {k: v.detach().cpu() for k, v in s.G.state_dict().items()},
                    "D": {k: v.detach().cpu() for k, v in s.D.state_dict().items()},
                    "gG": s.opt_G.flat_grad.cpu() / dp.world, "gD": s.opt_D.flat_grad.cpu() / dp.world}, out)
    dist.barrier()
    dist.destroy_process_group()
    del pg


def order(out):
    """Every rank runs ONE image pair of a two-pair CycleGAN batch (InstanceNorm: shards exactly) for MIGAN_TEST_STEPS steps under
    steps.set_dp_order(MIGAN_TEST_ORDER); rank 0 stores the rank-averaged losses of the last step, all four networks' weights and the
    three buckets' gradients (divided by world) of the last step."""
    import random

    import pytorch_gan_amd as pg
    from oracle import reference_steps as S
    from pytorch_gan_amd import dp as dpmod
    from pytorch_gan_amd import steps
    from util import gpu_copy

    dp = dpmod.init_from_env()
    steps.set_dp_order(os.environ.get("MIGAN_TEST_ORDER", "sequential"))
    _seed(0)
    side = int(os.environ.get("MIGAN_TEST_SIDE", "64"))
    shape = (3, side, side)
    base = S.make_cyclegan(shape, 1)
    s = steps.make_cyclegan_state(gpu_copy(base.G_AB), gpu_copy(base.G_BA), gpu_copy(base.D_A), gpu_copy(base.D_B), dp=dp)
    _seed(9)
    A = (torch.rand(2, *shape) * 2 - 1).cuda()
    B = (torch.rand(2, *shape) * 2 - 1).cuda()
    a, b = dp.shard(A).clone(), dp.shard(B).clone()
    nsteps = int(os.environ.get("MIGAN_TEST_STEPS", "1"))
    keys = ("loss_G", "loss_D", "loss_GAN", "loss_cycle", "loss_identity")
    first = None
    for t in range(nsteps):
        random.seed(5 + t)
        dp.begin_step()
        o = steps.cyclegan_step(s, a, b)
        dp.end_step()
        torch.cuda.synchronize()
        losses = torch.stack([o[k] for k in keys]).clone()
        dist.all_reduce(losses)
        losses /= dp.world
        if t == 0:   # the buckets hold the summed gradients of this step until the next zero_grad()
            first = {"losses": losses.cpu(), "grads": {n: getattr(s, n).flat_grad.cpu() / dp.world for n in ("opt_G", "opt_D_A", "opt_D_B")}}
    if dp.rank == 0:
        torch.save({"losses": losses.cpu(), "first": first,
                    "nets": {n: {k: v.detach().cpu() for k, v in getattr(s, n).state_dict().items()} for n in ("G_AB", "G_BA", "D_A", "D_B")},
                    "grads": {n: getattr(s, n).flat_grad.cpu() / dp.world for n in ("opt_G", "opt_D_A", "opt_D_B")}}, out)
    dist.barrier()
    dist.destroy_process_group()
    del pg


def srgan(out):
    """Row N3 (config 5's per-GPU shard, srgan.py:97-145): every rank runs its shard of an SRGAN batch with cross-replica BatchNorm on
    (generator BatchNorm2d(64, 0.8) + PReLU / PixelShuffle fused launches and the discriminator's BatchNorm layers all take the GLOBAL
    batch's statistics); rank 0 stores the rank-averaged losses, every parameter's gradient (divided by world) and the BatchNorm buffers.
    MIGAN_TEST_HR = high-resolution side (384), MIGAN_TEST_NRES = residual blocks (16), MIGAN_TEST_BATCH = GLOBAL batch (4)."""
    import pytorch_gan_amd as pg
    from oracle import reference_steps as S
    from pytorch_gan_amd import dp as dpmod
    from pytorch_gan_amd import steps
    from util import gpu_copy

    dp = dpmod.init_from_env()
    if os.environ.get("MIGAN_TEST_SYNCBN", "1") == "1":
        dp.enable_sync_batchnorm()
    hr, nres, batch = (int(os.environ.get(k, d)) for k, d in (("MIGAN_TEST_HR", "384"), ("MIGAN_TEST_NRES", "16"), ("MIGAN_TEST_BATCH", "4")))
    _seed(0)
    base = S.make_srgan((hr, hr), n_res=nres)
    s = steps.make_srgan_state(gpu_copy(base.G), gpu_copy(base.D), gpu_copy(base.V), dp=dp)
    _seed(12)
    lr_imgs, hr_imgs = torch.randn(batch, 3, hr // 4, hr // 4), torch.randn(batch, 3, hr, hr)
    dev = next(s.G.parameters()).device
    a, b = dp.shard(lr_imgs.to(dev)).clone(), dp.shard(hr_imgs.to(dev)).clone()
    dp.begin_step()
    o = steps.srgan_step(s, a, b)
    dp.end_step()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    keys = ("loss_G", "loss_D", "loss_content", "loss_GAN")
    losses = torch.stack([o[k] for k in keys]).clone()
    dist.all_reduce(losses)
    losses /= dp.world
    if dp.rank == 0:
        torch.save({"losses": losses.cpu(),
                    "grads": {n: {k: (p.grad.detach().cpu() / dp.world) for k, p in getattr(s, n).named_parameters() if p.grad is not None}
                              for n in ("G", "D")},
                    "buffers": {n: {k: v.detach().cpu() for k, v in getattr(s, n).named_buffers()} for n in ("G", "D")},
                    "weights": {n: {k: v.detach().cpu() for k, v in getattr(s, n).named_parameters()} for n in ("G", "D")}}, out)
    dist.barrier()
    dist.destroy_process_group()
    del pg


if __name__ == "__main__":
    if os.environ.get("MIGAN_TEST_EMU") == "1":
        # the same worker on the host execution model of the kernels (tests/hipemu): CPU tensors, gloo, no GPU
        import hipemu.host

        with hipemu.host.emulated_device():
            {"syncbn": syncbn, "order": order, "srgan": srgan}[sys.argv[1]](sys.argv[2])
    else:
        {"nccl1": nccl1, "syncbn": syncbn, "order": order, "srgan": srgan}[sys.argv[1]](sys.argv[2])
