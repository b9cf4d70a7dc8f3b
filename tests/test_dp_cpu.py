"""Data-parallel logic on the CPU (gloo, world_size 2 and 4): sharding + ONE summed all-reduce of the flat gradient
bucket per optimiser + 1/world scaling in the update reproduces the single-process full-batch step
(SURVEY.md §8e).  The HIP optimiser's update kernel needs a GPU, so a CPU optimiser with the SAME bucket
(`pytorch_gan_amd.optim.bucket_layout`: 256-byte aligned slots, padding included in the all-reduce), the same interface
(`flat_grad`, `step(grad_scale)`) and torch.optim.Adam's arithmetic drives `pytorch_gan_amd.dp.DataParallel` here and is
compared with stock `torch.optim.Adam` on the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FlatAdam:
    """optim.Adam's contract on the CPU: every p.grad is a view into ONE flat buffer laid out by optim.bucket_layout, step()
    applies grad_scale (the 1/world of the data-parallel mean) and torch.optim.Adam's update (`_single_tensor_adam`)."""

    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8):
        from pytorch_gan_amd.optim import bucket_layout

        self.params = list(params)
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0
        self.offsets, total = bucket_layout([p.numel() for p in self.params])
        assert total % 64 == 0 and total >= sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(total)
        self.exp_avg, self.exp_avg_sq = torch.zeros(total), torch.zeros(total)
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def zero_grad(self):
        self.flat_grad.zero_()

    def step(self, grad_scale=1.0):
        self.t += 1
        b1, b2 = self.betas
        with torch.no_grad():
            g = self.flat_grad * grad_scale
            self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
            upd = (self.exp_avg / bc1) / ((self.exp_avg_sq.sqrt() / bc2 ** 0.5) + self.eps)
            for p, o in zip(self.params, self.offsets):
                p.add_(upd[o:o + p.numel()].view_as(p), alpha=-self.lr)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle import reference_models as M
    from pytorch_gan_amd import dp as dpmod

    dp = dpmod.init_from_env(backend="gloo")
    assert isinstance(dp, dpmod.DataParallel) and dp.world == world and dp.rank == rank
    torch.manual_seed(0)
    D = M.MlpCritic((1, 8, 8))          # no BatchNorm: shards are exactly independent samples
    G = M.CycleDiscriminator((3, 32, 32))  # InstanceNorm: per-sample statistics, also exactly shardable
    if rank >= 1:  # replicas start from rank 0's weights
        with torch.no_grad():
            for p in list(D.parameters()) + list(G.parameters()):
                p.add_(float(rank))
    dp.broadcast_parameters(D, G)
    opt_D, opt_G = FlatAdam(D.parameters()), FlatAdam(G.parameters())
    g = torch.Generator().manual_seed(5)
    xb = torch.rand(8, 1, 8, 8, generator=g)       # global batches, identical on every rank (host RNG is shared)
    yb = torch.rand(4, 3, 32, 32, generator=g)
    # cross-replica BatchNorm protocol (host side): gather of the per-rank moments + Chan combination == moments of the
    # whole batch; all-reduced backward sums == sums over the whole batch
    sync = dp.enable_sync_batchnorm()
    assert dp.graph_ok is True and dp.sync_bn is sync   # recorded steps are cut at the BatchNorm collectives (graph.StepRunner)
    feat = torch.rand(8, 6, generator=torch.Generator().manual_seed(11)) * 3 + 5
    mine = dp.shard(feat)
    mom = torch.cat([mine.mean(0), mine.var(0, unbiased=False)])
    allm = sync.all_gather(mom).view(world, 2, 6).double()
    m = allm[:, 0].mean(0)
    var = (allm[:, 1] + (allm[:, 0] - m) ** 2).mean(0)   # equal shards: what migan_norm_sync_finalize computes
    assert torch.allclose(m, feat.double().mean(0), atol=1e-6) and torch.allclose(var, feat.double().var(0, unbiased=False), atol=1e-6)
    sums = sync.all_reduce_sum(mine.sum(0).clone())
    assert torch.allclose(sums, feat.sum(0), atol=1e-5)
    dp.disable_sync_batchnorm()
    assert dp.sync_bn is None
    dp.begin_step()
    opt_D.zero_grad()
    (-torch.mean(D(dp.shard(xb)))).backward()      # every loss on the path is a batch mean
    dp.step(opt_D)
    opt_G.zero_grad()
    torch.nn.functional.mse_loss(G(dp.shard(yb)), torch.ones(4 // world, 1, 2, 2)).backward()
    dp.step(opt_G)
    dp.end_step()
    state = {k: v.clone() for k, v in list(D.state_dict().items()) + [("G." + k, v) for k, v in G.state_dict().items()]}
    if rank == 0:
        torch.save(state, out)
    # every rank must hold identical parameters after the update
    flat = torch.cat([v.flatten() for v in state.values()])
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(flat, ref)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_step_equals_full_batch(tmp_path, world):
    sys.path.insert(0, ROOT)
    from oracle import reference_models as M

    out = str(tmp_path / "dp_state.pt")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out)
    # single-process reference on the full batch
    torch.manual_seed(0)
    D = M.MlpCritic((1, 8, 8))
    G = M.CycleDiscriminator((3, 32, 32))
    opt_D = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))   # the single-process reference optimiser
    opt_G = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(5)
    xb = torch.rand(8, 1, 8, 8, generator=g)
    yb = torch.rand(4, 3, 32, 32, generator=g)
    opt_D.zero_grad()
    (-torch.mean(D(xb))).backward()
    opt_D.step()
    opt_G.zero_grad()
    torch.nn.functional.mse_loss(G(yb), torch.ones(4, 1, 2, 2)).backward()
    opt_G.step()
    want = {k: v for k, v in list(D.state_dict().items()) + [("G." + k, v) for k, v in G.state_dict().items()]}
    gmax = {k: float(p.grad.abs().max()) for k, p in list(D.named_parameters()) + [("G." + k, p) for k, p in G.named_parameters()]}
    assert got.keys() == want.keys()
    strict = 0
    for k in want:
        if gmax.get(k, 1.0) < 1e-6:
            # a conv bias in front of InstanceNorm: the true gradient is exactly zero, what is left is rounding noise, and
            # Adam's first step turns its SIGN into a full +-lr step - the sharded and the full-batch run may differ by 2 lr
            assert float((got[k] - want[k]).abs().max()) <= 2.05 * 2e-4, k
        else:
            assert torch.allclose(got[k], want[k], rtol=1e-5, atol=2e-6), k
            strict += 1
    assert strict >= len(want) - 4


def test_shard_rejects_ragged_batch():
    sys.path.insert(0, ROOT)
    from pytorch_gan_amd import dp as dpmod

    class Fake(dpmod.DataParallel):
        def __init__(self):
            self.world, self.rank = 3, 1

    d = Fake()
    assert d.shard(torch.arange(6)).tolist() == [2, 3]
    try:
        d.shard(torch.arange(8))
    except ValueError:
        return
    raise AssertionError("ragged global batch must be rejected")


def test_local_stepper_is_default():
    sys.path.insert(0, ROOT)
    from pytorch_gan_amd import dp as dpmod

    os.environ.pop("WORLD_SIZE", None)
    s = dpmod.init_from_env()
    assert isinstance(s, dpmod.LocalStepper) and s.world == 1
    calls = []

    class Opt:
        def step(self):
            calls.append(1)

    s.begin_step()
    s.step(Opt())
    s.end_step()
    assert calls == [1]
    assert np.isfinite(1.0)
