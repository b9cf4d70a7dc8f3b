"""Single-node data parallelism for the GAN step: one process per GPU, RCCL over xGMI.

The reference is single-process; this is the MI355X-native scaling layer (§8e):
  * the minibatch is sharded over ranks (independent samples; every loss is a batch mean),
  * each optimiser owns ONE flat fp32 gradient bucket (optim.Adam.flat_grad), so the exchange is a single
    `all_reduce(SUM)` per network per step — G's after `loss_G.backward()`, D's after `loss_D.backward()`,
  * the all-reduce and the fused Adam update run on a side HIP stream, overlapping the next independent
    phase on the main stream (the D step consumes `gen.detach()` produced before the G update —
    dcgan.py:179, cyclegan.py:216-217 — so G's reduce+update overlaps D's forward/backward),
  * the 1/world_size averaging is folded into the Adam kernel (grad_scale).
BatchNorm layers use per-rank batch statistics by default (standard data-parallel behaviour); with
`enable_sync_batchnorm()` they use the statistics of the global batch, which restores the single-process
semantics of the reference for the BatchNorm models.  InstanceNorm models shard with no
semantic change either way.
"""
import os

import torch
import torch.distributed as dist


class LocalStepper:
    """world_size == 1: optimiser steps run in place on the current stream."""

    world = 1
    rank = 0
    segment = False  # graph.StepRunner: one graph for the whole step

    def begin_step(self):
        pass

    def step(self, opt):
        opt.step()

    def wait(self, opt=None):
        pass

    def end_step(self):
        pass


class DataParallel:
    def __init__(self, overlap=True, group=None):
        if not dist.is_initialized():
            raise RuntimeError("DataParallel needs torch.distributed to be initialised (see init_from_env)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # side-stream overlap whenever the buckets live on a GPU (RCCL in production; gloo stages CUDA tensors through
        # the host, which is how the 2-rank control flow is exercised on a single-GPU box)
        self.cuda = torch.cuda.is_available() and dist.get_backend(group) in ("nccl", "gloo")
        self.side = torch.cuda.Stream() if (self.cuda and overlap) else None
        self.segment = True     # graph.StepRunner cuts the captured step at every step(): the collective replays eagerly
        self.graph_ok = True    # graph.StepRunner may record the step (cross-replica BatchNorm cuts segments at its collectives)
        self.sync_bn = None
        self._pending = []      # optimisers with an update in flight on the side stream (event in opt.pending)
        self._segmenter = None  # set by graph.StepRunner while it records the step
        self.timing = None      # start_timing(): event pairs around every exchange (side stream) and every wait of the main stream

    # -- per-step protocol -------------------------------------------------------------------------
    # Ordering rule: `step(opt)` returns while the all-reduce + Adam of `opt` may still run on the side stream.  The
    # main stream has to wait for that update before it next READS those parameters (a forward/backward of that
    # network) or touches the bucket.  `opt.zero_grad()` / `opt.step()` wait by themselves; parameter reads are covered
    # by `wait(opt)` — the step bodies in steps.py call it where the reference reuses a network right after its update
    # (wgan_gp.py:179-186) — and by `begin_step()` at the top of every iteration.
    def begin_step(self):
        """Main stream must see every update launched by the previous step before parameters are reused."""
        if self._segmenter is None:  # while recording, graph.StepRunner.run() issues the step-boundary wait itself
            self._wait_now(None)

    def step(self, opt):
        """All-reduce opt.flat_grad (SUM) and apply the update with grads scaled by 1/world."""
        from . import functional as F

        F.join_wgrad_streams()   # before the segment is cut / the bucket is read: no weight-gradient stream is left forked
        if self._segmenter is not None:
            # hipGraph recording: close the current compute segment; the exchange + update replay eagerly.  The update
            # itself does not run while recording, but the recorded segments after it must not re-use weight packs made
            # before it (functional.weight_cache_scope): renew the epoch stamps as the real step would
            self._segmenter.cut(lambda: self._step_now(opt))
            opt.bump_epoch()
            return
        self._step_now(opt)

    def _step_now(self, opt):
        scale = 1.0 / self.world
        if self.side is None:
            dist.all_reduce(opt.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            opt.step(grad_scale=scale)
            return
        opt.wait_pending()
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            if self.timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.side)
            dist.all_reduce(opt.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            if self.timing is not None:
                e1.record(self.side)
                self.timing["exchange"].append((e0, e1, opt.flat_grad.numel() * 4))
            opt.step(grad_scale=scale)
            done = torch.cuda.Event()
            done.record(self.side)
        opt.pending = done
        if opt not in self._pending:
            self._pending.append(opt)

    def wait(self, opt=None):
        """Main stream waits for the in-flight update of `opt` (all optimisers when None)."""
        if self._segmenter is not None:
            self._segmenter.cut(lambda: self._wait_now(opt))
            return
        self._wait_now(opt)

    def start_timing(self):
        """From now on: HIP-event pairs around every bucket all-reduce (on the side stream) and around every wait of the main stream
        for an update in flight.  timing_report() turns them into milliseconds (it synchronises)."""
        self.timing = {"exchange": [], "wait": []}

    def timing_report(self, steps):
        """Per step: bytes and time of the bucket all-reduces, and the EXPOSED time - how long the main stream sat in front of an
        update still in flight (the part of exchange + Adam that the overlap did not hide)."""
        t, self.timing = self.timing, None
        if t is None:
            return None
        torch.cuda.synchronize()
        ex = [(a.elapsed_time(b), nb) for a, b, nb in t["exchange"]]
        wt = [a.elapsed_time(b) for a, b in t["wait"]]
        n = max(1, steps)
        return {"all_reduces_per_step": round(len(ex) / n, 2), "bucket_bytes_per_step": int(sum(nb for _, nb in ex) / n),
                "all_reduce_ms_per_step": round(sum(ms for ms, _ in ex) / n, 4),
                "exposed_wait_ms_per_step": round(sum(wt) / n, 4), "steps": steps}

    def _wait_now(self, opt):
        if self.timing is not None and self.side is not None and (self._pending if opt is None else opt in self._pending):
            main = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            self._wait_now_untimed(opt)
            e1.record(main)
            self.timing["wait"].append((e0, e1))
            return
        self._wait_now_untimed(opt)

    def _wait_now_untimed(self, opt):
        if opt is None:
            for o in self._pending:
                o.wait_pending()
            self._pending = []
        else:
            opt.wait_pending()
            if opt in self._pending:
                self._pending.remove(opt)

    def end_step(self):
        self._wait_now(None)

    # -- cross-replica BatchNorm -------------------------------------------------------------------
    def enable_sync_batchnorm(self):
        """BatchNorm statistics (forward) and batch sums (backward) over the GLOBAL batch: with equal shards the N-rank
        step then computes what the single-process reference computes on the whole batch for the BatchNorm models
        (dcgan.py:53-60, srgan/models.py:23-26,47,87-90, wgan_gp.py:49), instead of per-rank statistics.  Costs one
        all_gather of 2*C floats per BatchNorm forward and one all_reduce of 2*C floats per backward.  These run inside
        forward/backward: a recorded step (graph.StepRunner) is cut into one more hipGraph segment at each of them, the
        collective replaying eagerly in between on buffers that belong to the recording (DCGAN: 24 + the 2 optimiser cuts)."""
        from . import functional as F

        self.sync_bn = _SyncBN(self)
        F.set_sync_batchnorm(self.sync_bn)
        return self.sync_bn

    def disable_sync_batchnorm(self):
        from . import functional as F

        self.sync_bn = None
        F.set_sync_batchnorm(None)

    # -- helpers -----------------------------------------------------------------------------------
    def shard(self, t):
        """This rank's slice of a global batch (dim 0), equal shards."""
        n = t.shape[0]
        if n % self.world:
            raise ValueError("global batch %d is not divisible by world size %d" % (n, self.world))
        per = n // self.world
        return t[self.rank * per:(self.rank + 1) * per]

    def broadcast_parameters(self, *modules):
        """Make every replica start from rank 0's weights/buffers (the reference has a single copy)."""
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)


class _SyncBN:
    """The two collectives of cross-replica BatchNorm, on the current stream (functional._Norm calls them).

    While graph.StepRunner records the step (dp._segmenter set) a collective cannot go into the capture: the open hipGraph segment is
    closed, the collective is queued as an eager item between segments, and the next segment opens.  Its operands are tensors of the
    recording (allocated from the graph pool, kept alive by the queued closure), so every replay runs it on the addresses the
    neighbouring segments write and read.  The cut happens on the thread that records - graph.StepRunner turns the autograd worker
    threads off for the recording, so backward nodes run on it too."""

    def __init__(self, dp):
        self.dp, self.world = dp, dp.world
        self.cuts = 0   # collectives queued as eager items by the last recording

    def _run(self, fn):
        seg = self.dp._segmenter
        if seg is None:
            fn()
            return
        from . import functional as F

        F.join_wgrad_streams()   # a segment ends with every stream it forked joined
        seg.cut(fn)
        self.cuts += 1

    def all_gather(self, t):
        """[K] per rank -> [world * K], rank-major."""
        k = t.numel()
        src = t.contiguous()
        out = torch.empty(self.world * k, device=t.device, dtype=t.dtype)
        if dist.get_backend(self.dp.group) == "nccl":
            self._run(lambda: dist.all_gather_into_tensor(out, src, group=self.dp.group))
            return out
        # gloo (CPU tests, 2-ranks-on-one-GPU test mode): a summed all-reduce of a buffer in which every rank filled only
        # its own slot is the same gather, on the collective every backend implements for every device
        flat, lo = src.reshape(-1), self.dp.rank * k

        def gather():
            out.zero_()
            out[lo:lo + k] = flat
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.dp.group)

        self._run(gather)
        return out

    def all_reduce_sum(self, t):
        self._run(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.dp.group))
        return t


def init_from_env(backend=None):
    """One process per GPU as launched by torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return LocalStepper()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = os.environ.get("MIGAN_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        # MIGAN_DP_SINGLE_DEVICE=1 (test mode, with MIGAN_DP_BACKEND=gloo): every rank drives cuda:0
        torch.cuda.set_device(0 if os.environ.get("MIGAN_DP_SINGLE_DEVICE") == "1" else local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return DataParallel()
