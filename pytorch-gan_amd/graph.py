"""hipGraph capture of a whole training step (launch-bound inner loop -> one graph launch).

The eager step issues a few hundred small-to-medium kernels through Python/autograd/ctypes; on MI355X the
host becomes the bottleneck below ~3 us per kernel (MI355X_MICROARCH.md, row graph-replay-floor).  All
libmigan launchers are capture-safe (no allocation, no sync, device-side step counters and Philox offsets),
so the step is recorded once with torch.cuda.CUDAGraph (hipGraph on ROCm) and replayed.

world_size == 1: one graph for the whole step (forward, backward, Adam of G and D).
world_size  > 1: the step is cut at every `dp.step(opt)`; the compute segments are graphs, and the RCCL
all-reduce + fused Adam run eagerly between them on the side stream (no collective inside a capture).  With
cross-replica BatchNorm on (dp.enable_sync_batchnorm) the step is also cut at each of its collectives - inside forward
and backward; the recording runs autograd on the recording thread for that (a capture ends on the thread that began it).

What a capture freezes, and how the step bodies deal with it:
  * scalar kernel arguments — the learning rate is therefore read from a device scalar (optim.Adam.lr_t) that
    `run()` refreshes from `param_groups` before every replay, so LambdaLR schedules keep working;
  * host control flow and host RNG — `steps.wgan_gp_step` has two shapes (critic only / critic + generator):
    use `steps.WganGpRunner`, which captures both; `steps.ReplayBuffer` draws from python `random`: `steps.CycleGanRunner`
    draws in front of every replay into static device tables the recorded launches read (same draws, same order);
  * packed-weight cache entries never cross a capture boundary (functional.weight_cache_scope).
"""
import contextlib

import torch

from .optim import sync_all_lr


class _Segmenter:
    def __init__(self, pool, stream):
        self.pool, self.stream = pool, stream
        self.segments = []
        self._cm = None
        self._g = None

    def begin(self):
        self._g = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread polls its work events (hipEventQuery) while this thread records - under the
        # default "global" mode that query fails with hipErrorStreamCaptureUnsupported and takes the process down
        # (seen 2 runs in 5 with backend nccl, gpurun r2c/r2e)
        self._cm = torch.cuda.graph(self._g, pool=self.pool, stream=self.stream, capture_error_mode="thread_local")
        self._cm.__enter__()

    def end(self):
        self._cm.__exit__(None, None, None)
        self.segments.append(("graph", self._g))
        self._cm = self._g = None

    def cut(self, eager_fn):
        self.end()
        self.segments.append(("eager", eager_fn))
        self.begin()

    def abort(self, exc):
        """Close an open capture after an exception inside the recorded step (leaves no stream in capture mode)."""
        if self._cm is not None:
            cm, self._cm, self._g = self._cm, None, None
            try:
                cm.__exit__(type(exc), exc, exc.__traceback__)
            except Exception:
                pass


class StepRunner:
    """Runs `fn()` (one full training step on static device buffers) eagerly or as captured graph(s)."""

    def __init__(self, fn, dp, use_graph=True, warmup=3, before_capture=None):
        self.fn, self.dp, self.use_graph, self.warmup = fn, dp, use_graph, warmup
        self.before_capture = before_capture   # called once after the warm-up steps, before the recording (static buffers a capture needs)
        self.graphed = False
        self.out = None
        self._segments = None
        self.capture_error = None

    def _eager(self):
        self.dp.begin_step()
        out = self.fn()
        self.dp.end_step()
        return out

    def prepare(self):
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.out = self._eager()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        if not self.use_graph:
            return self
        if not getattr(self.dp, "graph_ok", True):
            self.capture_error = "not captured: the data-parallel wrapper does not allow it (graph_ok)"
            return self
        try:
            if self.before_capture is not None:
                self.before_capture()
            # the weight-pack plan learns a step's requests in one step and builds its device tables at the start of the NEXT - which cannot
            # happen inside a capture: behind a single warm-up step the recording would hold every pack as its own launch (and, with forked
            # streams, one copy per stream: 238 pack launches in the recorded one-image CycleGAN step instead of 20).  Build the tables now.
            from . import functional as F

            F.prebuild_pack_tables()
            self._capture(side)
            self.graphed = True
        except Exception as e:  # capture is an optimisation: fall back to eager launches, but say so
            self.capture_error = "%s: %s" % (type(e).__name__, e)
            self._segments = None
            self.graphed = False
            torch.cuda.synchronize()
        return self

    def _capture(self, stream):
        seg = _Segmenter(torch.cuda.graph_pool_handle(), stream)
        multi = getattr(self.dp, "segment", getattr(self.dp, "world", 1) > 1)  # DataParallel: collectives between segments
        if multi:
            self.dp._segmenter = seg
        sync_bn = getattr(self.dp, "sync_bn", None)
        if sync_bn is not None:
            sync_bn.cuts = 0
        # cross-replica BatchNorm cuts segments from inside backward nodes: those must run on this thread, not on autograd's device
        # worker (hipStreamEndCapture belongs to the thread of hipStreamBeginCapture under the thread_local mode)
        threads = torch.autograd.set_multithreading_enabled(False) if sync_bn is not None else contextlib.nullcontext()
        try:
            with threads:
                seg.begin()
                self.out = self.fn()
                seg.end()
        except BaseException as e:
            seg.abort(e)
            raise
        finally:
            if multi:
                self.dp._segmenter = None
        self._segments = seg.segments
        torch.cuda.synchronize()

    def run(self):
        if not self.graphed:
            self.out = self._eager()
            return self.out
        sync_all_lr()  # captured Adam launches read lr from a device scalar: push host-side schedule changes first
        self.dp.begin_step()
        for kind, item in self._segments:
            if kind == "graph":
                item.replay()
            else:
                item()
        self.dp.end_step()
        return self.out
