"""hipGraph capture of a whole training step (launch-bound inner loop -> one graph launch).

The eager step issues a few hundred small-to-medium kernels through Python/autograd/ctypes; on MI355X the
host becomes the bottleneck below ~3 us per kernel (MI355X_MICROARCH.md, row graph-replay-floor).  All
libmigan launchers are capture-safe (no allocation, no sync, device-side step counters and Philox offsets),
so the step is recorded once with torch.cuda.CUDAGraph (hipGraph on ROCm) and replayed.

world_size == 1: one graph for the whole step (forward, backward, Adam of G and D).
world_size  > 1: the step is cut at every `dp.step(opt)`; the compute segments are graphs, and the RCCL
all-reduce + fused Adam run eagerly between them on the side stream (no collective inside a capture).
"""
import torch


class _Segmenter:
    def __init__(self, pool, stream):
        self.pool, self.stream = pool, stream
        self.segments = []
        self._cm = None
        self._g = None

    def begin(self):
        self._g = torch.cuda.CUDAGraph()
        self._cm = torch.cuda.graph(self._g, pool=self.pool, stream=self.stream)
        self._cm.__enter__()

    def end(self):
        self._cm.__exit__(None, None, None)
        self.segments.append(("graph", self._g))
        self._cm = self._g = None

    def cut(self, eager_fn):
        self.end()
        self.segments.append(("eager", eager_fn))
        self.begin()


class StepRunner:
    """Runs `fn()` (one full training step on static device buffers) eagerly or as captured graph(s)."""

    def __init__(self, fn, dp, use_graph=True, warmup=3):
        self.fn, self.dp, self.use_graph, self.warmup = fn, dp, use_graph, warmup
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
        try:
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
        multi = getattr(self.dp, "world", 1) > 1
        if multi:
            self.dp._segmenter = seg
        try:
            seg.begin()
            self.out = self.fn()
            seg.end()
        finally:
            if multi:
                self.dp._segmenter = None
        self._segments = seg.segments
        torch.cuda.synchronize()

    def run(self):
        if not self.graphed:
            self.out = self._eager()
            return self.out
        self.dp.begin_step()
        for kind, item in self._segments:
            if kind == "graph":
                item.replay()
            else:
                item()
        self.dp.end_step()
        return self.out
