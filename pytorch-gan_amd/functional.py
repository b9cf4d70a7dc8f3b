"""torch.autograd.Function wrappers over the C ABI (include/migan.h).

Every op here launches hand-written HIP kernels from libmigan.so on torch's current HIP stream.  There is
no CPU or ATen fallback: CPU tensors raise.  4-D tensors are logical NCHW / physical NHWC
(torch.channels_last); inputs that arrive NCHW-contiguous (e.g. from `.view` in dcgan.py:68) are re-laid
out by migan_transpose_batched.

Backward passes are themselves built from Functions where the reference needs a second derivative
(Linear + LeakyReLU for wgan_gp.py:119-138), so `autograd.grad(..., create_graph=True)` works.
"""
import contextlib

import torch
from torch.autograd import Function

from ._lib import check, lib

CL = torch.channels_last
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
GATHER_ZERO, GATHER_REFLECT, GATHER_UP2 = 0, 1, 2
LOSS_BCE, LOSS_MSE, LOSS_L1, LOSS_MEAN, LOSS_BCE_LOGITS = 0, 1, 2, 3, 4


def _stream():
    return torch.cuda.current_stream().cuda_stream


_SIDE_STREAMS = {}


# Weight-gradient launches on their own stream, joined only where the optimiser needs them (steps / dp call join_wgrad_streams()
# before every optimiser step): a weight gradient feeds nothing but Adam, so the backward chain on the main stream - the HBM-bound
# normalisation passes and the next layer's input gradient - does not wait for it, and a streaming norm kernel runs beside an MFMA-bound
# weight gradient instead of after it.  All weight gradients of one main stream go to ONE side stream in launch order, so the
# accumulation order into a shared parameter's gradient (a generator applied three times in cyclegan.py:170-190) is unchanged and the
# results are bit-identical.  Only where the gradient goes straight into the optimiser's bucket (functional._grad_slot): a gradient
# returned to autograd is joined at once.  Tests and bench.py --no-overlap flip it.
_WGRAD_STREAM = True
# ... for gradients of at least this many elements (dy): the launch must be long enough (an MFMA-bound weight gradient of >= ~100 us) to
# be worth two more edges in the stream / hipGraph.  Measured with every weight gradient forked (profiles/r04_ab.txt, call 11): CycleGAN
# 149.6 -> 144.9 ms, but the captured DCGAN step 2.53 -> 2.72 ms and pix2pix 3.32 -> 3.62 ms (dozens of forks around 5-20 us launches)
_WGRAD_STREAM_MIN = 6 << 20   # (1 M and 256 k measured on SRGAN: 82.7-82.8 vs 82.5-82.7 ms, profiles/r04_ab.txt call 28)
# > 0 (one_wgrad_stream()): EVERY parameter-gradient launch of the region goes to ONE stream per device, whatever its size and whichever
# stream its backward node runs on.  A step body that runs two forward chains on two streams (cyclegan_step: the A -> B -> A and the
# B -> A -> B half of cyclegan.py:170-190 use the same two generators) has its backward on two streams as well - autograd runs a node on
# its forward's stream - and both halves add into the same parameters' gradients: on one stream, in the autograd engine's (fixed) node order,
# those additions stay serial and in the order of the one-stream step.
_WGRAD_ONE_STREAM = 0
_PENDING_WGRAD = {}
_PENDING_READS = []   # tensors read by deferred weight-gradient launches of a recording in progress (see _Fork.join)


@__import__("contextlib").contextmanager
def one_wgrad_stream():
    global _WGRAD_ONE_STREAM
    _WGRAD_ONE_STREAM += 1
    try:
        yield
    finally:
        _WGRAD_ONE_STREAM -= 1


def join_wgrad_streams():
    """The current stream waits for every weight-gradient stream with launches in flight (before an optimiser step, a gradient
    all-reduce, or the end of a captured hipGraph segment)."""
    if not _PENDING_WGRAD:
        return
    for dev_index, side in list(_PENDING_WGRAD.values()):
        torch.cuda.current_stream(dev_index).wait_stream(side)
    _PENDING_WGRAD.clear()
    _PENDING_READS.clear()


class _Fork:
    """fork(): the weight-gradient stream waits for the current one and becomes current;  join(): deferred to the next
    join_wgrad_streams() when everything the side launches produced went into gradient slots, else the current stream waits at once."""

    def __init__(self, device, numel=0, wgrad=False):
        first_order = not torch.is_grad_enabled()
        # only inside a step body (weight_cache_scope): its optimiser steps and its end join the stream - a bare loss.backward() of user
        # code reads .grad right away.  (Round 2's other form - the weight gradient beside its own layer's input gradient, joined before
        # backward returns - measured +0.3 % CycleGAN / -4 % DCGAN, profiles/r02_ab.txt, and is gone.)
        force = _WGRAD_ONE_STREAM > 0
        self.defer = (bool(wgrad) and first_order and device.type == "cuda" and _CACHE_SCOPE is not None
                      and (force or (_WGRAD_STREAM and numel >= _WGRAD_STREAM_MIN)))
        self.on = self.defer
        if self.on:
            self.main = torch.cuda.current_stream(device)
            key = (device.index, "all") if (force and self.defer) else (device.index, self.main.cuda_stream)
            if key not in _SIDE_STREAMS:
                _SIDE_STREAMS[key] = torch.cuda.Stream(device)
                ensure_splitk_ws(device, _SIDE_STREAMS[key])
            self.side = _SIDE_STREAMS[key]
            self.ctx = None
            self.key = key

    def __enter__(self):
        if self.on:
            self.side.wait_stream(self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self, returned=(), reads=()):
        """`returned`: what the side launches produced for autograd (all None = everything went into gradient slots);
        `reads`: main-stream tensors the side launches read (kept from being recycled under them when the join is deferred)."""
        if not self.on:
            return
        if self.defer and _WGRAD_ONE_STREAM > 0 and not all(t is None for t in returned):
            raise RuntimeError("one_wgrad_stream(): a parameter gradient without a gradient slot would be accumulated by autograd on its "
                               "own stream (the step body must own every parameter through an optimiser's bucket)")
        if self.defer and all(t is None for t in returned):
            if torch.cuda.is_current_stream_capturing():
                # inside a recording the allocator frees a block with a recorded stream use only when the capture ends: every activation
                # and gradient a forked weight gradient reads would stay pinned for the whole recorded step.  Holding the tensors by
                # reference until the join has the same effect on correctness (not recycled under the side launch) and releases them at
                # the join.
                _PENDING_READS.extend(t for t in reads if t is not None)
            else:
                for t in reads:
                    if t is not None:
                        t.record_stream(self.side)
            _PENDING_WGRAD[self.key] = (self.key[0], self.side)
            return
        # Every immediate side-stream use starts by waiting for the main stream and ends with the main stream waiting for it, so
        # buffers from either stream's allocator pool are never recycled under a kernel that still reads them.
        self.main.wait_stream(self.side)


def _plain(t):
    """Strip tensor subclasses (GanTensor / Parameter) without copying."""
    if t is None or type(t) is torch.Tensor:
        return t
    with torch._C.DisableTorchFunctionSubclass():
        return t.as_subclass(torch.Tensor)


def on_device(t):
    """THE definition of "this tensor lives on the HIP device": every device check of the host mirror asks here."""
    return t.is_cuda


def _check_dev(t):
    if not on_device(t):
        raise RuntimeError("pytorch_gan_amd: tensor is on %s; the HIP path has no CPU fallback" % t.device)
    if t.dtype != torch.float32:
        raise TypeError("pytorch_gan_amd: fp32 only (got %s)" % t.dtype)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _empty_nhwc(shape, ref):
    return torch.empty(shape, device=ref.device, dtype=torch.float32, memory_format=CL)


def to_nhwc(x):
    """Dense channels_last copy of a 4-D tensor (no-op when it already is)."""
    x = _plain(x)
    _check_dev(x)
    if x.is_contiguous(memory_format=CL):
        return x
    if not x.is_contiguous():
        x = x.contiguous()
    N, C, H, W = x.shape
    y = _empty_nhwc(x.shape, x)
    check(lib.migan_transpose_batched(x.data_ptr(), y.data_ptr(), N, C, H * W, _stream()), "transpose")
    return y


def to_nchw(x):
    """Dense NCHW-contiguous copy of a 4-D tensor (no-op when it already is)."""
    x = _plain(x)
    _check_dev(x)
    if x.is_contiguous():
        return x
    xs = to_nhwc(x)
    N, C, H, W = xs.shape
    y = torch.empty(xs.shape, device=xs.device, dtype=torch.float32)
    check(lib.migan_transpose_batched(xs.data_ptr(), y.data_ptr(), N, H * W, C, _stream()), "transpose")
    return y


def canon(x):
    """Canonical dense layout: channels_last for 4-D, contiguous otherwise."""
    x = _plain(x)
    _check_dev(x)
    if x.dim() == 4:
        return to_nhwc(x)
    return x if x.is_contiguous() else x.contiguous()


def _permute4(w, perm):
    w = _plain(w)
    if not w.is_contiguous():
        w = w.contiguous()
    d = list(w.shape)
    out = torch.empty([d[p] for p in perm], device=w.device, dtype=torch.float32)
    check(lib.migan_permute4d(w.data_ptr(), out.data_ptr(), d[0], d[1], d[2], d[3], perm[0], perm[1], perm[2],
                              perm[3], _stream()), "permute4d")
    return out


_SK_WS = {}


def _splitk_ws(ref, max_m, Co, Ci_src, ncls=1):
    """(pointer, bytes) of the split-K workspace of the conv / dgrad launch about to be issued on the current stream, or
    (None, 0) when the launch would not use one.  One persistent zero-initialised buffer per (device, stream): the tickets
    at its head return to zero at the end of every launch, and launches of one stream cannot overlap."""
    if not _SPLITK or lib.migan_conv_splitk_applies(int(max_m), int(Co), int(Ci_src), int(ncls)) != 1:
        return None, 0
    key = (ref.device.index, torch.cuda.current_stream(ref.device).cuda_stream)
    ws = _SK_WS.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            # The tickets must be zero AT REST: a buffer created inside a capture would be zeroed by a captured memset node only
            # (never eagerly) and would live in that graph's private pool while staying cached here under the stream key - a later
            # graph or eager launch on the stream would then run on unzeroed tickets and silently skip its output tile.
            # No cached buffer for this stream yet => this launch takes the un-split kernel: the same result up to summation order
            # (the recorded and the eager launch of this geometry then differ in the last bits).  Counted: the step bodies create the
            # workspace of every stream they fork when they create the stream (ensure_splitk_ws), so this should not happen -
            # tests assert SPLITK_CAPTURE_FALLBACKS stays 0.
            global SPLITK_CAPTURE_FALLBACKS
            SPLITK_CAPTURE_FALLBACKS += 1
            return None, 0
        ws = torch.zeros(lib.migan_conv_splitk_workspace() // 4, device=ref.device, dtype=torch.float32)
        _SK_WS[key] = ws
    return ws.data_ptr(), ws.numel() * 4


SPLITK_CAPTURE_FALLBACKS = 0


def ensure_splitk_ws(device, stream):
    """Create the (zeroed) split-K workspace of `stream` now, outside any capture: called where a step body creates a side stream,
    so a stream first used inside a recording already has one."""
    if not _SPLITK or device.type != "cuda":
        return
    key = (device.index, stream.cuda_stream)
    if key not in _SK_WS and not torch.cuda.is_current_stream_capturing():
        _SK_WS[key] = torch.zeros(lib.migan_conv_splitk_workspace() // 4, device=device, dtype=torch.float32)


_SPLITK = __import__("os").environ.get("MIGAN_SPLITK", "1") == "1"  # A/B knob


def _ws(nbytes, ref):
    return torch.empty(max(int(nbytes) // 4, 1), device=ref.device, dtype=torch.float32)


_DIRECT_GRAD = True
_WEIGHT_CACHE = True   # only effective inside weight_cache_scope()
# bias gradient inside the wgrad launches (include/migan.h): measured no faster than the column-sum launches -> opt-in
_FUSE_BIAS = False


def _bias_out(param, C, ref):
    """(buffer, accumulate flag, value to return to autograd) for a bias gradient produced by a wgrad launch."""
    slot = _grad_slot(param)
    if slot is not None:
        return slot, 1, None
    t = torch.empty(C, device=ref.device, dtype=torch.float32)
    return t, 0, t


def set_weight_cache(enabled):
    """Globally enable/disable the step-scoped re-use of packed weights (see weight_cache_scope)."""
    global _WEIGHT_CACHE
    _WEIGHT_CACHE = bool(enabled)


_CACHE_SCOPE = None          # token of the training-step invocation in progress (None: no caching)
_SCOPE_IDS = __import__("itertools").count(1)


_SCOPE_OWNER = None          # what the scope in progress belongs to (a step state): the per-step plans are kept per owner


def _owner_plans(shared):
    """The dict that holds the per-step plans of the scope in progress: on the owner object itself (they die with it - an
    arena of packed weights per plan), or `shared` (per class, per device) for scopes without an owner."""
    o = _SCOPE_OWNER
    if o is not None and hasattr(o, "__dict__"):
        return o.__dict__.setdefault("_migan_plans", {})
    return shared


@__import__("contextlib").contextmanager
def weight_cache_scope(owner=None):
    """Inside this scope the packed (OHWI / IHWO / phase-collapsed) copies of a weight are re-used between calls
    until its optimiser steps: a discriminator applied three times per DCGAN step is packed once (and once for dgrad).

    The scope is ONE invocation of a `steps.*_step` body, where weights change only through `optim.Adam.step()`
    (which renews the parameter's epoch stamp).  Cached packs never outlive the scope, so edits between steps that
    no stamp can see — `weights_init_normal` writing `m.weight.data` (dcgan.py:36-42), `load_state_dict`, an optimiser
    update replayed from a hipGraph — can not be served a stale pack; a capture and the eager steps around it never
    share entries either (the captured step is one scope, every eager step another)."""
    global _CACHE_SCOPE, _SCOPE_OWNER
    prev, prev_owner = _CACHE_SCOPE, _SCOPE_OWNER
    if prev is None:
        _CACHE_SCOPE = next(_SCOPE_IDS)
        # `owner` (the step state object): the pack / dropout-mask plans learn the request sequence of ONE step body.
        # Two bodies that alternate on a device (two models, a train and an eval step) each keep their own plan instead of
        # overwriting a shared one every step (tables rebuilt, arenas re-allocated, plans dropped during capture).
        _SCOPE_OWNER = owner
    try:
        yield
    finally:
        _CACHE_SCOPE, _SCOPE_OWNER = prev, prev_owner
        if prev is None and _STREAM_PACKS:
            # the per-stream pack copies a recording made (see _packed): no later scope overwrites a (kind, stream) key, so without this
            # the capture pool's tensors would stay referenced for the Parameter's lifetime - one leaked copy per recording
            for cache, key in _STREAM_PACKS:
                cache.pop(key, None)
            del _STREAM_PACKS[:]


_STREAM_PACKS = []   # (pack cache dict, (kind, stream) key) entries made inside the scope in progress


def _packed(param, w, kind, make):
    """`make()` -> packed tensor (or tuple of tensors) of plain weight `w`; cached on the Parameter object `param`.
    A parameter without an optimiser stamp (the frozen VGG of srgan.py:60-62) is cached for the scope as well: inside a step
    body nothing but `optim.Adam.step()` writes weights."""
    if not _WEIGHT_CACHE or param is None or _CACHE_SCOPE is None or not isinstance(param, torch.nn.Parameter):
        return make()
    ep = getattr(param, "_migan_epoch", None)
    stamp = (_CACHE_SCOPE, ep, w._version, w.data_ptr())
    cache = param.__dict__.setdefault("_migan_pack", {})
    hit = cache.get(kind)
    if hit is not None and hit[0] == stamp:
        if _TWO_STREAMS and len(hit) > 2 and hit[2] is not None:   # made on the other stream of a forked step body: wait for it
            cur = torch.cuda.current_stream(w.device)
            if hit[2][0] != cur.cuda_stream:
                if torch.cuda.is_current_stream_capturing():
                    # Inside a recording two forked streams must never wait for EACH OTHER's events: hipStreamWaitEvent makes the waiting
                    # stream a "parallel capture stream" of the event's stream every time, two side streams that each hit a pack the other
                    # made end up in each other's lists, and hip::Stream::EndCapture() walks those lists recursively without marking
                    # visited streams - unbounded recursion, SIGSEGV in hipStreamEndCapture (ROCm 7.0 runtime; rocgdb backtrace in
                    # profiles/r05_capture_crash.txt: the CycleGAN step with the second forward chain AND the discriminator halves forked,
                    # the PatchGAN 256 -> 512 weight - too large for the step's pack plan - packed on one and hit on the other).
                    # While recording, a stream that did not make the pack makes its own copy instead (same bits, one more launch).
                    own = cache.get((kind, cur.cuda_stream))
                    if own is not None and own[0] == stamp:
                        return own[1]
                    t = make()
                    cache[(kind, cur.cuda_stream)] = (stamp, t, None)
                    _STREAM_PACKS.append((cache, (kind, cur.cuda_stream)))
                    return t
                cur.wait_event(hit[2][1])
        return hit[1]
    t = make()
    made = None
    if _TWO_STREAMS and on_device(w):
        cur = torch.cuda.current_stream(w.device)
        ev = torch.cuda.Event()
        ev.record(cur)
        made = (cur.cuda_stream, ev)
    cache[kind] = (stamp, t, made)
    return t


# > 0 while a step body runs two halves on two streams (steps.dcgan_step: the discriminator update underneath the generator's
# backward): a weight pack made inside that region carries the event of its launch, and a hit from the other stream waits for it.
# Packs made before the fork (the step's multi-tensor plan launch, the forwards in front of the fork) need no event.
_TWO_STREAMS = 0


@__import__("contextlib").contextmanager
def two_streams():
    global _TWO_STREAMS
    _TWO_STREAMS += 1
    try:
        yield
    finally:
        _TWO_STREAMS -= 1


_BATCH_PACKS = True   # False = one permute launch per pack
# only weights up to this many elements go through the plan: the plan saves launch latency, and a large pack written at the
# start of the step has left the MALL by the time its conv runs (pix2pix, 54 M parameters: 8.65 -> 9.50 ms with every
# weight planned, profiles/r02_ab.txt)
_BATCH_PACKS_MAX = 1 << 20


class _PackPlan:
    """All permute-type weight packs (OHWI for forward / wgrad-of-dgrad, IHWO for dgrad) of one training step from ONE launch.
    A step body (weight_cache_scope) asks for the same packs every iteration, so the requests recorded in one step are the
    plan of the next: on its first request the new scope runs migan_multi_permute4d over the whole plan into one arena and
    pre-fills the per-parameter cache entries `_packed` looks up (same stamps, so a weight whose optimiser steps later in
    the scope is simply re-packed by its own launch when it is next used).  Device tables and the arena are rebuilt only
    when the set of (parameter storage, shape, permutation) changes, so a captured hipGraph replays the one launch."""

    plans = {}
    keep_alive = []   # (table, blocks, arena) of plans a captured hipGraph still launches
    instances = __import__("weakref").WeakSet()   # every live plan (they hang on their step-state owners): prebuild_pack_tables()

    def __init__(self):
        self.scope, self.seq, self.sig, self.tab, self.blk, self.arena, self.nblocks = None, {}, None, None, None, None, 0
        self.captured = False
        _PackPlan.instances.add(self)

    @classmethod
    def get(cls, device):
        plans, key = _owner_plans(cls.plans), ("pack", str(device))
        if key not in plans:
            plans[key] = cls()
        return plans[key]

    def note(self, param, w, kind, perm):
        scope = _CACHE_SCOPE
        if scope is None or not (_BATCH_PACKS and _WEIGHT_CACHE) or not isinstance(param, torch.nn.Parameter) or w.dim() != 4 \
                or w.numel() > _BATCH_PACKS_MAX:
            return
        if scope != self.scope:
            prev, self.scope, self.seq = self.seq, scope, {}
            if prev:
                self._prefill(prev, w.device)
        self.seq[(id(param), kind)] = (__import__("weakref").ref(param), kind, perm if isinstance(perm, str) else tuple(perm))

    def _prefill(self, prev, device):
        import numpy as np

        live, c64 = [], []
        for ref, kind, perm in prev.values():
            p = ref()
            if p is not None and on_device(p) and p.dtype == torch.float32 and p.is_contiguous():
                (c64 if perm == "c64" else live).append((p, kind, perm))
        if c64:
            self._prefill_c64([p for p, _, _ in c64], device)
        if not live:
            return
        capturing = torch.cuda.is_current_stream_capturing()
        if not self._tables(live, device):
            return
        self.captured = self.captured or capturing
        check(lib.migan_multi_permute4d(self.tab.data_ptr(), self.blk.data_ptr(), self.nblocks, _stream()), "multi_permute4d")
        off = 0
        for p, kind, perm in live:
            n = p.numel()
            view = self.arena[off:off + n].view([p.shape[i] for i in perm])
            off += n
            stamp = (_CACHE_SCOPE, getattr(p, "_migan_epoch", None), p._version, p.data_ptr())
            p.__dict__.setdefault("_migan_pack", {})[kind] = (stamp, view)


def _pack_tables(self, live, device):
    """Device tables + arena of the multi-tensor permute launch for `live` = [(param, kind, perm)]; False when they would have to be built
    inside a capture (a host->device copy)."""
    import numpy as np

    sig = tuple((p.data_ptr(), tuple(p.shape), kind, perm) for p, kind, perm in live)
    if sig == self.sig:
        return True
    if torch.cuda.is_current_stream_capturing():
        return False
    if self.captured:  # a live hipGraph replays a launch that reads these tables and writes this arena: never free them
        _PackPlan.keep_alive.append((self.tab, self.blk, self.arena))
        self.captured = False
    total = sum(p.numel() for p, _, _ in live)
    self.arena = torch.empty(total, device=device, dtype=torch.float32)
    ent = np.zeros(len(live), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("o", "<u4", 4), ("s", "<i8", 4), ("n", "<i8")]))
    blocks, off = [], 0
    for i, (p, kind, perm) in enumerate(live):
        n = p.numel()
        d = tuple(p.shape)
        st = (d[1] * d[2] * d[3], d[2] * d[3], d[3], 1)
        ent[i] = (p.data_ptr(), self.arena.data_ptr() + 4 * off, (d[perm[1]], d[perm[2]], d[perm[3]], 0),
                  tuple(st[q] for q in perm), n)
        blocks += [(i, c) for c in range((n + 1023) // 1024)]
        off += n
    self.tab = torch.from_numpy(ent.view(np.uint8).copy()).to(device)
    self.blk = torch.tensor(blocks, dtype=torch.int32).to(device)
    self.nblocks, self.sig = len(blocks), sig
    return True


_PackPlan._tables = _pack_tables


def _plan_live(seq):
    live, c64 = [], []
    for ref, kind, perm in seq.values():
        p = ref()
        if p is not None and on_device(p) and p.dtype == torch.float32 and p.is_contiguous():
            (c64 if perm == "c64" else live).append((p, kind, perm))
    return live, c64


def prebuild_pack_tables():
    """Build the device tables of every pack plan from the request sequence of the step that just ran (graph.StepRunner, in front of a
    recording): the recording's first pack request then finds them and issues the ONE multi-tensor launch."""
    if torch.cuda.is_current_stream_capturing():
        return
    for plan in list(_PackPlan.instances):
        if not plan.seq:
            continue
        live, c64 = _plan_live(plan.seq)
        dev = (live or c64)[0][0].device if (live or c64) else None
        if live:
            plan._tables(live, dev)
        if c64:
            plan._tables_c64([p for p, _, _ in c64], dev)


def _prefill_c64(self, params, device):
    """Both register-slice packs (forward, input gradient) of every Conv2d(64, 64, 3, 1, 1) weight the last step used, from ONE launch
    (csrc/conv_c64.hip c64_pack_multi_kernel): table and arena are rebuilt only when the set of weights changes."""
    import numpy as np

    n1 = lib.migan_c64_pack_floats()
    capturing = torch.cuda.is_current_stream_capturing()
    if not self._tables_c64(params, device):
        return
    self.c64_captured = getattr(self, "c64_captured", False) or capturing
    check(lib.migan_c64_pack_multi(self.c64_tab.data_ptr(), len(params), _stream()), "c64_pack_multi")
    for i, p in enumerate(params):
        stamp = (_CACHE_SCOPE, getattr(p, "_migan_epoch", None), p._version, p.data_ptr())
        cache = p.__dict__.setdefault("_migan_pack", {})
        cache["c64f"] = (stamp, self.c64_arena[2 * i * n1:(2 * i + 1) * n1])
        cache["c64d"] = (stamp, self.c64_arena[(2 * i + 1) * n1:(2 * i + 2) * n1])


def _tables_c64(self, params, device):
    import numpy as np

    n1 = lib.migan_c64_pack_floats()
    sig = tuple(p.data_ptr() for p in params)
    if sig == getattr(self, "c64_sig", None):
        return True
    if torch.cuda.is_current_stream_capturing():
        return False
    if getattr(self, "c64_captured", False):
        _PackPlan.keep_alive.append((self.c64_tab, self.c64_arena))
        self.c64_captured = False
    self.c64_arena = torch.empty(2 * n1 * len(params), device=device, dtype=torch.float32)
    ent = np.zeros((len(params), 3), dtype=np.uint64)
    for i, p in enumerate(params):
        base = self.c64_arena.data_ptr() + 4 * (2 * i) * n1
        ent[i] = (p.data_ptr(), base, base + 4 * n1)
    self.c64_tab = torch.from_numpy(ent.view(np.uint8).copy()).to(device)
    self.c64_sig = sig
    return True


_PackPlan._prefill_c64 = _prefill_c64
_PackPlan._tables_c64 = _tables_c64


def prefill_packs(device):
    """Run the step's multi-tensor pack launch NOW on the current stream (normally it rides on the first pack request of the step
    body): a body that forks streams before its first conv calls this in front of the fork, so that no stream uses a planned pack
    another stream is still writing."""
    plan = _PackPlan.get(device)
    scope = _CACHE_SCOPE
    if scope is None or not (_BATCH_PACKS and _WEIGHT_CACHE) or scope == plan.scope:
        return
    prev, plan.scope, plan.seq = plan.seq, scope, {}
    if prev:
        plan._prefill(prev, device)


def _packed_perm(param, w, kind, perm):
    """Cached `w.permute(perm).contiguous()` (weight pack), produced by the step's one multi-tensor launch when planned."""
    _PackPlan.get(w.device).note(param, w, kind, perm)
    return _packed(param, w, kind, lambda: _permute4(w, perm))


def set_direct_grad(enabled):
    """Parameter gradients are reduced straight INTO an existing contiguous `param.grad` (the optimiser's flat
    bucket) by the wgrad / bias / norm reductions, instead of being returned to autograd, which would launch one
    `aten::add` per parameter per backward pass (AccumulateGrad).  `param.grad` holds the same values either way.
    Disable before calling `torch.autograd.grad(..., params)` (it would see no gradient for those inputs)."""
    global _DIRECT_GRAD
    _DIRECT_GRAD = bool(enabled)


def _grad_slot(p):
    """The buffer a first-order backward may accumulate this leaf's gradient into directly, or None."""
    if not _DIRECT_GRAD or p is None or torch.is_grad_enabled():  # create_graph backward: stay differentiable
        return None
    if not (p.is_leaf and p.requires_grad):
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or g.device != p.device or g.shape != p.shape or not g.is_contiguous():
        return None
    return g


def _colsum(x2d_ptr_tensor, P, C, slot=None):
    """Column sums of a [P][C] matrix; with `slot` the result is ADDED into it and None is returned."""
    out = torch.empty(C, device=x2d_ptr_tensor.device, dtype=torch.float32) if slot is None else slot
    nb = lib.migan_colsum_workspace(P, C)
    ws = _ws(nb, x2d_ptr_tensor)
    check(lib.migan_colsum(x2d_ptr_tensor.data_ptr(), out.data_ptr(), P, C, ws.data_ptr(), nb,
                           0 if slot is None else 1, _stream()), "colsum")
    return out if slot is None else None


_COLSUM_FUSE = True


def _attach_colsum(t, slabs, nslab, C):
    """Remember that `slabs` ([nslab][C], written by the kernel that produced `t`) hold per-block column sums of `t`."""
    t._migan_colsum = (slabs, nslab, C, t.data_ptr(), t._version)
    return t


def _colsum_side(t, C):
    """(slabs, nslab) when the kernel that produced gradient `t` also left its column-sum slabs (and `t` is still that
    data), else None.  autograd hands a Function's returned gradient object to the next node as is when nothing is
    accumulated into it; pointer, version and width are checked anyway."""
    side = getattr(t, "_migan_colsum", None)
    if side is None or not _COLSUM_FUSE:
        return None
    slabs, nslab, c, ptr, ver = side
    if c != C or ptr != t.data_ptr() or ver != t._version or not t.is_contiguous(memory_format=CL):
        return None
    return slabs, nslab


def _act_bwd_colsum(dy, y, mask, N, HW, C, act, slope):
    """dx = dy * mask[n][c] * act'(y) and its column-sum slabs in one pass (backward of conv -> act [-> Dropout2d])."""
    g = torch.empty_like(dy)
    nslab = lib.migan_norm_colsum_slabs(N, HW, C)
    slabs = torch.empty(max(nslab * C, 1), device=dy.device, dtype=torch.float32)
    check(lib.migan_act_bwd_colsum(dy.data_ptr(), _ptr(y), _ptr(mask), g.data_ptr(), slabs.data_ptr(), N, HW, C, act,
                                   slope, _stream()), "act_bwd_colsum")
    return g, (slabs, nslab)


def _act_bwd_raw(dy, y, act, slope):
    dx = torch.empty_like(y)
    check(lib.migan_act_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), act, slope, _stream()), "act_bwd")
    return dx


# ---------------------------------------------------------------------------------------------- convolution
def _conv_out(HL, pt, pb, R, stride):
    return (HL + pt + pb - R) // stride + 1


_INPUT_GRAD_ONLY = False


def _first_order_only(what):
    """Backward passes made of raw kernel launches record no graph: refuse to run under create_graph=True instead of
    silently returning a gradient that cannot be differentiated again."""
    if torch.is_grad_enabled():
        raise NotImplementedError("%s: double backward (create_graph=True) is not implemented for this op - the reference "
                                  "only differentiates twice through Conv2d / Linear / BatchNorm / InstanceNorm / "
                                  "activations / Dropout2d (gradient penalties)" % what)


@__import__("contextlib").contextmanager
def input_grad_only():
    """Inside this scope a differentiable (create_graph=True) backward computes input gradients only - the case of the
    gradient penalties, `autograd.grad(outputs=D(x), inputs=x, create_graph=True)` (wgan_gp.py:128-135,
    dragan.py:156-163): autograd calls every node's backward in full, and without this hint each conv / linear would also
    run the weight-gradient kernels whose results that call discards."""
    global _INPUT_GRAD_ONLY
    prev, _INPUT_GRAD_ONLY = _INPUT_GRAD_ONLY, True
    try:
        yield
    finally:
        _INPUT_GRAD_ONLY = prev


class _ConvDgradFn(Function):
    """dx = dgrad(dy, w) of a zero-padded conv as a differentiable op: its backward is a conv forward (w.r.t. dy) and a
    weight gradient with the incoming gradient in the place of x (w.r.t. w) - the second-order terms of the conv-critic
    gradient penalties, built from the first-order kernels."""

    @staticmethod
    def forward(ctx, dy, w, geom):
        N, H, W, Ci, Ho, Wo, Co, R, S, stride, pt, pl = geom
        dy = to_nhwc(dy)
        wp = _plain(w)
        wt = _permute4(wp, (1, 2, 3, 0))
        dx = _empty_nhwc((N, Ci, H, W), dy)
        skp, skb = _splitk_ws(dy, N * -(-H // stride) * -(-W // stride), Ci, Co, stride * stride)
        check(lib.migan_conv2d_dgrad_ws(dy.data_ptr(), wt.data_ptr(), None, dx.data_ptr(), N, H, W, Ci, Ho, Wo, Co, R, S,
                                        stride, pt, pl, 0, 0.0, skp, skb, _stream()), "conv2d_dgrad")
        ctx.geom = geom
        ctx.save_for_backward(dy, wp)
        ctx.param = w
        return dx

    @staticmethod
    def backward(ctx, g):
        dy, w = ctx.saved_tensors
        N, H, W, Ci, Ho, Wo, Co, R, S, stride, pt, pl = ctx.geom
        g = to_nhwc(g)
        st = _stream()
        g_dy = g_w = None
        if ctx.needs_input_grad[0]:   # d/d(dy): the forward conv applied to g
            wo = _permute4(w, (0, 2, 3, 1))
            g_dy = _empty_nhwc((N, Co, Ho, Wo), g)
            check(lib.migan_conv2d_fwd(g.data_ptr(), wo.data_ptr(), None, g_dy.data_ptr(), N, H, W, Ci, Ho, Wo, Co, R, S,
                                       stride, pt, pl, GATHER_ZERO, 0, 0.0, st), "conv2d_fwd (dgrad of dgrad)")
        if ctx.needs_input_grad[1]:   # d/dw: wgrad with g in the place of x
            slot = _grad_slot(ctx.param)
            g_w = torch.empty_like(w) if slot is None else slot
            nb = lib.migan_conv2d_wgrad_workspace(N, Ho, Wo, Co, R, S, Ci)
            ws = _ws(nb, g)
            check(lib.migan_conv2d_wgrad(g.data_ptr(), dy.data_ptr(), g_w.data_ptr(), ws.data_ptr(), nb, N, H, W, Ci, Ho, Wo,
                                         Co, R, S, stride, pt, pl, GATHER_ZERO, 0 if slot is None else 1, None, 0, None, 0,
                                         st), "conv2d_wgrad (wgrad of dgrad)")
            if slot is not None:
                g_w = None
        return g_dy, g_w, None


def _fewpix_nt(a, w, b, out, M, N, K, act, slope, st, what):
    """out[M][N] = act(a[M][K] w[N][K]^T + b) of the few-pixel conv path: K split over workgroups (csrc/fewpix.hip)"""
    nb = lib.migan_fewpix_nt_workspace(M, N, K)
    ws = _ws(nb, a) if nb else None
    check(lib.migan_fewpix_nt(a.data_ptr(), w.data_ptr(), _ptr(b), out.data_ptr(), _ptr(ws), nb, M, N, K, act, slope, st), what)


_C64 = __import__("os").environ.get("MIGAN_C64", "1") == "1"   # A/B knob (round 6): 0 = the general kernels for Conv2d(64, 64, 3, 1, 1)


def _packed_c64(param, w, flip):
    """The forward (flip = 0) / input-gradient (flip = 1) register-slice pack of a Conv2d(64, 64, 3, 1, 1) weight: from the step's one
    multi-tensor launch when planned (the plan learns the layer here), else its own launch."""
    _PackPlan.get(w.device).note(param, w, "c64", "c64")
    return _packed(param, w, "c64d" if flip else "c64f", lambda: _c64_pack(w, flip))


def _c64_pack(w, flip):
    """Register-slice pack of an OIHW [64][64][3][3] weight for csrc/conv_c64.hip (flip = 1: the input-gradient form)."""
    wp = torch.empty(lib.migan_c64_pack_floats(), device=w.device, dtype=torch.float32)
    check(lib.migan_c64_pack(w.data_ptr(), wp.data_ptr(), int(flip), _stream()), "c64_pack")
    return wp


def _rgb_limits_ok(N, H, W, Ho, Wo, R, S, pt, pl, gather):
    """What migan_rgb_conv_fwd / _wgrad refuse beyond migan_rgb_conv_ok(): more images than grid.z holds, and reflection pads that
    reach past the image on either side.  Checked here so that such a layer falls through to the general kernels instead of raising."""
    if N > 65535:
        return False
    if gather == GATHER_REFLECT and (pt >= H or pl >= W or Ho + R - 1 - pt - H >= H or Wo + S - 1 - pl - W >= W):
        return False
    return True


class _Conv2d(Function):
    """y = act(conv2d(gather(x), w) + b); gather folds ReflectionPad2d / ZeroPad2d / Upsample(2) into the loader."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pads, gather, act, slope, mask=None, stats_buf=None, stats_chunks=0, stats_inst=0,
                relu_in=False, handed=False):
        # relu_in: x is the output of a fused conv+ReLU whose backward THIS conv applies to its input gradient (in the epilogue of
        # the dgrad launch where the kernel has one); handed: this conv's own ReLU backward is applied by its consumer (a conv with
        # relu_in, or F.maxpool2(relu_in=True)) - the pair is set up by nn.Sequential for conv, ReLU, conv | MaxPool2d chains
        # (vgg19.features[:18], srgan/models.py:8-15: eight ReLU-backward passes of 90-360 us per SRGAN step)
        if handed and (act != ACT_RELU or mask is not None):
            raise ValueError("conv2d: only a fused ReLU can be handed to the consumer")
        ctx.relu_in, ctx.handed = bool(relu_in), bool(handed)
        xs = to_nhwc(x)
        w_in, b_in = w, b
        w = _plain(w)
        b = _plain(b)
        _check_dev(w)
        N, Ci, H, W = xs.shape
        Co, Ciw, R, S = w.shape
        if Ciw != Ci:
            raise ValueError("conv2d: weight expects %d input channels, got %d" % (Ciw, Ci))
        pt, pl, pb, pr = pads
        HL, WL = (2 * H, 2 * W) if gather == GATHER_UP2 else (H, W)
        if gather == GATHER_REFLECT and (max(pt, pb) >= H or max(pl, pr) >= W):
            raise ValueError("reflection padding must be smaller than the input")
        Ho, Wo = _conv_out(HL, pt, pb, R, stride), _conv_out(WL, pl, pr, S, stride)
        if Ho <= 0 or Wo <= 0:
            raise ValueError("conv2d: empty output")
        ctx.geom = (N, H, W, Ci, Ho, Wo, Co, R, S, stride, pt, pl, pb, pr, gather, act, slope)
        ctx.has_bias = b is not None
        ctx.params = (w_in, b_in)
        ctx.toep = ctx.few = ctx.rgb = False
        if (gather == GATHER_ZERO and mask is None and stats_buf is None and w.is_contiguous()
                and lib.migan_fewpix_ok(N * Ho * Wo, Co, Ci * R * S) == 1):
            # a handful of output pixels against megabytes of weights (inner U-Net levels, pix2pix/models.py:62-67 at batch 1):
            # im2col (<= 64 rows) + the skinny GEMM on the weight as stored - no OHWI pack, no split-K (csrc/fewpix.hip)
            M, K, st = N * Ho * Wo, Ci * R * S, _stream()
            col = torch.empty((M, K), device=xs.device, dtype=torch.float32)
            check(lib.migan_im2col_small(xs.data_ptr(), col.data_ptr(), N, H, W, Ci, Ho, Wo, R, S, stride, pt, pl, st), "im2col_small")
            y = _empty_nhwc((N, Co, Ho, Wo), xs)
            _fewpix_nt(col, w, b, y, M, Co, K, act, slope, st, "fewpix_conv_fwd")
            ctx.few = True
            ctx.save_for_backward(xs, w, y if act != ACT_NONE else None, None, col)
            return y
        ctx.rgb = False
        if (_RGB and mask is None and stats_buf is None and w.is_contiguous() and Ci == 3 and act in (ACT_NONE, ACT_LRELU, ACT_RELU)
                and _rgb_limits_ok(N, H, W, Ho, Wo, R, S, pt, pl, gather)
                and lib.migan_rgb_conv_ok(Ci, Co, R, S, stride, gather, N * Ho * Wo) == 1):
            # image-input layer (3 source channels: srgan/models.py:85, vgg19.features[0], cyclegan/models.py:50): K = R*S*3 as it is
            # on the MFMA units, operands straight from staged image rows (csrc/rgb_conv.hip)
            wk = _packed_perm(w_in, w, "hwio", (2, 3, 1, 0))
            y = _empty_nhwc((N, Co, Ho, Wo), xs)
            check(lib.migan_rgb_conv_fwd(xs.data_ptr(), wk.data_ptr(), _ptr(b), y.data_ptr(), N, H, W, Ci, Ho, Wo, Co, R, S, pt, pl,
                                         gather, act, slope, 0, _stream()), "rgb_conv_fwd")
            ctx.rgb = True
            ctx.save_for_backward(xs, w, y if act != ACT_NONE else None, None, None)
            return y
        if (_C64 and mask is None and stats_buf is None and w.is_contiguous() and act in (ACT_NONE, ACT_LRELU, ACT_RELU)
                and lib.migan_c64_conv_ok(N, H, W, Ci, Co, R, S, stride, pt, pl, pb, pr, gather) == 1):
            # Conv2d(64, 64, 3, 1, 1) on a large map (the residual trunk, srgan/models.py:22-30,47; vgg19.features[2]): the
            # weight-stationary kernel - weights in registers, input rows in an LDS ring (csrc/conv_c64.hip)
            wk = _packed_c64(w_in, w, 0)
            y = _empty_nhwc((N, Co, Ho, Wo), xs)
            check(lib.migan_c64_conv_fwd(xs.data_ptr(), wk.data_ptr(), _ptr(b), y.data_ptr(), N, H, W, act, slope, 0, None, None, None, None, 0, 0.0,
                                         None, _stream()), "c64_conv_fwd")
            ctx.save_for_backward(xs, w, y if act != ACT_NONE else None, None, None)
            return y
        wp = _packed_perm(w_in, w, "ohwi", (0, 2, 3, 1))
        y = _empty_nhwc((N, Co, Ho, Wo), xs)
        if mask is not None:
            mask = _plain(mask)
            if tuple(mask.shape) != (N, Co) or not mask.is_contiguous() or Co % 4 != 0:
                raise ValueError("conv2d: dropout mask must be a contiguous (N, Co) tensor with Co % 4 == 0")
        toep = (_TOEPLITZ and mask is None and stats_buf is None and Co <= 4 and N * Ho * Wo >= _TOEP_MIN_PIXELS
                and lib.migan_thin_toeplitz_ok(Co, R, S, Ci, stride, gather) == 1)
        if toep:  # image-output 7x7 / 9x9 conv: width-Toeplitz expansion onto the MFMA kernels (csrc/thin_toeplitz.hip)
            wtd = _packed(w_in, w, "toep", lambda: _toep_pack(w))
            nb = lib.migan_thin_toeplitz_workspace(N, Ho, W, Co, S)
            ws = _ws(nb, xs)
            check(lib.migan_thin_toeplitz_fwd(xs.data_ptr(), wtd.data_ptr(), _ptr(b), y.data_ptr(), ws.data_ptr(), nb, N, H, W,
                                              Ci, Ho, Wo, Co, R, S, pt, pl, gather, act, slope, _stream()), "thin_toeplitz_fwd")
        elif stats_buf is not None:  # per-tile statistics for the norm layer behind this conv, from the conv epilogue
            check(lib.migan_conv2d_fwd_stats(xs.data_ptr(), wp.data_ptr(), _ptr(b), _ptr(mask), y.data_ptr(), N, H, W, Ci,
                                             Ho, Wo, Co, R, S, stride, pt, pl, gather, act, slope, stats_buf.data_ptr(),
                                             stats_chunks, stats_inst, _stream()), "conv2d_fwd_stats")
        else:  # mask: fused Dropout2d, y = act(conv) * mask[n][co]
            skp, skb = _splitk_ws(xs, N * Ho * Wo, Co, Ci)
            check(lib.migan_conv2d_fwd_ws(xs.data_ptr(), wp.data_ptr(), _ptr(b), _ptr(mask), y.data_ptr(), N, H, W, Ci, Ho,
                                          Wo, Co, R, S, stride, pt, pl, gather, act, slope, skp, skb, _stream()),
                  "conv2d_fwd" if mask is None else "conv2d_dropout_fwd")
        ctx.toep = toep
        ctx.save_for_backward(xs, w, y if (act != ACT_NONE or mask is not None) else None, mask, None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs = ctx.saved_tensors[0]
        ctx.dx_masked = False
        out = _Conv2d._bwd(ctx, dy)
        dx = out[0]
        if ctx.relu_in and dx is not None and not ctx.dx_masked:   # a path without the mask epilogue: the ReLU backward as its own pass
            dx = _ActBwd.apply(dx, xs, ACT_RELU, 0.0) if torch.is_grad_enabled() else _act_bwd_raw(to_nhwc(dx), xs, ACT_RELU, 0.0)
        return (dx,) + tuple(out[1:3]) + (None,) * 11

    @staticmethod
    def _bwd(ctx, dy):
        xs, w, y, mask, col = ctx.saved_tensors
        N, H, W, Ci, Ho, Wo, Co, R, S, stride, pt, pl, pb, pr, gather, act, slope = ctx.geom
        if torch.is_grad_enabled():
            return _Conv2d._backward_differentiable(ctx, dy, xs, w, y, mask)
        dy = to_nhwc(dy)
        if ctx.handed:   # dy arrives through this conv's ReLU already (the consumer's epilogue)
            act = ACT_NONE
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.few:   # few-pixel path (csrc/fewpix.hip): both gradients from skinny GEMMs on the stored weight / the kept im2col
            if act != ACT_NONE:
                dy = _act_bwd_raw(dy, y, act, slope)
            M, K, st = N * Ho * Wo, Ci * R * S, _stream()
            dx = dw = db = None
            if ctx.needs_input_grad[1]:
                slot = _grad_slot(ctx.params[0])
                dwt = torch.empty_like(w) if slot is None else slot
                dbp, dba = None, 0
                if want_db:
                    dbt, dba, db = _bias_out(ctx.params[1], Co, xs)
                    dbp, want_db = dbt.data_ptr(), False
                check(lib.migan_skinny_tn(dy.data_ptr(), col.data_ptr(), dwt.data_ptr(), dbp, M, Co, K, 0 if slot is None else 1,
                                          dba, st), "fewpix_conv_wgrad")
                dw = dwt if slot is None else None
            if want_db:
                db = _colsum(dy, M, Co, _grad_slot(ctx.params[1]))
            if ctx.needs_input_grad[0]:
                ycol = torch.empty((M, K), device=xs.device, dtype=torch.float32)
                check(lib.migan_skinny_nn(dy.data_ptr(), w.data_ptr(), ycol.data_ptr(), M, Co, K, st), "fewpix_conv_dgrad")
                dx = _empty_nhwc((N, Ci, H, W), xs)
                check(lib.migan_col2im_small(ycol.data_ptr(), None, dx.data_ptr(), N, H, W, Ci, Ho, Wo, R, S, stride, pt, pl, 0, 0.0,
                                             st), "col2im_small")
            return dx, dw, db, None, None, None, None, None, None, None, None, None
        if (getattr(ctx, "rgb", False) and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]
                and act in (ACT_NONE, ACT_LRELU, ACT_RELU) and _rgb_limits_ok(N, H, W, Ho, Wo, R, S, pt, pl, gather)
                and lib.migan_rgb_conv_wgrad_ok(Ci, Co, R, S, stride, gather, N * Ho * Wo) == 1):
            # image-input layer, weights only (the discriminator's first conv in its own update, srgan.py:129-141): the activation
            # backward, the bias column sums and the weight gradient in ONE launch that reads dy and y once (csrc/rgb_conv.hip) -
            # they were three passes, one of which wrote a 604 MB gradient only for the next to read it
            fork = _Fork(xs.device, dy.numel(), True)
            with fork:
                slot = _grad_slot(ctx.params[0])
                dw = torch.empty_like(w) if slot is None else slot
                dbt, dba, db = (None, 0, None)
                if want_db:
                    dbt, dba, db = _bias_out(ctx.params[1], Co, xs)
                nb = lib.migan_rgb_conv_wgrad_workspace(Co, R, S)
                ws = _ws(nb, xs)
                check(lib.migan_rgb_conv_wgrad(xs.data_ptr(), dy.data_ptr(), _ptr(y) if act != ACT_NONE else None, dw.data_ptr(),
                                               _ptr(dbt), ws.data_ptr(), nb, N, H, W, Ho, Wo, Co, R, S, pt, pl, gather, act, slope,
                                               0 if slot is None else 1, dba, _stream()), "rgb_conv_wgrad")
                if slot is not None:
                    dw = None
            fork.join((dw, db), (dy, xs, y))
            return None, dw, db, None, None, None, None, None, None, None, None, None
        # bias gradient = column sums of the gradient the wgrad consumes: taken from the kernel that writes that gradient
        # (this conv's activation backward, or the norm layer behind the conv) and reduced inside the wgrad launch
        side = None
        fuse_db = want_db and _COLSUM_FUSE and ctx.needs_input_grad[1] and not ctx.toep
        if mask is not None or act != ACT_NONE:
            if fuse_db:
                dy, side = _act_bwd_colsum(dy, y, mask, N, Ho * Wo, Co, act, slope)
            elif mask is not None:
                g = torch.empty_like(y)
                check(lib.migan_act_bwd_nc(dy.data_ptr(), y.data_ptr(), mask.data_ptr(), g.data_ptr(), N, Ho * Wo, Co, act,
                                           slope, _stream()), "act_bwd_nc")
                dy = g
            else:
                dy = _act_bwd_raw(dy, y, act, slope)
        elif fuse_db:
            side = _colsum_side(dy, Co)
        dx = dw = db = None
        if ctx.toep:
            return _conv2d_backward_toeplitz(ctx, dy, xs, w, want_db)
        fork = _Fork(xs.device, dy.numel(), ctx.needs_input_grad[1])
        # ReflectionPad2d(1)+Conv3x3 with both gradients wanted: the input gradient's main launch goes FIRST, its ring
        # correction (a latency-bound launch of tiny workgroups, 64 us on CycleGAN's R256) runs on the side stream underneath
        # the weight-gradient launch that follows on this stream
        ring = None
        if (_RING_OVERLAP and not fork.on and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and _reflect1_applies(ctx.geom)
                and not torch.cuda.is_current_stream_capturing()):
            dx, ring = _conv2d_dgrad_raw(ctx, dy, xs, w, ring_on_side=True)
        with fork:
            st = _stream()
            if ctx.needs_input_grad[1]:
                slot = _grad_slot(ctx.params[0])
                dw = torch.empty_like(w) if slot is None else slot
                nb = lib.migan_conv2d_wgrad_workspace(N, Ho, Wo, Co, R, S, Ci)
                ws = _ws(nb, xs)
                dbp, dba, sl, nsl = None, 0, None, 0
                if want_db and side is not None:
                    dbt, dba, db = _bias_out(ctx.params[1], Co, xs)
                    dbp, want_db, sl, nsl = dbt.data_ptr(), False, side[0].data_ptr(), side[1]
                elif want_db and _FUSE_BIAS and lib.migan_conv2d_wgrad_fuses_bias(Co, R, S, Ci, stride, gather):
                    dbt, dba, db = _bias_out(ctx.params[1], Co, xs)
                    dbp, want_db = dbt.data_ptr(), False
                if (_C64 and (dbp is None or sl is not None)
                        and lib.migan_c64_conv_ok(N, H, W, Ci, Co, R, S, stride, pt, pl, pb, pr, gather) == 1):
                    # Conv2d(64, 64, 3, 1, 1): accumulators stationary, x and dy read once (csrc/conv_c64.hip c64_wgrad_kernel)
                    nbc = lib.migan_c64_wgrad_workspace(N, H, W)
                    wsc = _ws(nbc, xs)
                    check(lib.migan_c64_conv_wgrad(xs.data_ptr(), dy.data_ptr(), dw.data_ptr(), wsc.data_ptr(), nbc, N, H, W,
                                                   0 if slot is None else 1, dbp, dba, sl, nsl, None, None, None, None, 0, 0.0, None, st),
                          "c64_conv_wgrad")
                else:
                    check(lib.migan_conv2d_wgrad(xs.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H, W, Ci,
                                                 Ho, Wo, Co, R, S, stride, pt, pl, gather, 0 if slot is None else 1,
                                                 dbp, dba, sl, nsl, st), "conv2d_wgrad")
                if slot is not None:
                    dw = None
            if want_db:
                db = _colsum(dy, N * Ho * Wo, Co, _grad_slot(ctx.params[1]))
        if ring is not None:
            ring()   # this stream waits for the ring correction before dx leaves the Function
        elif ctx.needs_input_grad[0]:
            dx = _conv2d_dgrad_raw(ctx, dy, xs, w)
        fork.join((dw, db), (dy, xs, side[0] if side is not None else None))
        return dx, dw, db, None, None, None, None, None, None, None, None, None


_RING_OVERLAP = True
_RGB = __import__("os").environ.get("MIGAN_RGB", "1") == "1"   # A/B knob (round 5): 0 = the general kernels for the 3-channel layers


def _reflect1_applies(geom):
    N, H, W, Ci, Ho, Wo, Co, R, S, stride, pt, pl, pb, pr, gather, act, slope = geom
    return (gather == GATHER_REFLECT and _REFLECT1 and (R, S, stride) == (3, 3, 1) and (pt, pl, pb, pr) == (1, 1, 1, 1)
            and H >= 4 and W >= 4 and Co % 4 == 0 and Co >= 8 and Ci > 4)


def _conv2d_dgrad_raw(ctx, dy, xs, w, ring_on_side=False):
    """Input gradient of _Conv2d on the generic kernels (dy already through the activation backward).
    ring_on_side (reflect-1 geometry only): returns (dx, join) - the pad-1 launch is queued on the current stream, the ring
    correction on the side stream; join() makes the current stream wait for it."""
    N, H, W, Ci, Ho, Wo, Co, R, S, stride, pt, pl, pb, pr, gather, act, slope = ctx.geom
    st = _stream()
    dx = _empty_nhwc((N, Ci, H, W), xs)
    if (gather == GATHER_ZERO and _C64 and not ctx.relu_in and not ring_on_side and w.is_contiguous()
            and lib.migan_c64_conv_ok(N, H, W, Ci, Co, R, S, stride, pt, pl, pb, pr, gather) == 1):
        # the input gradient of Conv2d(64, 64, 3, 1, 1) IS that convolution with the taps reversed and the channel roles swapped
        wk = _packed_c64(ctx.params[0], w, 1)
        check(lib.migan_c64_conv_fwd(dy.data_ptr(), wk.data_ptr(), None, dx.data_ptr(), N, H, W, ACT_NONE, 0.0, 0, None, None, None, None, 0, 0.0, None,
                                     st), "c64_conv_dgrad")
        return dx
    wt = _packed_perm(ctx.params[0], w, "ihwo", (1, 2, 3, 0))
    if ring_on_side:
        skp, skb = _splitk_ws(dy, N * H * W, Ci, Co)
        check(lib.migan_conv2d_dgrad_ws(dy.data_ptr(), wt.data_ptr(), None, dx.data_ptr(), N, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 0, 0.0,
                                        skp, skb, st), "conv2d_dgrad")
        main = torch.cuda.current_stream(xs.device)
        key = (xs.device.index, main.cuda_stream)
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(xs.device)
            ensure_splitk_ws(xs.device, _SIDE_STREAMS[key])
        side = _SIDE_STREAMS[key]
        side.wait_stream(main)
        with torch.cuda.stream(side):   # the ring's split-K workspace is the side stream's own
            skp, skb = _splitk_ws(dy, N * H * W, Ci, Co)
        check(lib.migan_conv2d_dgrad_reflect1_ring_ws(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), N, H, W, Ci, Co, skp, skb,
                                                      side.cuda_stream), "conv2d_dgrad_reflect1_ring")
        # the closure keeps dy / wt alive until the join: they were allocated on `main`, and a tensor dropped before the
        # side-stream launch has run would be handed to the next main-stream allocation (the wgrad workspace) under it
        def join(_keep=(dy, wt, dx)):
            main.wait_stream(side)

        return dx, join
    # (The input gradient of a thin-OUTPUT layer - dcgan.py:62: 64 channels back from 1 or 3 - also runs on the image-input forward kernel
    # with the taps reversed, migan_rgb_conv_fwd(..., flip = 1): 45.5 vs 54.4 us stand-alone, but the captured DCGAN step measured
    # 2.530 / 2.531 ms with it against 2.521 / 2.529 without, profiles/r05_ab.txt call 18 - the host mirror stays on the general kernel.)
    if gather == GATHER_ZERO:
        skp, skb = _splitk_ws(dy, N * -(-H // stride) * -(-W // stride), Ci, Co, stride * stride)
        rc = 801
        if ctx.relu_in:   # dx leaves through the derivative of the ReLU that produced xs, in the launch's epilogue
            rc = lib.migan_conv2d_dgrad_relu_ws(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), xs.data_ptr(), N, H, W, Ci, Ho, Wo, Co, R,
                                                S, stride, pt, pl, skp, skb, st)
            if rc != 801:   # hipErrorNotSupported: a geometry the LDS-DMA kernels do not take
                check(rc, "conv2d_dgrad_relu")
                ctx.dx_masked = True
        if rc == 801:
            check(lib.migan_conv2d_dgrad_ws(dy.data_ptr(), wt.data_ptr(), None, dx.data_ptr(), N, H, W, Ci, Ho, Wo,
                                            Co, R, S, stride, pt, pl, 0, 0.0, skp, skb, st), "conv2d_dgrad")
    elif _reflect1_applies(ctx.geom):
        # ReflectionPad2d(1) + Conv3x3 (cyclegan/models.py:26-35): no padded intermediate, no fold pass
        skp, skb = _splitk_ws(dy, N * H * W, Ci, Co)
        check(lib.migan_conv2d_dgrad_reflect1_ws(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), N, H, W, Ci, Co, skp, skb, st),
              "conv2d_dgrad_reflect1")
    else:
        if gather == GATHER_REFLECT:
            Hp, Wp, gpt, gpl, dpt, dpl = H + pt + pb, W + pl + pr, pt, pl, 0, 0
        else:
            Hp, Wp, gpt, gpl, dpt, dpl = 2 * H, 2 * W, 0, 0, pt, pl
        tmp = _empty_nhwc((N, Ci, Hp, Wp), xs)
        check(lib.migan_conv2d_dgrad(dy.data_ptr(), wt.data_ptr(), None, tmp.data_ptr(), N, Hp, Wp, Ci, Ho, Wo,
                                     Co, R, S, stride, dpt, dpl, 0, 0.0, st), "conv2d_dgrad")
        check(lib.migan_gather2d_bwd(tmp.data_ptr(), dx.data_ptr(), N, H, W, Ci, Hp, Wp, gpt, gpl, gather, st),
              "gather2d_bwd")
    return dx


# A/B knob: 0 = the direct VALU kernels (thin_conv_kernel / thin_wgrad_tile_kernel) for the 7x7 / 9x9 image-output convs
_TOEPLITZ = __import__("os").environ.get("MIGAN_TOEPLITZ", "1") == "1"
_TOEP_MIN_PIXELS = 65536  # below this the two extra streaming launches outweigh the GEMM's gain


def _toep_pack(w):
    """[wt | wd] of migan_thin_toeplitz_pack in one buffer (one packed-weight cache entry)."""
    Co, Ci, R, S = w.shape
    n = lib.migan_thin_toeplitz_cols(Co, S) * R * Ci
    buf = torch.empty(2 * n, device=w.device, dtype=torch.float32)
    check(lib.migan_thin_toeplitz_pack(w.data_ptr(), buf.data_ptr(), buf.data_ptr() + 4 * n, Co, Ci, R, S, _stream()),
          "thin_toeplitz_pack")
    return buf


def _conv2d_backward_toeplitz(ctx, dy, xs, w, want_db):
    """First-order backward of a conv that ran through the width-Toeplitz expansion: dy (already through the activation
    backward) is expanded once into q, which both the weight and the input gradient GEMMs consume."""
    N, H, W, Ci, Ho, Wo, Co, R, S, stride, pt, pl, pb, pr, gather, act, slope = ctx.geom
    st = _stream()
    dx = dw = db = None
    nq = lib.migan_thin_toeplitz_workspace(N, Ho, W, Co, S)
    q = _ws(nq, xs)
    check(lib.migan_thin_toeplitz_expand(dy.data_ptr(), q.data_ptr(), N, Ho, Wo, Co, W, S, pl, gather, st), "thin_toeplitz_expand")
    fork = _Fork(xs.device, dy.numel(), ctx.needs_input_grad[1] or want_db)
    with fork:
        if ctx.needs_input_grad[1]:
            slot = _grad_slot(ctx.params[0])
            dw = torch.empty_like(w) if slot is None else slot
            nb = lib.migan_thin_toeplitz_wgrad_workspace(N, Ho, W, Ci, Co, R, S)
            ws = _ws(nb, xs)
            check(lib.migan_thin_toeplitz_wgrad(xs.data_ptr(), q.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H, W, Ci, Ho, Co, R,
                                                S, pt, gather, 0 if slot is None else 1, _stream()), "thin_toeplitz_wgrad")
            if slot is not None:
                dw = None
        if want_db:
            db = _colsum(dy, N * Ho * Wo, Co, _grad_slot(ctx.params[1]))
    fork.join((dw, db), (dy, xs, q))
    if ctx.needs_input_grad[0] and gather != GATHER_ZERO:
        # reflection padding: the expansion's input gradient needs a row-padded intermediate + fold and measured slower
        # than the direct kernel (c7s1-3: 294 vs 208 us, profiles/r02_conv_microbench.txt)
        dx = _conv2d_dgrad_raw(ctx, dy, xs, w)
    elif ctx.needs_input_grad[0]:
        wtd = _packed(ctx.params[0], w, "toep", lambda: _toep_pack(w))
        n = lib.migan_thin_toeplitz_cols(Co, S) * R * Ci
        dx = _empty_nhwc((N, Ci, H, W), xs)
        nb = lib.migan_thin_toeplitz_dgrad_workspace(N, H, W, Ci, Ho, R, gather)
        ws = _ws(nb, xs)
        check(lib.migan_thin_toeplitz_dgrad(q.data_ptr(), wtd.data_ptr() + 4 * n, dx.data_ptr(), ws.data_ptr(), nb, N, H, W, Ci,
                                            Ho, Co, R, S, pt, gather, st), "thin_toeplitz_dgrad")
    return dx, dw, db, None, None, None, None, None, None, None, None, None


_REFLECT1 = True   # False = padded extent + fold pass


def _conv2d_backward_differentiable(ctx, dy, xs, w, y, mask):
    """Backward of _Conv2d recorded as differentiable ops (autograd.grad(..., create_graph=True), the conv-critic gradient
    penalties of dragan.py:144-167 / stargan.py:142-161): activation' and the Dropout2d mask through _ActBwd / _MulMask,
    the input gradient through _ConvDgradFn.  Weight and bias gradients of THIS backward are produced by the ordinary
    kernels (not differentiable again: the reference never differentiates a weight gradient) and skipped entirely inside
    `input_grad_only()`."""
    N, H, W, Ci, Ho, Wo, Co, R, S, stride, pt, pl, pb, pr, gather, act, slope = ctx.geom
    if gather != GATHER_ZERO:
        raise NotImplementedError("double backward through a reflection-padded / upsampled conv is not on the reference path")
    g = dy
    if act != ACT_NONE and not ctx.handed:
        # dx = dy * act'(y): y is the (masked) layer output; where the Dropout2d mask is 0 the product with the mask below
        # is 0 whatever act' evaluates to
        g = _ActBwd.apply(g, y, act, slope)
    if mask is not None:
        g = _MulMask.apply(g, mask)
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        dx = _ConvDgradFn.apply(g, ctx.params[0], (N, H, W, Ci, Ho, Wo, Co, R, S, stride, pt, pl))
    if not _INPUT_GRAD_ONLY:
        with torch.no_grad():
            gd = to_nhwc(g.detach())
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(w)
                nb = lib.migan_conv2d_wgrad_workspace(N, Ho, Wo, Co, R, S, Ci)
                ws = _ws(nb, xs)
                check(lib.migan_conv2d_wgrad(xs.data_ptr(), gd.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H, W, Ci, Ho,
                                             Wo, Co, R, S, stride, pt, pl, gather, 0, None, 0, None, 0, _stream()),
                      "conv2d_wgrad")
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = _colsum(gd, N * Ho * Wo, Co, None)
    return dx, dw, db, None, None, None, None, None, None, None, None, None


_Conv2d._backward_differentiable = staticmethod(_conv2d_backward_differentiable)


# Statistics of the following norm layer from the conv epilogue: implemented, parity-tested, and measured SLOWER than the
# norm layer's own statistics pass (profiles/r02_ab.txt: DCGAN step 3.93 -> 4.12 ms, CycleGAN 173.3 -> 174.5 ms; the
# two-pass per-tile reduction in the epilogue of every conv workgroup costs more than one streaming pass at 5 TB/s saves),
# so it is off (functional._CONV_STATS; the parity tests of the epilogue flip it).
_CONV_STATS = False


def _attach_stats(y, buf, chunks, inst, G, P, C):
    y._migan_stats = (buf, chunks, inst, G, P, C, y.data_ptr(), y._version)
    return y


def _stats_side(x, inst, G, P, C):
    """(buffer, chunks) when the conv that produced `x` left its per-tile statistics for exactly this normalisation."""
    side = getattr(x, "_migan_stats", None)
    if side is None or not _CONV_STATS:
        return None
    buf, chunks, sinst, sG, sP, sC, ptr, ver = side
    if (sinst, sG, sP, sC) != (int(inst), G, P, C) or ptr != x.data_ptr() or ver != x._version:
        return None
    return buf, chunks


def conv2d(x, w, b=None, stride=1, pads=(0, 0, 0, 0), gather=GATHER_ZERO, act=ACT_NONE, slope=0.0, dropout_mask=None,
           stats=None, relu_in=False, handed=False):
    """`dropout_mask` (N, Co), already scaled by 1/(1-p): fuses a following nn.Dropout2d into the conv epilogue.
    `stats` ("batch" | "instance"): the conv epilogue also leaves per-tile statistics of its output for the BatchNorm /
    InstanceNorm layer that follows (picked up by `norm()`; ignored where the geometry does not support it)."""
    stride, pads = int(stride), tuple(int(p) for p in pads)
    if stats is not None and _CONV_STATS and x.dim() == 4:
        inst = 1 if stats == "instance" else 0
        N, Ci, H, W = x.shape
        Co, _, R, S = w.shape
        HL, WL = (2 * H, 2 * W) if gather == GATHER_UP2 else (H, W)
        Ho, Wo = _conv_out(HL, pads[0], pads[2], R, stride), _conv_out(WL, pads[1], pads[3], S, stride)
        chunks = lib.migan_conv2d_stats_chunks(N, H, W, Ci, Ho, Wo, Co, R, S, stride, pads[0], pads[1], int(gather), inst) \
            if Ho > 0 and Wo > 0 else 0
        if chunks > 0:
            G = N if inst else 1
            buf = torch.empty(G * chunks * Co * 3, device=x.device, dtype=torch.float32)
            y = _Conv2d.apply(x, w, b, stride, pads, int(gather), int(act), float(slope), dropout_mask, buf, chunks, inst,
                              bool(relu_in), bool(handed))
            return _attach_stats(y, buf, chunks, inst, G, (Ho * Wo) if inst else N * Ho * Wo, Co)
    return _Conv2d.apply(x, w, b, stride, pads, int(gather), int(act), float(slope), dropout_mask, None, 0, 0, bool(relu_in),
                         bool(handed))


class _UpConv3x3(Function):
    """Upsample(scale_factor=2) -> Conv2d(3x3, stride 1, padding 1) in its phase-collapsed form (2.25x fewer FLOPs)."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope, stats_buf=None, stats_chunks=0, stats_inst=0):
        xs = to_nhwc(x)
        ctx.params = (w, b)
        w, b = _plain(w), _plain(b)
        N, Ci, H, W = xs.shape
        Co = w.shape[0]
        if tuple(w.shape) != (Co, Ci, 3, 3):
            raise ValueError("upconv3x3: weight must be (Co, %d, 3, 3)" % Ci)
        wc = w if w.is_contiguous() else w.contiguous()
        wf = torch.empty(Co * 16 * Ci, device=xs.device, dtype=torch.float32)
        wd = torch.empty_like(wf)
        st = _stream()
        check(lib.migan_upconv3x3_pack(wc.data_ptr(), wf.data_ptr(), wd.data_ptr(), Co, Ci, st), "upconv_pack")
        y = _empty_nhwc((N, Co, 2 * H, 2 * W), xs)
        if stats_buf is not None:
            check(lib.migan_upconv3x3_fwd_stats(xs.data_ptr(), wf.data_ptr(), _ptr(b), y.data_ptr(), N, H, W, Ci, Co, act,
                                                slope, stats_buf.data_ptr(), stats_chunks, stats_inst, st), "upconv_fwd_stats")
        else:
            check(lib.migan_upconv3x3_fwd(xs.data_ptr(), wf.data_ptr(), _ptr(b), y.data_ptr(), N, H, W, Ci, Co, act, slope,
                                          st), "upconv_fwd")
        ctx.geom = (N, H, W, Ci, Co, act, slope)
        ctx.has_bias = b is not None
        ctx.save_for_backward(xs, w, wd, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only('Upsample+Conv3x3')
        xs, w, wd, y = ctx.saved_tensors
        N, H, W, Ci, Co, act, slope = ctx.geom
        dy = to_nhwc(dy)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        side = None
        fuse_db = want_db and _COLSUM_FUSE and ctx.needs_input_grad[1]
        if act != ACT_NONE:
            if fuse_db:
                dy, side = _act_bwd_colsum(dy, y, None, N, 4 * H * W, Co, act, slope)
            else:
                dy = _act_bwd_raw(dy, y, act, slope)
        elif fuse_db:
            side = _colsum_side(dy, Co)
        dx = dw = db = None
        fork = _Fork(xs.device, dy.numel(), ctx.needs_input_grad[1])
        with fork:
            st = _stream()
            if ctx.needs_input_grad[1]:
                slot = _grad_slot(ctx.params[0])
                dw = torch.empty_like(w) if slot is None else slot
                acc = 0 if slot is None else 1
                dbp, dba, sl, nsl = None, 0, None, 0
                if want_db and side is not None:
                    dbt, dba, db = _bias_out(ctx.params[1], Co, xs)
                    dbp, want_db, sl, nsl = dbt.data_ptr(), False, side[0].data_ptr(), side[1]
                if Co % 4 == 0 and Ci % 4 == 0:
                    nb = lib.migan_upconv3x3_wgrad_workspace(N, H, W, Co, Ci)
                    ws = _ws(nb, xs)
                    if want_db and _FUSE_BIAS:
                        dbt, dba, db = _bias_out(ctx.params[1], Co, xs)
                        dbp, want_db = dbt.data_ptr(), False
                    check(lib.migan_upconv3x3_wgrad(xs.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H,
                                                    W, Ci, Co, acc, dbp, dba, sl, nsl, st), "upconv_wgrad")
                else:  # same gradient through the dense gathered wgrad
                    nb = lib.migan_conv2d_wgrad_workspace(N, 2 * H, 2 * W, Co, 3, 3, Ci)
                    ws = _ws(nb, xs)
                    check(lib.migan_conv2d_wgrad(xs.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H, W,
                                                 Ci, 2 * H, 2 * W, Co, 3, 3, 1, 1, 1, GATHER_UP2, acc, dbp, dba, sl, nsl,
                                                 st), "conv2d_wgrad")
                if slot is not None:
                    dw = None
            if want_db:
                db = _colsum(dy, N * 4 * H * W, Co, _grad_slot(ctx.params[1]))
        if ctx.needs_input_grad[0]:
            dx = _empty_nhwc((N, Ci, H, W), xs)
            check(lib.migan_upconv3x3_dgrad(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), N, H, W, Ci, Co, _stream()),
                  "upconv_dgrad")
        fork.join((dw, db), (dy, xs, side[0] if side is not None else None))
        return dx, dw, db, None, None, None, None, None


_UPCONV_COLLAPSE = True


def set_upconv_collapse(enabled):
    """Use the phase-collapsed kernels for Upsample(2)->Conv3x3(p=1) (default) or the dense gathered conv."""
    global _UPCONV_COLLAPSE
    _UPCONV_COLLAPSE = bool(enabled)


def upconv3x3(x, w, b=None, act=ACT_NONE, slope=0.0, stats=None):
    if _UPCONV_COLLAPSE:
        if stats is not None and _CONV_STATS and x.dim() == 4:
            inst = 1 if stats == "instance" else 0
            N, Ci, H, W = x.shape
            Co = w.shape[0]
            chunks = lib.migan_upconv3x3_stats_chunks(N, H, W, Ci, Co, inst)
            if chunks > 0:
                G = N if inst else 1
                buf = torch.empty(G * chunks * Co * 3, device=x.device, dtype=torch.float32)
                y = _UpConv3x3.apply(x, w, b, int(act), float(slope), buf, chunks, inst)
                return _attach_stats(y, buf, chunks, inst, G, (4 * H * W) if inst else N * 4 * H * W, Co)
        return _UpConv3x3.apply(x, w, b, int(act), float(slope))
    return conv2d(x, w, b, 1, (1, 1, 1, 1), GATHER_UP2, act, slope, None, stats)


class _ConvTranspose2d(Function):
    """nn.ConvTranspose2d(Cin, Cout, k, s, p) forward == dgrad of the (Cout -> Cin) conv (pix2pix/models.py:39)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, slope):
        xs = to_nhwc(x)
        ctx.params = (w, b)
        w = _plain(w)
        b = _plain(b)
        N, Cin, Hin, Win = xs.shape
        Cinw, Cout, R, S = w.shape
        if Cinw != Cin:
            raise ValueError("conv_transpose2d: weight expects %d input channels, got %d" % (Cinw, Cin))
        Hout = (Hin - 1) * stride - 2 * pad + R
        Wout = (Win - 1) * stride - 2 * pad + S
        ctx.geom = (N, Cin, Hin, Win, Cout, Hout, Wout, R, S, stride, pad, act, slope)
        ctx.has_bias = b is not None
        ctx.few = False
        if w.is_contiguous() and lib.migan_fewpix_ok(N * Hin * Win, Cin, Cout * R * S) == 1:
            # a handful of INPUT pixels against megabytes of weights (pix2pix/models.py:68-71 at batch 1): ycol = x W on the weight
            # as stored ([Cin][Cout*R*S]), then the col2im sum with bias and activation (csrc/fewpix.hip) - no IHWO pack, no split-K
            M, K, st = N * Hin * Win, Cout * R * S, _stream()
            ycol = torch.empty((M, K), device=xs.device, dtype=torch.float32)
            check(lib.migan_skinny_nn(xs.data_ptr(), w.data_ptr(), ycol.data_ptr(), M, Cin, K, st), "fewpix_convT_fwd")
            y = _empty_nhwc((N, Cout, Hout, Wout), xs)
            check(lib.migan_col2im_small(ycol.data_ptr(), _ptr(b), y.data_ptr(), N, Hout, Wout, Cout, Hin, Win, R, S, stride, pad, pad,
                                         act, slope, st), "col2im_small")
            ctx.few = True
            ctx.save_for_backward(xs, w, y if act != ACT_NONE else None)
            return y
        # [Cout][R][S][Cin] == w_ihwo of the transposed-role conv
        wp = _packed_perm(ctx.params[0], w, "ihwo", (1, 2, 3, 0))
        y = _empty_nhwc((N, Cout, Hout, Wout), xs)
        skp, skb = _splitk_ws(xs, N * -(-Hout // stride) * -(-Wout // stride), Cout, Cin, stride * stride)
        check(lib.migan_conv2d_dgrad_ws(xs.data_ptr(), wp.data_ptr(), _ptr(b), y.data_ptr(), N, Hout, Wout, Cout, Hin,
                                        Win, Cin, R, S, stride, pad, pad, act, slope, skp, skb, _stream()), "convT_fwd")
        ctx.save_for_backward(xs, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only('ConvTranspose2d')
        xs, w, y = ctx.saved_tensors
        N, Cin, Hin, Win, Cout, Hout, Wout, R, S, stride, pad, act, slope = ctx.geom
        dy = to_nhwc(dy)
        if act != ACT_NONE:
            dy = _act_bwd_raw(dy, y, act, slope)
        dx = dw = db = None
        if ctx.few:   # few-pixel path: dycol = im2col(dy) once, then dx = dycol W^T and dW (+)= x^T dycol on the stored weight
            M, K, st = N * Hin * Win, Cout * R * S, _stream()
            dycol = torch.empty((M, K), device=xs.device, dtype=torch.float32)
            check(lib.migan_im2col_small(dy.data_ptr(), dycol.data_ptr(), N, Hout, Wout, Cout, Hin, Win, R, S, stride, pad, pad, st),
                  "im2col_small")
            if ctx.needs_input_grad[1]:
                slot = _grad_slot(ctx.params[0])
                dwt = torch.empty_like(w) if slot is None else slot
                check(lib.migan_skinny_tn(xs.data_ptr(), dycol.data_ptr(), dwt.data_ptr(), None, M, Cin, K, 0 if slot is None else 1,
                                          0, st), "fewpix_convT_wgrad")
                dw = dwt if slot is None else None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = _colsum(dy, N * Hout * Wout, Cout, _grad_slot(ctx.params[1]))
            if ctx.needs_input_grad[0]:
                dx = _empty_nhwc((N, Cin, Hin, Win), xs)
                _fewpix_nt(dycol, w, None, dx, M, Cin, K, ACT_NONE, 0.0, st, "fewpix_convT_dgrad")
            return dx, dw, db, None, None, None, None
        fork = _Fork(xs.device, dy.numel(), ctx.needs_input_grad[1])
        with fork:
            st = _stream()
            if ctx.needs_input_grad[1]:
                slot = _grad_slot(ctx.params[0])
                dw = torch.empty_like(w) if slot is None else slot
                nb = lib.migan_conv2d_wgrad_workspace(N, Hin, Win, Cin, R, S, Cout)
                ws = _ws(nb, xs)
                check(lib.migan_conv2d_wgrad(dy.data_ptr(), xs.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, Hout,
                                             Wout, Cout, Hin, Win, Cin, R, S, stride, pad, pad, GATHER_ZERO,
                                             0 if slot is None else 1, None, 0, None, 0, st), "convT_wgrad")
                if slot is not None:
                    dw = None
            if ctx.has_bias and ctx.needs_input_grad[2]:  # (the reference's ConvTranspose2d layers have bias=False)
                db = _colsum(dy, N * Hout * Wout, Cout, _grad_slot(ctx.params[1]))
        st = _stream()
        if ctx.needs_input_grad[0]:
            # [Cin][R][S][Cout]: OHWI of the conv Cout->Cin
            wo = _packed_perm(ctx.params[0], w, "ohwi", (0, 2, 3, 1))
            dx = _empty_nhwc((N, Cin, Hin, Win), xs)
            skp, skb = _splitk_ws(dy, N * Hin * Win, Cin, Cout)
            check(lib.migan_conv2d_fwd_ws(dy.data_ptr(), wo.data_ptr(), None, None, dx.data_ptr(), N, Hout, Wout, Cout, Hin,
                                          Win, Cin, R, S, stride, pad, pad, GATHER_ZERO, 0, 0.0, skp, skb, st), "convT_dgrad")
        fork.join((dw, db), (dy, xs))
        return dx, dw, db, None, None, None, None


def conv_transpose2d(x, w, b=None, stride=2, pad=1, act=ACT_NONE, slope=0.0):
    return _ConvTranspose2d.apply(x, w, b, int(stride), int(pad), int(act), float(slope))


# ---------------------------------------------------------------------------------------------- GEMM primitives
def _mm_nt_raw(a, b, bias, act=ACT_NONE, slope=0.0):
    M, K = a.shape
    Nn, Kb = b.shape
    if K != Kb:
        raise ValueError("mm_nt: inner dimensions differ (%d vs %d)" % (K, Kb))
    out = torch.empty((M, Nn), device=a.device, dtype=torch.float32)
    if _SKINNY and lib.migan_skinny_nt_ok(M, Nn, K):  # <= 64 rows: latency-bound, MFMA 16x16x4 straight from L2
        check(lib.migan_skinny_nt(a.data_ptr(), b.data_ptr(), _ptr(bias), out.data_ptr(), M, Nn, K, act, slope, _stream()),
              "skinny_nt")
        return out
    check(lib.migan_conv2d_fwd(a.data_ptr(), b.data_ptr(), _ptr(bias), out.data_ptr(), M, 1, 1, K, 1, 1, Nn, 1, 1, 1,
                               0, 0, GATHER_ZERO, act, slope, _stream()), "mm_nt")
    return out


_SKINNY = True


class _MMNN(Function):
    """a[M,R] @ b[R,N] — the Linear input gradient dy @ W with W in its stored layout.  <= 64 rows: skinny kernel, no
    transposed copy of W; otherwise transpose + the tiled kernel.  Backward is built from differentiable Functions."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        ac, bc = canon(a), canon(b)
        M, R = ac.shape
        Rb, Nn = bc.shape
        if R != Rb:
            raise ValueError("mm_nn: inner dimensions differ (%d vs %d)" % (R, Rb))
        if _SKINNY and lib.migan_skinny_nn_ok(M, R, Nn):
            out = torch.empty((M, Nn), device=ac.device, dtype=torch.float32)
            check(lib.migan_skinny_nn(ac.data_ptr(), bc.data_ptr(), out.data_ptr(), M, R, Nn, _stream()), "skinny_nn")
            return out
        if R == 1 or Nn == 1:
            bt = bc.reshape(Nn, R)   # the transpose of a row / column vector is the same memory (Linear(K, 1): dcgan.py:92)
        else:
            bt = torch.empty((Nn, R), device=bc.device, dtype=torch.float32)
            check(lib.migan_transpose_batched(bc.data_ptr(), bt.data_ptr(), 1, R, Nn, _stream()), "transpose")
        return _mm_nt_raw(ac, bt, None)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = db = None
        if ctx.needs_input_grad[0]:
            da = _MMNT.apply(g, b, None, ACT_NONE, 0.0)   # [M,N] @ [R,N]^T
        if ctx.needs_input_grad[1]:
            db = _MMTN.apply(a, g)         # [M,R]^T @ [M,N]
        return da, db


class _Transpose(Function):
    @staticmethod
    def forward(ctx, a):
        a = canon(a)
        R, C = a.shape
        out = torch.empty((C, R), device=a.device, dtype=torch.float32)
        check(lib.migan_transpose_batched(a.data_ptr(), out.data_ptr(), 1, R, C, _stream()), "transpose")
        return out

    @staticmethod
    def backward(ctx, g):
        return _Transpose.apply(g)


class _MMNT(Function):
    """a[M,K] @ b[N,K]^T + bias[N]  — nn.Linear forward (conv kernel, 1x1 geometry)."""

    @staticmethod
    def forward(ctx, a, b, bias, act=ACT_NONE, slope=0.0):
        # save the ORIGINAL inputs: their autograd history is what makes the backward differentiable again
        ctx.has_bias = bias is not None
        ctx.bias_param = bias
        ctx.act, ctx.slope = act, slope
        y = _mm_nt_raw(canon(a), canon(b), _plain(bias), act, slope)
        ctx.save_for_backward(a, b, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, g):
        a, b, y = ctx.saved_tensors
        if ctx.act != ACT_NONE:  # fused activation epilogue (Linear -> LeakyReLU / Tanh / Sigmoid): differentiable act'
            g = _ActBwd.apply(g, y, ctx.act, ctx.slope)
        da = db = dbias = None
        fork = _Fork(g.device, g.numel(), ctx.needs_input_grad[1])
        gc, ac = canon(g), canon(a)  # on the main stream: both branches read them
        with fork:
            want_db = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                slot = _grad_slot(b)
                if slot is not None:  # first-order backward of a Linear weight: reduce straight into weight.grad
                    dbuf = None
                    # few rows (the MLP critic at batch 64): the bias gradient inside the wgrad launch saves two latency-bound
                    # column-sum launches per layer; for large P it stays opt-in (the column-0 workgroups become the tail)
                    if want_db and (_FUSE_BIAS or gc.shape[0] <= 256) and lib.migan_conv2d_wgrad_fuses_bias(
                            gc.shape[1], 1, 1, ac.shape[1], 1, GATHER_ZERO):
                        dbt, dba, dbias = _bias_out(ctx.bias_param, gc.shape[1], gc)
                        dbuf, want_db = (dbt, dba), False
                    _mm_tn_raw(gc, ac, slot, 1, dbuf)
                else:
                    db = _MMTN.apply(g, a)
            if want_db:
                dbias = _colsum(gc, gc.shape[0], gc.shape[1], _grad_slot(ctx.bias_param))
        if ctx.needs_input_grad[0]:
            da = mm_nn(g, b)
        fork.join((db, dbias), (gc, ac))
        return da, db, dbias, None, None


def _mm_tn_raw(a, b, out=None, accumulate=0, dbuf=None):
    """a[P,M]^T @ b[P,N] -> out[M,N]; dbuf = (tensor[M], accumulate): also the column sums of `a` (a Linear's bias grad)."""
    P, M = a.shape
    Pb, Nn = b.shape
    if P != Pb:
        raise ValueError("mm_tn: row counts differ")
    if out is None:
        out = torch.empty((M, Nn), device=a.device, dtype=torch.float32)
    if _SKINNY and lib.migan_skinny_tn_ok(P, M, Nn):  # <= 64 rows: one direct launch, weight AND bias gradient
        check(lib.migan_skinny_tn(a.data_ptr(), b.data_ptr(), out.data_ptr(), dbuf[0].data_ptr() if dbuf else None, P, M, Nn,
                                  accumulate, dbuf[1] if dbuf else 0, _stream()), "skinny_tn")
        return out
    nb = lib.migan_conv2d_wgrad_workspace(P, 1, 1, M, 1, 1, Nn)
    ws = _ws(nb, a)
    check(lib.migan_conv2d_wgrad(b.data_ptr(), a.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, P, 1, 1, Nn, 1, 1,
                                 M, 1, 1, 1, 0, 0, GATHER_ZERO, accumulate, dbuf[0].data_ptr() if dbuf else None,
                                 dbuf[1] if dbuf else 0, None, 0, _stream()), "mm_tn")
    return out


class _MMTN(Function):
    """a[P,M]^T @ b[P,N] -> [M,N]  — Linear weight gradient (split-K wgrad kernel)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)  # originals (see _MMNT.forward)
        return _mm_tn_raw(canon(a), canon(b))

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = db = None
        if ctx.needs_input_grad[0]:
            da = _MMNT.apply(b, g, None, ACT_NONE, 0.0)  # [P,N] @ [M,N]^T
        if ctx.needs_input_grad[1]:
            db = mm_nn(a, g)  # [P,M] @ [M,N]
        return da, db


def mm_nt(a, b, bias=None):
    return _MMNT.apply(a, b, bias, ACT_NONE, 0.0)


def mm_nn(a, b):
    return _MMNN.apply(a, b)


def linear(x, w, b=None, act=ACT_NONE, slope=0.0):
    """`act`: a following LeakyReLU / ReLU / Tanh / Sigmoid in the GEMM epilogue (wgan_gp.py:46-56,73-77)."""
    if x.dim() != 2:
        raise ValueError("linear: expected a 2-D input (the reference only feeds (B, features))")
    return _MMNT.apply(x, w, b, int(act), float(slope))


# ---------------------------------------------------------------------------------------------- activations
class _ActBwd(Function):
    @staticmethod
    def forward(ctx, g, y, act, slope):
        g, y = canon(g), canon(y)
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(g, y)
        return _act_bwd_raw(g, y, act, slope)

    @staticmethod
    def backward(ctx, gg):
        g, y = ctx.saved_tensors
        gy = None
        if ctx.needs_input_grad[1] and ctx.act in (ACT_TANH, ACT_SIGMOID):
            # dx = g * f'(y) also depends on y (dragan.py:92 ends the critic in a Sigmoid): d/dy = gg * g * f''
            ggc = canon(gg)
            gy = torch.empty_like(y)
            check(lib.migan_act_bwd2(g.data_ptr(), ggc.data_ptr(), y.data_ptr(), gy.data_ptr(), y.numel(), ctx.act,
                                     _stream()), "act_bwd2")
        return _ActBwd.apply(gg, y, ctx.act, ctx.slope), gy, None, None


class _Act(Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        xs = canon(x)
        y = torch.empty_like(xs)
        check(lib.migan_act_fwd(xs.data_ptr(), y.data_ptr(), xs.numel(), act, slope, _stream()), "act_fwd")
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return _ActBwd.apply(g, y, ctx.act, ctx.slope), None, None


def activation(x, act, slope=0.0):
    return _Act.apply(x, int(act), float(slope))


class _PReLU(Function):
    @staticmethod
    def forward(ctx, x, a):
        xs = canon(x)
        ctx.param = a
        a = _plain(a)
        if a.numel() != 1:
            raise ValueError("PReLU: only the single shared slope of nn.PReLU() is supported")
        y = torch.empty_like(xs)
        check(lib.migan_prelu_fwd(xs.data_ptr(), a.data_ptr(), y.data_ptr(), xs.numel(), _stream()), "prelu_fwd")
        ctx.save_for_backward(xs, a)
        return y

    @staticmethod
    def backward(ctx, g):
        _first_order_only('PReLU')
        xs, a = ctx.saved_tensors
        g = canon(g)
        dx = torch.empty_like(xs)
        da = torch.empty_like(a)
        ws = _ws(lib.migan_reduce_workspace(), xs)
        check(lib.migan_prelu_bwd(xs.data_ptr(), g.data_ptr(), a.data_ptr(), dx.data_ptr(), da.data_ptr(),
                                  ws.data_ptr(), xs.numel(), _stream()), "prelu_bwd")
        slot = _grad_slot(ctx.param) if ctx.needs_input_grad[1] else None
        if slot is not None:   # into the optimiser's bucket with the library's add (autograd's AccumulateGrad would launch an ATen add_)
            check(lib.migan_axpby(slot.data_ptr(), 1.0, da.data_ptr(), 1.0, slot.data_ptr(), 1, _stream()), "prelu dweight")
            da = None
        return dx, da


def prelu(x, a):
    return _PReLU.apply(x, a)


# ---------------------------------------------------------------------------------------------- normalisation
# Cross-replica BatchNorm (data parallel): set by dp.DataParallel.enable_sync_batchnorm().  An object with
#   world, all_gather(tensor[K]) -> tensor[world*K], all_reduce_sum(tensor) (in place)
# None: BatchNorm statistics are those of the local batch (the only mode for world size 1).
_SYNC_BN = None


def set_sync_batchnorm(sync):
    global _SYNC_BN
    _SYNC_BN = sync


class _Norm(Function):
    """BatchNorm (G=1) / InstanceNorm (G=N) + fused activation + optional residual add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, running_mean, running_var, use_batch_stats, momentum, eps, instance, act,
                slope, nbt=None, prelu=None, shuffle=0, mask=None):
        # mask: the nn.Dropout behind the activation (pix2pix/models.py:25-28,41-45) as an NHWC multiplier of the output's shape,
        # applied inside the launch - only the one-launch small-tensor route takes it (nn.Sequential checks norm_small_takes())
        ctx.mask = None
        # prelu: the weight of an nn.PReLU() (one shared slope) behind the norm layer (srgan/models.py:23-24,55-57): applied in
        # the norm's apply launch, differentiated inside the norm's backward launches (csrc/norm.hip)
        # shuffle = 2: nn.PixelShuffle(2) between the two (srgan/models.py:55-57): the output is written - and its gradient
        # read - through the shuffle's index map; no shuffled copy exists in either direction
        ctx.prelu = prelu
        pw = _plain(prelu)
        if pw is not None and (pw.numel() != 1 or act != ACT_NONE):
            raise ValueError("norm: fused PReLU needs num_parameters == 1 and no other fused activation")
        if shuffle not in (0, 2) or (shuffle and (x.dim() != 4 or instance or res is not None or act != ACT_NONE
                                                 or x.shape[1] % 4 != 0)):
            raise ValueError("norm: fused PixelShuffle is upscale_factor 2 behind a BatchNorm2d with C % 4 == 0")
        ctx.shuffle = None
        xs = canon(x)
        # an NCHW-contiguous input (e.g. `out.view(B, 128, s, s)`, dcgan.py:68) gets its gradient back in NCHW through
        # the HIP transpose, instead of autograd's ViewBackward materialising it with an ATen strided copy
        ctx.x_nchw = x.dim() == 4 and x.is_contiguous() and not x.is_contiguous(memory_format=CL)
        ctx.params = (gamma, beta)
        gamma, beta = _plain(gamma), _plain(beta)
        if xs.dim() == 4:
            N, C, H, W = xs.shape
            G, P = (N, H * W) if instance else (1, N * H * W)
        elif xs.dim() == 2:
            G, P, C = 1, xs.shape[0], xs.shape[1]
        else:
            raise ValueError("norm: expected 2-D or 4-D input")
        st = _stream()
        sync = _SYNC_BN if (use_batch_stats and not instance and _SYNC_BN is not None and _SYNC_BN.world > 1) else None
        # batch_groups(k): this BatchNorm call stands for k calls on k consecutive sub-batches (statistics, running-statistics
        # updates and num_batches_tracked per sub-batch, in order) - [G = k][P / k][C] in the kernels' group view
        ctx.bn_groups = 1
        if _BN_GROUPS > 1 and use_batch_stats and not instance:
            if sync is not None or shuffle or pw is not None or xs.shape[0] % _BN_GROUPS != 0:
                raise NotImplementedError("batch_groups: plain BatchNorm on a batch divisible by the group count only")
            ctx.bn_groups = G = _BN_GROUPS
            P //= G
        if use_batch_stats:
            if P <= 1 and not instance and sync is None:
                raise ValueError("Expected more than 1 value per channel when training")
            mean = torch.empty(G * C, device=xs.device, dtype=torch.float32)
            invstd = torch.empty_like(mean)
            nb = lib.migan_norm_workspace(G, P, C)
            ws = _ws(nb, xs)
            side = _stats_side(x, instance, G, P, C) if (sync is None and ctx.bn_groups == 1) else None
            if (instance and side is None and sync is None and pw is None and not shuffle and running_mean is None
                    and lib.migan_norm_small_ok(G, P, C)):
                # small instance-normalised tensor (inner U-Net levels, PatchGAN at batch 1): statistics, normalisation,
                # activation and residual in ONE launch (csrc/norm.hip norm_small_fwd_kernel)
                rs = canon(res) if res is not None else None
                y = torch.empty_like(xs)
                mk = None
                if mask is not None:
                    mk = canon(mask)
                    if mk.shape != xs.shape:
                        raise ValueError("norm: dropout mask of shape %s for an output of shape %s" % (tuple(mk.shape), tuple(xs.shape)))
                    ctx.mask = mk
                check(lib.migan_norm_fwd_small(xs.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma),
                                               _ptr(beta), _ptr(rs), _ptr(mk), G, P, C, act, slope, eps, st), "norm_fwd_small")
                ctx.cfg = (G, P, C, act, slope, use_batch_stats, gamma is not None, res is not None)
                ctx.sync = sync
                ctx.save_for_backward(xs, gamma, beta, mean, invstd, x)
                return y
            if mask is not None:
                raise ValueError("norm: a fused dropout mask needs the small-tensor route (norm_small_takes())")
            if side is not None:  # per-tile (mean, M2, count) left by the conv epilogue: no pass over the tensor
                check(lib.migan_norm_stats_from_conv(side[0].data_ptr(), side[1], mean.data_ptr(), invstd.data_ptr(),
                                                     _ptr(running_mean), _ptr(running_var), _ptr(nbt), momentum, eps, G, C,
                                                     st), "norm_stats_from_conv")
            elif sync is None:
                check(lib.migan_norm_stats(xs.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(running_mean),
                                           _ptr(running_var), _ptr(nbt), momentum, eps, G, P, C, ws.data_ptr(), nb,
                                           st), "norm_stats")
            else:
                # statistics of the GLOBAL batch, as the single-process reference computes them: local moments ->
                # all_gather of 2*C floats -> Chan combination (+ running statistics) on every rank
                mom = torch.empty(2 * C, device=xs.device, dtype=torch.float32)
                check(lib.migan_norm_moments(xs.data_ptr(), mom.data_ptr(), mom.data_ptr() + 4 * C, 1, P, C,
                                             ws.data_ptr(), nb, st), "norm_moments")
                allm = sync.all_gather(mom)
                check(lib.migan_norm_sync_finalize(allm.data_ptr(), sync.world, P, mean.data_ptr(), invstd.data_ptr(),
                                                   _ptr(running_mean), _ptr(running_var), _ptr(nbt), momentum, eps, C,
                                                   st), "norm_sync_finalize")
        else:
            if mask is not None:
                raise ValueError("norm: a fused dropout mask needs batch statistics")
            mean = _plain(running_mean)
            invstd = torch.empty_like(mean)  # eval mode: invstd = 1/sqrt(running_var + eps)
            check(lib.migan_rsqrt_eps(_plain(running_var).data_ptr(), invstd.data_ptr(), C, eps, st), "rsqrt_eps")
        rs = canon(res) if res is not None else None
        if shuffle:
            y = _empty_nhwc((N, C // 4, 2 * H, 2 * W), xs)
            ctx.shuffle = (H, W)
        else:
            y = torch.empty_like(xs)
        if pw is None and not shuffle:
            check(lib.migan_norm_apply(xs.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma),
                                       _ptr(beta), _ptr(rs), G, P, C, act, slope, st), "norm_apply")
        else:
            sh, sw = ctx.shuffle or (0, 0)   # (cross-replica statistics: mean / invstd above are the global batch's; the launch is the same)
            check(lib.migan_norm_apply_prelu(xs.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma),
                                             _ptr(beta), _ptr(rs), _ptr(pw), G, P, C, sh, sw, st), "norm_apply_prelu")
        ctx.cfg = (G, P, C, act, slope, use_batch_stats, gamma is not None, res is not None)
        ctx.sync = sync
        # x itself is saved next to its dense copy: a differentiable backward (gradient penalties) needs the input WITH its
        # autograd history (same storage when x already is dense NHWC)
        ctx.save_for_backward(xs, gamma, beta, mean, invstd, x)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, gamma, beta, mean, invstd, x_in = ctx.saved_tensors
        G, P, C, act, slope, batch_stats, affine, has_res = ctx.cfg
        if not batch_stats:
            raise NotImplementedError("backward through eval-mode BatchNorm is not on the reference path")
        if ctx.mask is not None:
            if torch.is_grad_enabled():
                raise NotImplementedError("double backward through a normalisation with a fused Dropout is not on the reference path")
            dy = canon(dy)
            dx = torch.empty_like(xs)
            slabs, nslab = None, 0
            if _COLSUM_FUSE and not ctx.x_nchw:
                nslab = lib.migan_norm_colsum_slabs(G, P, C)
                slabs = torch.empty(max(nslab * C, 1), device=xs.device, dtype=torch.float32)
            check(lib.migan_norm_bwd_small(xs.data_ptr(), dy.data_ptr(), ctx.mask.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                           _ptr(gamma), _ptr(beta), dx.data_ptr(), G, P, C, act, slope, _ptr(slabs), _stream()),
                  "norm_bwd_small")
            if ctx.x_nchw:
                dx = to_nchw(dx)
            elif slabs is not None:
                _attach_colsum(dx, slabs, nslab, C)
            return (dx,) + (None,) * 15
        if torch.is_grad_enabled():
            if ctx.bn_groups > 1:
                raise NotImplementedError("double backward through batch_groups() BatchNorm is not on the reference path")
            if ctx.prelu is not None or ctx.shuffle:
                raise NotImplementedError("double backward through a fused BatchNorm [+PixelShuffle] +PReLU is not on the reference path")
            return _norm_backward_differentiable(ctx, dy, xs, gamma, beta, mean, invstd, x_in)
        dy = canon(dy)
        dx = torch.empty_like(xs)
        dgamma = dbeta = None
        acc = 0
        if affine and (G == 1 or ctx.bn_groups > 1):
            sg, sb = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
            if sg is not None and sb is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
                dgamma, dbeta, acc = sg, sb, 1
            else:
                dgamma = torch.empty(C, device=xs.device, dtype=torch.float32)
                dbeta = torch.empty_like(dgamma)
        nb = lib.migan_norm_workspace(G, P, C)
        ws = _ws(nb, xs)
        # column-sum slabs of dx: the bias gradient of the conv in front of this layer is reduced from them inside that
        # conv's wgrad launch (4-D activations only; a Linear in front of BatchNorm1d has a per-feature bias)
        slabs, nslab = None, 0
        if xs.dim() == 4 and _COLSUM_FUSE and not ctx.x_nchw:
            nslab = lib.migan_norm_colsum_slabs(G, P, C)
            slabs = torch.empty(max(nslab * C, 1), device=xs.device, dtype=torch.float32)
        st = _stream()
        dprelu = None
        if ctx.prelu is not None or ctx.shuffle:
            pw = _plain(ctx.prelu)
            want_dp = pw is not None and ctx.needs_input_grad[13]
            pslot = _grad_slot(ctx.prelu) if want_dp else None
            sh, sw = ctx.shuffle or (0, 0)
            dpt = pslot if pslot is not None else (torch.empty(1, device=xs.device, dtype=torch.float32) if want_dp else None)
            nbp = lib.migan_norm_workspace_prelu(G, P, C)
            wsp = _ws(nbp, xs)
            if ctx.sync is None:
                check(lib.migan_norm_bwd_prelu(xs.data_ptr(), dy.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma),
                                               _ptr(beta), _ptr(pw), dx.data_ptr(), _ptr(dgamma), _ptr(dbeta), _ptr(dpt), G, P, C,
                                               wsp.data_ptr(), nbp, acc, 1 if pslot is not None else 0, _ptr(slabs), sh, sw, st),
                      "norm_bwd_prelu")
            else:
                # cross-replica BatchNorm (srgan.py:97-145 sharded 2 images per rank): the two batch sums cover the global batch -
                # all-reduce of 2*C floats between the halves; dgamma / dbeta / dprelu stay this rank's sums (the bucket exchange adds them)
                sums = torch.empty(2 * C, device=xs.device, dtype=torch.float32)
                check(lib.migan_norm_bwd_sums_prelu(xs.data_ptr(), dy.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma),
                                                    _ptr(beta), _ptr(pw), sums.data_ptr(), _ptr(dgamma), _ptr(dbeta), _ptr(dpt), G, P, C,
                                                    wsp.data_ptr(), nbp, acc, 1 if pslot is not None else 0, sh, sw, st),
                      "norm_bwd_sums_prelu")
                ctx.sync.all_reduce_sum(sums)
                check(lib.migan_norm_bwd_apply_prelu(xs.data_ptr(), dy.data_ptr(), dx.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                     _ptr(gamma), _ptr(beta), _ptr(pw), sums.data_ptr(), G, P, C, P * ctx.sync.world,
                                                     _ptr(slabs), sh, sw, st), "norm_bwd_apply_prelu")
            if want_dp and pslot is None:
                dprelu = dpt.view(ctx.prelu.shape)
        elif ctx.sync is None:
            check(lib.migan_norm_bwd(xs.data_ptr(), dy.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma),
                                     _ptr(beta), dx.data_ptr(), _ptr(dgamma), _ptr(dbeta), G, P, C, act, slope,
                                     ws.data_ptr(), nb, acc, _ptr(slabs), st), "norm_bwd")
        else:
            # cross-replica BatchNorm: the two batch sums cover the global batch (all-reduce of 2*C floats); dgamma/dbeta
            # stay local sums - they are summed over ranks with the rest of the gradient bucket
            sums = torch.empty(2 * C, device=xs.device, dtype=torch.float32)
            check(lib.migan_norm_bwd_sums(xs.data_ptr(), dy.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma),
                                          _ptr(beta), sums.data_ptr(), _ptr(dgamma), _ptr(dbeta), G, P, C, act, slope,
                                          ws.data_ptr(), nb, acc, st), "norm_bwd_sums")
            ctx.sync.all_reduce_sum(sums)
            check(lib.migan_norm_bwd_apply(xs.data_ptr(), dy.data_ptr(), dx.data_ptr(), mean.data_ptr(),
                                           invstd.data_ptr(), _ptr(gamma), _ptr(beta), sums.data_ptr(), G, P, C, act,
                                           slope, P * ctx.sync.world, _ptr(slabs), st), "norm_bwd_apply")
        if acc:
            dgamma = dbeta = None
        if ctx.x_nchw:
            dx = to_nchw(dx)
        elif slabs is not None:
            _attach_colsum(dx, slabs, nslab, C)
        return dx, dgamma, dbeta, (dy if has_res else None), None, None, None, None, None, None, None, None, None, dprelu, None, None


class _NormBwdFn(Function):
    """dx of the normalisation backward as a differentiable op (inputs: the norm input x, the gradient d w.r.t. the norm
    output, gamma); its own backward is migan_norm_bwd2 (formulas in csrc/norm.hip).  mean / invstd are the statistics of
    x from the forward pass: their dependence on x is part of those formulas."""

    @staticmethod
    def forward(ctx, x, d, gamma, mean, invstd, cfg):
        G, P, C = cfg
        xs, ds = canon(x), canon(d)
        ctx.x_nchw = x.dim() == 4 and x.is_contiguous() and not x.is_contiguous(memory_format=CL)
        gm = _plain(gamma)
        dx = torch.empty_like(xs)
        nb = lib.migan_norm_workspace(G, P, C)
        ws = _ws(nb, xs)
        check(lib.migan_norm_bwd(xs.data_ptr(), ds.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gm), None,
                                 dx.data_ptr(), None, None, G, P, C, ACT_NONE, 0.0, ws.data_ptr(), nb, 0, None, _stream()),
              "norm_bwd")
        ctx.cfg = cfg
        ctx.gamma_param = gamma
        ctx.save_for_backward(xs, ds, gm, mean, invstd)
        return to_nchw(dx) if ctx.x_nchw else dx

    @staticmethod
    def backward(ctx, u):
        _first_order_only("normalisation (third derivative)")
        xs, ds, gm, mean, invstd = ctx.saved_tensors
        G, P, C = ctx.cfg
        u = canon(u)
        if u.dim() == 4 and ctx.x_nchw:
            u = to_nhwc(u)
        gx = torch.empty_like(xs) if ctx.needs_input_grad[0] else None
        gd = torch.empty_like(xs) if ctx.needs_input_grad[1] else None
        gg, acc = None, 0
        if gm is not None and ctx.needs_input_grad[2] and G == 1:
            slot = _grad_slot(ctx.gamma_param)
            gg, acc = (slot, 1) if slot is not None else (torch.empty(C, device=xs.device, dtype=torch.float32), 0)
        nb = lib.migan_norm_workspace2(G, P, C)
        ws = _ws(nb, xs)
        check(lib.migan_norm_bwd2(xs.data_ptr(), ds.data_ptr(), u.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gm),
                                  _ptr(gd), _ptr(gx), _ptr(gg), acc, G, P, C, ws.data_ptr(), nb, _stream()), "norm_bwd2")
        if gx is not None and ctx.x_nchw:
            gx = to_nchw(gx)
        return gx, gd, (None if acc else gg), None, None, None


def _norm_backward_differentiable(ctx, dy, xs, gamma, beta, mean, invstd, x_in):
    """_Norm.backward recorded as differentiable ops (create_graph=True: dragan.py:156-163 through BatchNorm2d(C, 0.8),
    dualgan.py through InstanceNorm2d).  The fused LeakyReLU/ReLU is un-fused: its derivative comes from the recomputed
    layer output, a constant of the second differentiation (piecewise-linear activations only)."""
    G, P, C, act, slope, batch_stats, affine, has_res = ctx.cfg
    if ctx.sync is not None:
        raise NotImplementedError("double backward through cross-replica BatchNorm is not implemented")
    g = dy
    if act != ACT_NONE:
        if act not in (ACT_LRELU, ACT_RELU):
            raise NotImplementedError("norm: fused activation %d has no second derivative path" % act)
        with torch.no_grad():
            y_out = torch.empty_like(xs)
            check(lib.migan_norm_apply(xs.data_ptr(), y_out.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma),
                                       _ptr(beta), None, G, P, C, act, slope, _stream()), "norm_apply")
        g = _ActBwd.apply(g, y_out, act, slope)
    dx = _NormBwdFn.apply(x_in, g, ctx.params[0] if affine else None, mean, invstd, (G, P, C))
    dgamma = dbeta = None
    if affine and G == 1 and not _INPUT_GRAD_ONLY and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
        with torch.no_grad():  # first-order parameter gradients of THIS backward (not differentiable again)
            gd = canon(g.detach())
            dgamma = torch.empty(C, device=xs.device, dtype=torch.float32)
            dbeta = torch.empty_like(dgamma)
            sums = torch.empty(2 * C, device=xs.device, dtype=torch.float32)
            nb = lib.migan_norm_workspace(G, P, C)
            ws = _ws(nb, xs)
            check(lib.migan_norm_bwd_sums(xs.data_ptr(), gd.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma),
                                          _ptr(beta), sums.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), G, P, C, ACT_NONE,
                                          0.0, ws.data_ptr(), nb, 0, _stream()), "norm_bwd_sums")
    return dx, dgamma, dbeta, (dy if has_res else None), None, None, None, None, None, None, None, None, None, None, None, None


_BN_GROUPS = 1


@contextlib.contextmanager
def batch_groups(k):
    """Inside: a training-mode BatchNorm call on a batch of k * n samples behaves as k calls on its k consecutive
    sub-batches of n samples (per-sub-batch statistics; the k running-statistics updates applied in order).  Lets a step
    run D(real) and D(fake) (dcgan.py:176-177) as ONE pass over cat(real, fake) with the reference's numerics."""
    global _BN_GROUPS
    prev, _BN_GROUPS = _BN_GROUPS, int(k)
    try:
        yield
    finally:
        _BN_GROUPS = prev


def norm_small_takes(x, instance):
    """True when norm(x, instance=...) takes the one-launch small-tensor route - the only one that applies a dropout `mask`."""
    if not instance or x.dim() != 4 or (_SYNC_BN is not None and _SYNC_BN.world > 1):
        return False
    N, C, H, W = x.shape
    return bool(lib.migan_norm_small_ok(N, H * W, C)) and _stats_side(x, True, N, H * W, C) is None


def norm(x, gamma=None, beta=None, res=None, running_mean=None, running_var=None, use_batch_stats=True, momentum=0.1,
         eps=1e-5, instance=False, act=ACT_NONE, slope=0.0, num_batches_tracked=None, prelu=None, shuffle=0, mask=None):
    """`num_batches_tracked` (int64 scalar on the device) is incremented by the statistics kernel itself; `prelu`: weight of
    an nn.PReLU() (single slope) applied behind the normalisation inside the same launches; `shuffle` = 2: nn.PixelShuffle(2)
    as the store index map of the same launches (output (N, C/4, 2H, 2W))."""
    return _Norm.apply(x, gamma, beta, res, running_mean, running_var, bool(use_batch_stats), float(momentum),
                       float(eps), bool(instance), int(act), float(slope), num_batches_tracked, prelu, int(shuffle), mask)


_BN_FOLD = __import__("os").environ.get("MIGAN_BN_FOLD", "1") == "1"   # A/B knob (round 6): 0 = BatchNorm+PReLU as launches of their own


def bn_prelu_conv64_takes(x, w, stride, pads, dilation=(1, 1), groups=1):
    """True when bn_prelu_conv64(x, ...) serves the chain: local-batch BatchNorm statistics, first-order backward, the geometry of
    csrc/conv_c64.hip."""
    if not (_BN_FOLD and _C64) or x.dim() != 4 or w.dim() != 4 or not on_device(x):
        return False
    if (_SYNC_BN is not None and _SYNC_BN.world > 1) or _BN_GROUPS != 1 or tuple(dilation) != (1, 1) or groups != 1:
        return False
    N, C, H, W = x.shape
    Co, Ci, R, S = w.shape
    if Ci != C or not w.is_contiguous() or N * H * W <= 1:
        return False
    pt, pl, pb, pr = pads
    return lib.migan_c64_conv_ok(N, H, W, Ci, Co, R, S, int(stride), pt, pl, pb, pr, GATHER_ZERO) == 1


class _BnPreluConv64(Function):
    """conv2d(prelu(batch_norm(x)), w, b) for BatchNorm2d(64, 0.8) -> PReLU() -> Conv2d(64, 64, 3, 1, 1) (srgan/models.py:23-25) with the
    normalised, activated tensor never stored: one statistics pass over x, then the convolution reads x through the affine map + PReLU
    on its way into LDS (csrc/conv_c64.hip, INMAP), and so does its weight gradient.  Backward: input gradient of the conv -> the
    BatchNorm+PReLU backward launches of _Norm.  First order only (the reference never differentiates srgan's generator twice)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, momentum, eps, prelu, w, b):
        xs = canon(x)
        N, C, H, W = xs.shape
        P = N * H * W
        ctx.x_nchw = x.is_contiguous() and not x.is_contiguous(memory_format=CL)
        ctx.params = (gamma, beta, prelu, w, b)
        gm, bt, pw, wt, bs = _plain(gamma), _plain(beta), _plain(prelu), _plain(w), _plain(b)
        st = _stream()
        mean = torch.empty(C, device=xs.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        nb = lib.migan_norm_workspace(1, P, C)
        ws = _ws(nb, xs)
        check(lib.migan_norm_stats(xs.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                   momentum, eps, 1, P, C, ws.data_ptr(), nb, st), "norm_stats")
        wk = _packed_c64(w, wt, 0)
        y = _empty_nhwc((N, C, H, W), xs)
        check(lib.migan_c64_conv_fwd(xs.data_ptr(), wk.data_ptr(), _ptr(bs), y.data_ptr(), N, H, W, ACT_NONE, 0.0, 0, mean.data_ptr(),
                                     invstd.data_ptr(), _ptr(gm), _ptr(bt), ACT_LRELU, 0.0, pw.data_ptr(), st), "c64_conv_fwd")
        ctx.save_for_backward(xs, gm, bt, mean, invstd, pw, wt)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only("BatchNorm2d -> PReLU -> Conv2d(64, 64, 3, 1, 1) folded into one convolution")
        xs, gm, bt, mean, invstd, pw, wt = ctx.saved_tensors
        gamma, beta, prelu, w, b = ctx.params
        N, C, H, W = xs.shape
        P = N * H * W
        dy = to_nhwc(dy)
        st = _stream()
        dw = db = dgamma = dbeta = dprelu = dx = None
        # -- the conv's parameter gradients: x read through the same map as the forward; on the weight-gradient stream inside a step body,
        #    as _Conv2d's (the launch has one workgroup per CU at the trunk size: the input gradient that follows shares the CUs with it)
        side = _colsum_side(dy, C) if (b is not None and ctx.needs_input_grad[10]) else None
        fork = _Fork(xs.device, dy.numel(), ctx.needs_input_grad[9])
        with fork:
            stw = _stream()
            if ctx.needs_input_grad[9]:
                slot = _grad_slot(w)
                dw = torch.empty_like(wt) if slot is None else slot
                dbp, dba, sl, nsl = None, 0, None, 0
                if side is not None:
                    dbt, dba, db = _bias_out(b, C, xs)
                    dbp, sl, nsl = dbt.data_ptr(), side[0].data_ptr(), side[1]
                nbc = lib.migan_c64_wgrad_workspace(N, H, W)
                wsc = _ws(nbc, xs)
                check(lib.migan_c64_conv_wgrad(xs.data_ptr(), dy.data_ptr(), dw.data_ptr(), wsc.data_ptr(), nbc, N, H, W,
                                               0 if slot is None else 1, dbp, dba, sl, nsl, mean.data_ptr(), invstd.data_ptr(), _ptr(gm),
                                               _ptr(bt), ACT_LRELU, 0.0, pw.data_ptr(), stw), "c64_conv_wgrad")
                if slot is not None:
                    dw = None
            if b is not None and ctx.needs_input_grad[10] and (side is None or not ctx.needs_input_grad[9]):
                db = _colsum(dy, P, C, _grad_slot(b))
        fork.join((dw, db), (dy, xs, side[0] if side is not None else None, mean, invstd))
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2] or ctx.needs_input_grad[8]):
            return (None,) * 9 + (dw, db)
        # -- gradient at the conv's input (= behind the PReLU): the same convolution with reversed taps
        wk = _packed_c64(w, wt, 1)
        g = _empty_nhwc((N, C, H, W), xs)
        check(lib.migan_c64_conv_fwd(dy.data_ptr(), wk.data_ptr(), None, g.data_ptr(), N, H, W, ACT_NONE, 0.0, 0, None, None, None, None, 0,
                                     0.0, None, st), "c64_conv_dgrad")
        # -- BatchNorm + PReLU backward, as _Norm.backward's fused-PReLU branch
        dx = torch.empty_like(xs)
        acc = 0
        if gm is not None:
            sg, sb = _grad_slot(gamma), _grad_slot(beta)
            if sg is not None and sb is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
                dgamma, dbeta, acc = sg, sb, 1
            else:
                dgamma = torch.empty(C, device=xs.device, dtype=torch.float32)
                dbeta = torch.empty_like(dgamma)
        slabs, nslab = None, 0
        if _COLSUM_FUSE and not ctx.x_nchw:
            nslab = lib.migan_norm_colsum_slabs(1, P, C)
            slabs = torch.empty(max(nslab * C, 1), device=xs.device, dtype=torch.float32)
        want_dp = ctx.needs_input_grad[8]
        pslot = _grad_slot(prelu) if want_dp else None
        dpt = pslot if pslot is not None else (torch.empty(1, device=xs.device, dtype=torch.float32) if want_dp else None)
        nbp = lib.migan_norm_workspace_prelu(1, P, C)
        wsp = _ws(nbp, xs)
        check(lib.migan_norm_bwd_prelu(xs.data_ptr(), g.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gm), _ptr(bt), pw.data_ptr(),
                                       dx.data_ptr(), _ptr(dgamma), _ptr(dbeta), _ptr(dpt), 1, P, C, wsp.data_ptr(), nbp, acc,
                                       1 if pslot is not None else 0, _ptr(slabs), 0, 0, st), "norm_bwd_prelu")
        if want_dp and pslot is None:
            dprelu = dpt.view(prelu.shape)
        if acc:
            dgamma = dbeta = None
        if ctx.x_nchw:
            dx = to_nchw(dx)
        elif slabs is not None:
            _attach_colsum(dx, slabs, nslab, C)
        return dx, dgamma, dbeta, None, None, None, None, None, dprelu, dw, db


def bn_prelu_conv64(x, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, prelu, w, b):
    """Training-mode BatchNorm2d (local batch statistics; running statistics and num_batches_tracked updated by the statistics kernel)
    -> single-slope PReLU -> Conv2d(64, 64, 3, 1, 1), the activated tensor never stored (see _BnPreluConv64)."""
    return _BnPreluConv64.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, float(momentum), float(eps), prelu, w, b)


def bn_act_conv1_takes(x, w, stride, pads, dilation=(1, 1), groups=1):
    """True when bn_act_conv1(x, ...) serves the chain BatchNorm2d -> [LeakyReLU | ReLU] -> Conv2d(C, 1, 3, 1, 1): local-batch statistics,
    first-order backward, the geometry of migan_conv2d_fwd_normed / migan_bn_conv1_bwd."""
    if not _BN_FOLD or x.dim() != 4 or w.dim() != 4 or not on_device(x):
        return False
    if (_SYNC_BN is not None and _SYNC_BN.world > 1) or _BN_GROUPS != 1 or tuple(dilation) != (1, 1) or groups != 1:
        return False
    N, C, H, W = x.shape
    Co, Ci, R, S = w.shape
    if (Co, Ci, R, S, int(stride)) != (1, C, 3, 3, 1) or tuple(pads) != (1, 1, 1, 1) or not w.is_contiguous():
        return False
    return (lib.migan_conv2d_fwd_normed_ok(N, H, W, C, H, W, 1, 3, 3, 1, 1, 1) == 1 and lib.migan_bn_conv1_bwd_ok(N, H, W, C) == 1)


class _BnActConv1(Function):
    """act_out(conv2d(act_in(batch_norm(x)), w, b)) for the generator's last block (dcgan.py:60-62: BatchNorm2d(64, 0.8), LeakyReLU(0.2),
    Conv2d(64, 1, 3, 1, 1), Tanh) with neither the normalised tensor nor the conv's input gradient stored: one statistics pass over x, the
    thin-N conv reads x through the normalisation; backward = one walk over x for the conv's weight gradient AND the BatchNorm sums, one for
    dx - the C-channel input gradient of the conv is nine products with the one-channel dz, recomputed in both (csrc/norm.hip
    bn_conv1_bwd_*_kernel).  First order only."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, momentum, eps, act_in, slope_in, w, b, act_out, slope_out):
        xs = canon(x)
        N, C, H, W = xs.shape
        P = N * H * W
        ctx.x_nchw = x.is_contiguous() and not x.is_contiguous(memory_format=CL)
        ctx.params = (gamma, beta, w, b)
        ctx.cfg = (act_in, slope_in, act_out, slope_out)
        gm, bt, wt, bs = _plain(gamma), _plain(beta), _plain(w), _plain(b)
        st = _stream()
        mean = torch.empty(C, device=xs.device, dtype=torch.float32)
        invstd = torch.empty_like(mean)
        nb = lib.migan_norm_workspace(1, P, C)
        ws = _ws(nb, xs)
        check(lib.migan_norm_stats(xs.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                   momentum, eps, 1, P, C, ws.data_ptr(), nb, st), "norm_stats")
        wp = _packed_perm(w, wt, "ohwi", (0, 2, 3, 1))
        y = _empty_nhwc((N, 1, H, W), xs)
        check(lib.migan_conv2d_fwd_normed(xs.data_ptr(), wp.data_ptr(), _ptr(bs), y.data_ptr(), N, H, W, C, H, W, 1, 3, 3, 1, 1, 1, act_out,
                                          slope_out, mean.data_ptr(), invstd.data_ptr(), _ptr(gm), _ptr(bt), act_in, slope_in, st),
              "conv2d_fwd_normed")
        ctx.save_for_backward(xs, gm, bt, mean, invstd, wt, y if act_out != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only("BatchNorm2d -> activation -> Conv2d(C, 1, 3, 1, 1) folded into one convolution")
        xs, gm, bt, mean, invstd, wt, y = ctx.saved_tensors
        gamma, beta, w, b = ctx.params
        act_in, slope_in, act_out, slope_out = ctx.cfg
        N, C, H, W = xs.shape
        P = N * H * W
        dz = to_nhwc(dy)
        if act_out != ACT_NONE:
            dz = _act_bwd_raw(dz, y, act_out, slope_out)
        st = _stream()
        db, dbt, dba = None, None, 0
        if b is not None and ctx.needs_input_grad[11]:
            dbt, dba, db = _bias_out(b, 1, xs)   # the sum of dz comes out of the first walk
        wslot = _grad_slot(w) if ctx.needs_input_grad[10] else None
        dw = wslot if wslot is not None else torch.empty_like(wt)
        dgamma = dbeta = None
        acc = 0
        if gm is not None:
            sg, sb = _grad_slot(gamma), _grad_slot(beta)
            if sg is not None and sb is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
                dgamma, dbeta, acc = sg, sb, 1
            else:
                dgamma = torch.empty(C, device=xs.device, dtype=torch.float32)
                dbeta = torch.empty_like(dgamma)
        slabs, nslab = None, 0
        if _COLSUM_FUSE and not ctx.x_nchw:
            nslab = lib.migan_norm_colsum_slabs(1, P, C)
            slabs = torch.empty(max(nslab * C, 1), device=xs.device, dtype=torch.float32)
        wp = _packed_perm(w, wt, "ohwi", (0, 2, 3, 1))
        dx = torch.empty_like(xs)
        nb = lib.migan_bn_conv1_bwd_workspace(N, H, W, C)
        ws = _ws(nb, xs)
        check(lib.migan_bn_conv1_bwd(xs.data_ptr(), dz.data_ptr(), wp.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gm), _ptr(bt),
                                     act_in, slope_in, dx.data_ptr(), dw.data_ptr(), 1 if wslot is not None else 0, _ptr(dbt), dba, _ptr(dgamma),
                                     _ptr(dbeta), acc, _ptr(slabs), ws.data_ptr(), nb, N, H, W, C, st), "bn_conv1_bwd")
        if wslot is not None or not ctx.needs_input_grad[10]:
            dw = None
        if acc:
            dgamma = dbeta = None
        if ctx.x_nchw:
            dx = to_nchw(dx)
        elif slabs is not None:
            _attach_colsum(dx, slabs, nslab, C)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, dw, db, None, None


def bn_act_conv1(x, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, act_in, slope_in, w, b, act_out=ACT_NONE,
                 slope_out=0.0):
    """Training-mode BatchNorm2d -> act_in -> Conv2d(C, 1, 3, 1, 1) -> act_out as one Function (see _BnActConv1)."""
    return _BnActConv1.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, float(momentum), float(eps), int(act_in),
                             float(slope_in), w, b, int(act_out), float(slope_out))


# ---------------------------------------------------------------------------------------------- index remaps
class _Gather2d(Function):
    """Standalone ReflectionPad2d / ZeroPad2d / Upsample(scale_factor=2)."""

    @staticmethod
    def forward(ctx, x, pads, mode):
        xs = to_nhwc(x)
        N, C, H, W = xs.shape
        pt, pl, pb, pr = pads
        HL, WL = (2 * H, 2 * W) if mode == GATHER_UP2 else (H, W)
        if mode == GATHER_REFLECT and (max(pt, pb) >= H or max(pl, pr) >= W):
            raise ValueError("reflection padding must be smaller than the input")
        Ho, Wo = HL + pt + pb, WL + pl + pr
        y = _empty_nhwc((N, C, Ho, Wo), xs)
        check(lib.migan_gather2d_fwd(xs.data_ptr(), y.data_ptr(), N, H, W, C, Ho, Wo, pt, pl, mode, _stream()), "gather2d")
        ctx.geom = (N, H, W, C, Ho, Wo, pt, pl, mode)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only('pad/upsample')
        N, H, W, C, Ho, Wo, pt, pl, mode = ctx.geom
        dy = to_nhwc(dy)
        dx = _empty_nhwc((N, C, H, W), dy)
        check(lib.migan_gather2d_bwd(dy.data_ptr(), dx.data_ptr(), N, H, W, C, Ho, Wo, pt, pl, mode, _stream()),
              "gather2d_bwd")
        return dx, None, None


def gather2d(x, pads, mode):
    return _Gather2d.apply(x, tuple(int(p) for p in pads), int(mode))


class _PixelShuffle(Function):
    @staticmethod
    def forward(ctx, x, r):
        xs = to_nhwc(x)
        N, Crr, H, W = xs.shape
        if Crr % (r * r):
            raise ValueError("pixel_shuffle: channels not divisible by r^2")
        C = Crr // (r * r)
        y = _empty_nhwc((N, C, H * r, W * r), xs)
        check(lib.migan_pixel_shuffle(xs.data_ptr(), y.data_ptr(), N, H, W, C, r, 1, _stream()), "pixel_shuffle")
        ctx.geom = (N, H, W, C, r)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only('PixelShuffle')
        N, H, W, C, r = ctx.geom
        dy = to_nhwc(dy)
        dx = _empty_nhwc((N, C * r * r, H, W), dy)
        check(lib.migan_pixel_shuffle(dy.data_ptr(), dx.data_ptr(), N, H, W, C, r, 0, _stream()), "pixel_unshuffle")
        return dx, None


def pixel_shuffle(x, r):
    return _PixelShuffle.apply(x, int(r))


class _MaxPool2(Function):
    @staticmethod
    def forward(ctx, x, relu_in=False):
        ctx.relu_in = bool(relu_in)   # x is the output of a conv+ReLU that handed its ReLU backward to this pool (see _Conv2d.forward)
        xs = to_nhwc(x)
        N, C, H, W = xs.shape
        if H % 2 or W % 2:
            raise ValueError("maxpool2: odd spatial size is not on the reference path")
        y = _empty_nhwc((N, C, H // 2, W // 2), xs)
        check(lib.migan_maxpool2_fwd(xs.data_ptr(), y.data_ptr(), N, H, W, C, _stream()), "maxpool2")
        ctx.save_for_backward(xs)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only('MaxPool2d')
        (xs,) = ctx.saved_tensors
        N, C, H, W = xs.shape
        dy = to_nhwc(dy)
        dx = torch.empty_like(xs)
        fn = lib.migan_maxpool2_relu_bwd if ctx.relu_in else lib.migan_maxpool2_bwd
        check(fn(xs.data_ptr(), dy.data_ptr(), dx.data_ptr(), N, H, W, C, _stream()), "maxpool2_bwd")
        return dx, None


def maxpool2(x, relu_in=False):
    return _MaxPool2.apply(x, bool(relu_in))


class _CatC(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = to_nhwc(a), to_nhwc(b)
        N, Ca, H, W = a.shape
        Cb = b.shape[1]
        if b.shape[0] != N or b.shape[2:] != a.shape[2:]:
            raise ValueError("cat: shapes differ outside dim 1")
        y = _empty_nhwc((N, Ca + Cb, H, W), a)
        check(lib.migan_cat_channels(a.data_ptr(), b.data_ptr(), y.data_ptr(), N * H * W, Ca, Cb, 1, _stream()), "cat")
        ctx.geom = (N, Ca, Cb, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only('cat')
        N, Ca, Cb, H, W = ctx.geom
        dy = to_nhwc(dy)
        da = _empty_nhwc((N, Ca, H, W), dy)
        db = _empty_nhwc((N, Cb, H, W), dy)
        check(lib.migan_cat_channels(da.data_ptr(), db.data_ptr(), dy.data_ptr(), N * H * W, Ca, Cb, 0, _stream()),
              "split")
        return da, db


def cat_channels(a, b):
    return _CatC.apply(a, b)


class _Axpby(Function):
    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        a = canon(a)
        b = canon(b) if b is not None else None
        if b is not None and b.shape != a.shape:
            raise ValueError("axpby: shapes differ (no broadcasting on this path)")
        y = torch.empty_like(a)
        check(lib.migan_axpby(a.data_ptr(), alpha, _ptr(b), beta, y.data_ptr(), a.numel(), _stream()), "axpby")
        ctx.ab = (alpha, beta, b is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        alpha, beta, has_b = ctx.ab
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = g if alpha == 1.0 else _Axpby.apply(g, None, alpha, 0.0)
        if has_b and ctx.needs_input_grad[1]:
            gb = g if beta == 1.0 else _Axpby.apply(g, None, beta, 0.0)
        return ga, gb, None, None


def copy_into(dst, src):
    """dst[...] = src for a dense destination (e.g. a batch slice of an NHWC buffer) without autograd history: one HIP launch."""
    src = canon(src.detach())
    if dst.numel() != src.numel() or dst.requires_grad:
        raise ValueError("copy_into: size mismatch or destination requires grad")
    if dst.dim() == 4 and not (dst.is_contiguous(memory_format=CL) or dst.is_contiguous()):
        raise ValueError("copy_into: destination must be dense")
    if dst.dim() == 4 and dst.shape[1] != 1 and dst.is_contiguous(memory_format=CL) != src.is_contiguous(memory_format=CL):
        raise ValueError("copy_into: source and destination layouts differ")
    check(lib.migan_axpby(src.data_ptr(), 1.0, None, 0.0, dst.data_ptr(), src.numel(), _stream()), "copy_into")
    return dst


def axpby(a, b=None, alpha=1.0, beta=1.0):
    return _Axpby.apply(a, b, float(alpha), float(beta))


def add(a, b):
    return _Axpby.apply(a, b, 1.0, 1.0)


class _Fork2(Function):
    """x -> (x, x) for a tensor with TWO consumers (a residual block's input: `x + self.block(x)`, cyclegan/models.py:37,
    srgan/models.py:30,68; a U-Net skip, pix2pix/models.py:50).  Forward is free (two views); backward adds the two incoming gradients with
    the library's own kernel.  Without it autograd accumulates them itself - an ATen `add` launch per such tensor and step (54 per CycleGAN
    step) on a path that otherwise runs hand-written kernels only."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        if ga.dim() == 4:
            ga, gb = to_nhwc(ga), to_nhwc(gb)
        return add(ga, gb)


_FORK2 = __import__("os").environ.get("MIGAN_FORK2", "1") == "1"   # A/B knob: 0 = autograd's own accumulation (an ATen add)


def fork2(x):
    if not _FORK2:
        return x, x
    a, b = _Fork2.apply(x)
    return a, b


def zero_(t):
    """t.zero_() through the C ABI (hipMemsetAsync on the current stream): the optimisers' gradient buckets (torch's fill is an ATen kernel)."""
    if t.numel():
        if not _ZERO_ABI:
            return t.zero_()
        check(lib.migan_zero(t.data_ptr(), t.numel() * t.element_size(), _stream()), "zero")
    return t


_ZERO_ABI = __import__("os").environ.get("MIGAN_ZERO", "1") == "1"   # A/B knob: 0 = torch's fill kernel


class _SubBatchMean(Function):
    """a - b.mean(0, keepdim=True): the relativistic average logits of esrgan.py:137,165-166 in one launch."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = canon(a), canon(b)
        if a.shape != b.shape or a.stride() != b.stride():
            raise ValueError("sub_batch_mean: operands must have the same shape and layout")
        y = torch.empty_like(a)
        N = a.shape[0]
        check(lib.migan_batch_mean_axpy(a.data_ptr(), b.data_ptr(), y.data_ptr(), N, a.numel() // N, 1.0, -1.0, _stream()),
              "batch_mean_axpy")
        return y

    @staticmethod
    def backward(ctx, g):
        _first_order_only("sub_batch_mean")
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = g
        if ctx.needs_input_grad[1]:
            gc = canon(g)
            gb = torch.empty_like(gc)
            N = gc.shape[0]
            check(lib.migan_batch_mean_axpy(None, gc.data_ptr(), gb.data_ptr(), N, gc.numel() // N, 0.0, -1.0, _stream()),
                  "batch_mean_axpy")
        return ga, gb


def sub_batch_mean(a, b):
    return _SubBatchMean.apply(a, b)


# ---------------------------------------------------------------------------------------------- dropout
def rand_mask(shape, p, seed, counter, device):
    """Bernoulli(1-p)/(1-p) mask from the device Philox stream (counter: device uint64 tensor or None)."""
    n = 1
    for s in shape:
        n *= int(s)
    mask = torch.empty(shape, device=device, dtype=torch.float32)
    check(lib.migan_rand_mask(mask.data_ptr(), n, float(p), int(seed), _ptr(counter), _stream()), "rand_mask")
    return mask


class _MulMask(Function):
    """y = x * mask; mask is [N,C] (Dropout2d: one draw per plane) or full-size (Dropout)."""

    @staticmethod
    def forward(ctx, x, mask):
        xs = canon(x)
        mask = _plain(mask)
        nhwc = mask.dim() == 4 and xs.dim() == 4 and mask.shape == xs.shape and mask.is_contiguous(memory_format=CL)
        if not nhwc:   # (a mask that already has the activations' NHWC layout is used in place: no copy, no transpose)
            mask = mask.contiguous()
        y = torch.empty_like(xs)
        ctx.per_plane = xs.dim() == 4 and mask.dim() == 2
        if ctx.per_plane:
            N, C, H, W = xs.shape
            if tuple(mask.shape) != (N, C):
                raise ValueError("dropout2d mask must be [N, C]")
            check(lib.migan_mul_nc(xs.data_ptr(), mask.data_ptr(), y.data_ptr(), N, H * W, C, _stream()), "mul_nc")
        else:
            if mask.numel() != xs.numel():
                raise ValueError("dropout mask must match the input")
            if xs.dim() == 4 and not nhwc:
                mask = to_nhwc(mask.view(xs.shape)) if mask.shape == xs.shape else mask
            check(lib.migan_mul(xs.data_ptr(), mask.data_ptr(), y.data_ptr(), xs.numel(), _stream()), "mul")
        ctx.save_for_backward(mask)
        ctx.shape4 = tuple(xs.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        if torch.is_grad_enabled():  # create_graph: the mask multiply is linear, its backward is the same op
            return _MulMask.apply(g, mask), None
        g = canon(g)
        dx = torch.empty_like(g)
        if ctx.per_plane:
            N, C, H, W = ctx.shape4
            check(lib.migan_mul_nc(g.data_ptr(), mask.data_ptr(), dx.data_ptr(), N, H * W, C, _stream()), "mul_nc")
        else:
            check(lib.migan_mul(g.data_ptr(), mask.data_ptr(), dx.data_ptr(), g.numel(), _stream()), "mul")
        return dx, None


def mul_mask(x, mask):
    return _MulMask.apply(x, mask)


# ---------------------------------------------------------------------------------------------- losses
class _Loss(Function):
    """mean-reduced BCELoss / MSELoss / L1Loss / plain mean; target tensor or constant."""

    @staticmethod
    def forward(ctx, x, target, kind, tconst):
        xs = canon(x)
        t = None
        if target is not None:
            t = canon(target)
            if t.shape != xs.shape:
                raise ValueError("loss: target shape %s != input shape %s" % (tuple(t.shape), tuple(xs.shape)))
        out = torch.empty((), device=xs.device, dtype=torch.float32)
        nb = lib.migan_reduce_workspace()
        ws = _ws(nb, xs)
        check(lib.migan_loss_fwd(kind, xs.data_ptr(), _ptr(t), tconst, out.data_ptr(), xs.numel(), ws.data_ptr(), nb,
                                 _stream()), "loss_fwd")
        ctx.kind, ctx.tconst = kind, tconst
        ctx.save_for_backward(xs, t)
        return out

    @staticmethod
    def backward(ctx, g):
        _first_order_only('loss')
        xs, t = ctx.saved_tensors
        g = _plain(g).contiguous()
        dx = torch.empty_like(xs)
        check(lib.migan_loss_bwd(ctx.kind, xs.data_ptr(), _ptr(t), ctx.tconst, g.data_ptr(), dx.data_ptr(), xs.numel(),
                                 _stream()), "loss_bwd")
        return dx, None, None, None


class _Pullaway(Function):
    """ebgan.py:142-148 pullaway_loss(embeddings) in one launch (and one for its gradient)."""

    @staticmethod
    def forward(ctx, e):
        es = _plain(e).contiguous()
        if es.dim() != 2 or es.shape[0] < 2:
            raise ValueError("pullaway_loss: expected (B >= 2, D) embeddings")
        B, D = es.shape
        out = torch.empty((), device=es.device, dtype=torch.float32)
        ws = torch.empty(B + D, device=es.device, dtype=torch.float32)
        check(lib.migan_pullaway_fwd(es.data_ptr(), out.data_ptr(), ws.data_ptr(), B, D, _stream()), "pullaway_fwd")
        ctx.save_for_backward(es, ws)
        return out

    @staticmethod
    def backward(ctx, g):
        _first_order_only('pullaway_loss')
        es, ws = ctx.saved_tensors
        g = _plain(g).contiguous()
        de = torch.empty_like(es)
        check(lib.migan_pullaway_bwd(es.data_ptr(), ws.data_ptr(), g.data_ptr(), de.data_ptr(), es.shape[0], es.shape[1],
                                     _stream()), "pullaway_bwd")
        return de


def pullaway_loss(embeddings):
    return _Pullaway.apply(embeddings)


def loss(kind, x, target=None, tconst=0.0):
    return _Loss.apply(x, target, int(kind), float(tconst))


def mean(x):
    return _Loss.apply(x, None, LOSS_MEAN, 0.0)


# ---------------------------------------------------------------------------------------------- DCGAN-block clones (F2)
class _Embedding(Function):
    """nn.Embedding lookup (acgan.py:50): y[i] = weight[idx[i]]."""

    @staticmethod
    def forward(ctx, idx, weight):
        w = _plain(weight)
        _check_dev(w)
        if idx.dtype != torch.int64 or not on_device(idx):
            raise TypeError("embedding: indices must be an int64 tensor on the GPU (the reference passes LongTensor labels)")
        flat = idx.reshape(-1).contiguous()
        V, D = w.shape
        y = torch.empty(flat.numel(), D, device=w.device, dtype=torch.float32)
        check(lib.migan_embedding_fwd(w.data_ptr(), flat.data_ptr(), y.data_ptr(), flat.numel(), D, V, _stream()),
              "embedding_fwd")
        ctx.save_for_backward(flat)
        ctx.wshape, ctx.ishape, ctx.param = (V, D), tuple(idx.shape), weight
        return y.view(*idx.shape, D)

    @staticmethod
    def backward(ctx, dy):
        _first_order_only('Embedding')
        (flat,) = ctx.saved_tensors
        V, D = ctx.wshape
        dy = canon(dy.reshape(-1, D))
        slot = _grad_slot(ctx.param)
        dw = torch.empty(V, D, device=dy.device, dtype=torch.float32) if slot is None else slot
        check(lib.migan_embedding_bwd(dy.data_ptr(), flat.data_ptr(), dw.data_ptr(), flat.numel(), D, V,
                                      0 if slot is None else 1, _stream()), "embedding_bwd")
        return None, (dw if slot is None else None)


def embedding(idx, weight):
    return _Embedding.apply(idx, weight)


class _Softmax(Function):
    """Softmax over the class dimension of a (B, C) tensor (nn.Softmax() of acgan.py:100)."""

    @staticmethod
    def forward(ctx, x):
        xs = canon(x)
        if xs.dim() != 2:
            raise ValueError("softmax: expected (B, classes) input on this path")
        y = torch.empty_like(xs)
        check(lib.migan_softmax_fwd(xs.data_ptr(), y.data_ptr(), xs.shape[0], xs.shape[1], _stream()), "softmax_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        _first_order_only('Softmax')
        (y,) = ctx.saved_tensors
        dy = canon(dy)
        dx = torch.empty_like(y)
        check(lib.migan_softmax_bwd(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.shape[0], y.shape[1], _stream()),
              "softmax_bwd")
        return dx


def softmax(x):
    return _Softmax.apply(x)


class _CrossEntropy(Function):
    """nn.CrossEntropyLoss() (mean) on (B, C) scores with int64 class targets (acgan.py:113,175-176)."""

    @staticmethod
    def forward(ctx, x, target):
        xs = canon(x)
        if xs.dim() != 2 or target.dim() != 1 or target.shape[0] != xs.shape[0]:
            raise ValueError("cross_entropy: expected (B, C) scores and (B,) class indices")
        if target.dtype != torch.int64 or not on_device(target):
            raise TypeError("cross_entropy: targets must be an int64 tensor on the GPU")
        t = target.contiguous()
        B, C = xs.shape
        out = torch.empty((), device=xs.device, dtype=torch.float32)
        ws = torch.empty(2 * B, device=xs.device, dtype=torch.float32)
        check(lib.migan_cross_entropy_fwd(xs.data_ptr(), t.data_ptr(), out.data_ptr(), ws.data_ptr(), B, C, _stream()),
              "cross_entropy_fwd")
        ctx.save_for_backward(xs, t, ws)
        return out

    @staticmethod
    def backward(ctx, g):
        _first_order_only('CrossEntropyLoss')
        xs, t, ws = ctx.saved_tensors
        B, C = xs.shape
        g = _plain(g).contiguous()
        dx = torch.empty_like(xs)
        check(lib.migan_cross_entropy_bwd(xs.data_ptr(), t.data_ptr(), ws.data_ptr() + 4 * B, g.data_ptr(), dx.data_ptr(),
                                          B, C, _stream()), "cross_entropy_bwd")
        return dx, None


def cross_entropy(x, target):
    return _CrossEntropy.apply(x, target)


class _Mul(Function):
    """Elementwise product of two same-shape tensors (`torch.mul(self.label_emb(labels), noise)`, acgan.py:61)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = canon(a), canon(b)
        if a.shape != b.shape:
            raise ValueError("mul: shapes differ (no broadcasting on this path)")
        y = torch.empty_like(a)
        check(lib.migan_mul(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream()), "mul")
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, g):
        _first_order_only('mul')
        a, b = ctx.saved_tensors
        g = canon(g)
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = torch.empty_like(a)
            check(lib.migan_mul(g.data_ptr(), b.data_ptr(), ga.data_ptr(), a.numel(), _stream()), "mul")
        if ctx.needs_input_grad[1]:
            gb = torch.empty_like(b)
            check(lib.migan_mul(g.data_ptr(), a.data_ptr(), gb.data_ptr(), a.numel(), _stream()), "mul")
        return ga, gb


def mul(a, b):
    return _Mul.apply(a, b)


class _RowNorm(Function):
    @staticmethod
    def forward(ctx, x):
        xs = canon(x)
        B, D = xs.shape
        out = torch.empty(B, device=xs.device, dtype=torch.float32)
        check(lib.migan_rownorm_fwd(xs.data_ptr(), out.data_ptr(), B, D, _stream()), "rownorm")
        ctx.save_for_backward(xs, out)
        return out

    @staticmethod
    def backward(ctx, dn):
        _first_order_only('row norm')
        xs, nrm = ctx.saved_tensors
        dn = _plain(dn).contiguous()
        B, D = xs.shape
        dx = torch.empty_like(xs)
        check(lib.migan_rownorm_bwd(xs.data_ptr(), nrm.data_ptr(), dn.data_ptr(), dx.data_ptr(), B, D, _stream()),
              "rownorm_bwd")
        return dx


def rownorm(x):
    return _RowNorm.apply(x)


def dragan_interpolate(x, alpha, noise):
    """alpha*X + (1-alpha)*(X + 0.5*X.std()*noise) (dragan.py:149; no autograd: the result becomes the leaf the penalty
    differentiates with respect to).  X.std() = unbiased std over all elements, computed on the device."""
    xs, a, nz = canon(x), canon(alpha), canon(noise)
    if a.shape != xs.shape or nz.shape != xs.shape:
        raise ValueError("dragan_interpolate: alpha and noise must have the shape of X")
    n = xs.numel()
    mom = torch.empty(2, device=xs.device, dtype=torch.float32)
    nb = lib.migan_norm_workspace(1, n, 1)
    ws = _ws(nb, xs)
    st = _stream()
    check(lib.migan_norm_moments(xs.data_ptr(), mom.data_ptr(), mom.data_ptr() + 4, 1, n, 1, ws.data_ptr(), nb, st),
          "norm_moments")
    out = torch.empty_like(xs)
    check(lib.migan_dragan_interp(xs.data_ptr(), a.data_ptr(), nz.data_ptr(), mom.data_ptr() + 4, out.data_ptr(), n, st),
          "dragan_interp")
    return out


def rowscale(x, s):
    """y[b,:] = x[b,:]*s[b] (no autograd; used for the WGAN-GP interpolation of detached samples)."""
    xs = canon(x)
    B = xs.shape[0]
    D = xs.numel() // B
    y = torch.empty_like(xs)
    s = _plain(s).reshape(B).contiguous()
    check(lib.migan_rowscale(xs.data_ptr(), s.data_ptr(), y.data_ptr(), B, D, _stream()), "rowscale")
    return y


class _Relayout(Function):
    """Differentiable NCHW<->NHWC re-layout (gradient passes through; layouts are free in autograd)."""

    @staticmethod
    def forward(ctx, x, to_cl):
        return to_nhwc(x) if to_cl else to_nchw(x)

    @staticmethod
    def backward(ctx, g):
        return g, None


def relayout(x, channels_last):
    return _Relayout.apply(x, bool(channels_last))
