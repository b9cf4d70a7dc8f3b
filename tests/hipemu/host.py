"""TEST INFRASTRUCTURE: run the product's Python host mirror (pytorch_gan_amd.functional / nn / optim / steps) on torch CPU
tensors with every `migan_*` call executed by the host-side execution model of the HIP kernels (tests/hipemu).

    with hipemu.host.emulated_device():
        y = pg.functional.conv2d(x_cpu, w_cpu, ...)      # autograd Functions, modules, whole training steps

Everything is patched from HERE, inside the test process, and restored on exit: the library handle the host modules call, the
one device predicate they ask (`functional.on_device`), and the handful of `torch.cuda` stream / event entry points they use
(streams are a no-op in the model: every launch completes before the call returns).  The product contains no emulation
switch and keeps failing loudly on CPU tensors outside this context."""
import contextlib
import sys

import torch


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self, stream=None):
        pass

    def wait(self, stream=None):
        pass

    def synchronize(self):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return 1.0   # ms: time is not modelled; a positive constant keeps rate computations of callers finite


class _Stream:
    cuda_stream = 0
    device = torch.device("cpu")

    def __init__(self, *a, **k):
        pass

    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass

    def record_event(self, ev=None):
        return ev or _Event()

    def synchronize(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_MAIN = _Stream()


@contextlib.contextmanager
def emulated_device():
    import hipemu
    import pytorch_gan_amd as pg
    from pytorch_gan_amd import _lib, functional

    emu = hipemu.load()
    saved = []

    def patch(obj, name, value):
        saved.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    real = _lib.lib
    for mod in list(sys.modules.values()):
        if getattr(mod, "__name__", "").startswith("pytorch_gan_amd") and getattr(mod, "lib", None) is real:
            patch(mod, "lib", emu)
    patch(functional, "on_device", lambda t: True)
    patch(torch.cuda, "current_stream", lambda device=None: _MAIN)
    patch(torch.cuda, "Stream", _Stream)
    patch(torch.cuda, "Event", _Event)
    patch(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    patch(torch.cuda, "is_current_stream_capturing", lambda: False)
    patch(torch.cuda, "synchronize", lambda device=None: None)
    patch(torch.nn.Module, "cuda", lambda self, device=None: self)   # `.cuda()` of callers (bench.py builders): stay where we are
    patch(torch.Tensor, "cuda", lambda self, *a, **k: self)
    try:
        yield emu
    finally:
        for obj, name, value in reversed(saved):
            setattr(obj, name, value)
        # per-stream workspaces and plans created on CPU tensors must not leak into a later real-device use
        getattr(functional, "_SK_WS", {}).clear()
        del pg
