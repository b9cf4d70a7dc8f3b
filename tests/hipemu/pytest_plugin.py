"""TEST INFRASTRUCTURE: run the `-m gpu` parity tests on the host execution model of the kernels.

    python -m pytest tests/test_ops_gpu.py -m gpu -p hipemu.pytest_plugin [-k ...] [-x]

Every collected module's DEV becomes "cpu", util.gpu_copy keeps models on the CPU, torch.cuda.is_available() answers True for
the tests' own guards, and the whole session runs inside hipemu.host.emulated_device().  Tests that need real device features
(hipGraph capture, RCCL, multi-process launches, full-size batches) are expected to fail or take very long here - select with
-k.  An exploration tool (it found the K-tail over-read); the curated subset that runs in the CPU suite is
tests/test_kernels_emu_cpu.py."""
import contextlib

import torch

_STACK = contextlib.ExitStack()


def pytest_sessionstart(session):
    import hipemu.host

    _STACK.enter_context(hipemu.host.emulated_device())
    _STACK.callback(setattr, torch.cuda, "is_available", torch.cuda.is_available)
    torch.cuda.is_available = lambda: True


def pytest_collection_modifyitems(session, config, items):
    import util

    cpu_copy = lambda m, device="cpu": util.gpu_copy.__wrapped__(m, "cpu") if hasattr(util.gpu_copy, "__wrapped__") else _copy(m)  # noqa: E731

    def _copy(m):
        import copy

        import pytorch_gan_amd as pg

        c = copy.deepcopy(m)
        pg.swap(c)
        return c

    for mod in {it.module for it in items}:
        if hasattr(mod, "DEV"):
            mod.DEV = "cpu"
        if hasattr(mod, "gpu_copy"):
            mod.gpu_copy = lambda m, device="cpu": _copy(m)


def pytest_sessionfinish(session, exitstatus):
    _STACK.close()
