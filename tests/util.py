"""Shared helpers for the parity tests.  Stated fp32 tolerances:
   per-op forward / dgrad  rel_fro <= 2e-6 (3e-6 allowed: long reductions, K up to 5184, measured 2.3e-6),
   wgrad / statistics over >=1e5 terms  rel_fro <= 1e-5, whole-model forward rel_fro <= 1e-5,
   whole-model gradients rel_fro <= 5e-3 (two fp32 evaluations of a LeakyReLU/ReLU/MaxPool network may
   take different branches for elements within rounding of 0; ONE flipped element of an N-element activation
   moves the gradient by ~(slope gap)/sqrt(N) ~ 1e-3 at the test sizes — measured: torch CPU fp32 with 8 vs
   128 threads differ by 2.8e-3 on the DCGAN generator), losses |d| <= 1e-4*max(1,|loss|)."""
import copy

import numpy as np
import torch

# measured on MI355X against torch CPU fp32 (gpurun_out/r2a/test_errors.log, 456 comparisons): conv/linear forward <= 1.9e-6,
# dgrad <= 2.3e-6 (K = 5184 terms; the CPU fp32 result is itself ~1e-6 from fp64), weight gradients <= 2.4e-6, bias
# gradients (column sums over up to 4e5 pixels) <= 1.0e-5, whole-model forward <= 3.4e-6
TOL_FWD = 3e-6
TOL_WGRAD = 1e-5
TOL_BIAS = 2e-5
TOL_MODEL_FWD = 1e-5
TOL_MODEL_GRAD = 5e-3  # see test_models_gpu._noise_aware: activation sign-decision flips


def rel_fro(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    den = b.norm().item()
    return (a - b).norm().item() / (den if den > 0 else 1.0)


def assert_close(a, b, tol, what=""):
    r = rel_fro(a, b)
    log = __import__("os").environ.get("MIGAN_TEST_ERRLOG")  # measured errors, for calibrating the stated tolerances
    if log:
        with open(log, "a") as fh:
            fh.write("%-28s tol %.1e  rel_fro %.3e  shape %s\n" % (what, tol, r, tuple(a.shape)))
    assert r <= tol, "%s: rel_fro %.3e > %.1e" % (what, r, tol)
    return r


def gpu_copy(model, device=None):
    import os

    import pytorch_gan_amd as pg

    device = device or os.environ.get("MIGAN_TEST_DEVICE", "cuda:0")   # "cpu": worker processes of the execution-model tests

    m = copy.deepcopy(model)
    pg.swap(m)
    return m.to(device)


class Launches:
    """Launch counters of the C ABI (migan_debug_launch_count): `with Launches() as n: ...; n("kernel_name")` = launches of the
    kernels whose launch expression contains that text since the block was entered.  Dispatch is by geometry only, so tests
    that are ABOUT a specialised kernel assert that it - and not the general kernel next to it - served their shape."""

    def __init__(self, lib=None):
        if lib is None:
            from pytorch_gan_amd import functional

            lib = functional.lib   # the library handle in use (the execution-model tests patch it)
        self.lib = lib

    def __enter__(self):
        self.lib.migan_debug_launch_reset()
        return self

    def __exit__(self, *exc):
        return False

    def __call__(self, sym):
        return int(self.lib.migan_debug_launch_count(sym.encode()))


def load_golden(golden_dir, name):
    import os

    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def digest(t):
    t = t.detach().double().cpu().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def run_ranks(cmd, cwd, env, timeout):
    """Run a multi-process launch (torch.distributed.run or bench.py --gpus N) in its OWN process group and return
    (returncode | None on timeout, stdout, stderr).  On timeout the whole group is killed: ranks orphaned by killing only the
    launcher would keep the GPU busy under every later test (and under the smoke / bench runs that follow the suite)."""
    import os
    import signal
    import subprocess

    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
        return p.returncode, out, err
    except subprocess.TimeoutExpired:
        victims = []
        try:  # every descendant, whatever session the launcher put its workers in
            import psutil

            victims = psutil.Process(p.pid).children(recursive=True)
        except Exception:  # noqa: BLE001
            pass
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        for v in victims:
            try:
                v.kill()
            except Exception:  # noqa: BLE001
                pass
        out, err = p.communicate()
        return None, out, err


_SUITE_T0 = __import__("time").time()


def suite_budget(need_s, what):
    """The driver gives the whole `-m gpu` suite 1200 s on a box whose HOST cores may be several times slower than the last one
    (the oracle steps are CPU work).  A heavy test asks here first: if the time already spent plus its own typical need would
    pass MIGAN_SUITE_BUDGET_S (default 1000 s), it is skipped WITH this reason instead of taking every later test down with
    the limit.  Light tests never ask."""
    import os
    import time

    import pytest

    limit = float(os.environ.get("MIGAN_SUITE_BUDGET_S", "1000"))
    spent = time.time() - _SUITE_T0
    if spent + need_s > limit:
        pytest.skip("%s: suite time budget (%.0f s spent + ~%.0f s needed > %.0f s) - run it alone" % (what, spent, need_s, limit))
