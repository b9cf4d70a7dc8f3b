"""Hardware self-check of the staged kernels ("written without hardware time").

Some kernels of libmigan.so entered the tree when no MI355X was reachable: their logic is verified on the host execution
model of tests/hipemu, which cannot see what only the hardware shows (wave scheduling, memory ordering, a launch the device
rejects).  Each of them REPLACES an older HIP kernel that has passed the GPU suite on the hardware, and each sits behind one
bit of `migan_staged()` (include/migan.h); the persistent WGAN-GP kernels (csrc/critic_fused.hip, mlp_fused.hip) sit behind
`steps._K7`.  Before the first conv / norm / step call of a process on the device, `ensure()` obtains a verdict for this
library build on this GPU model:

  * from a PROBE PROCESS (`python -c ... selfcheck._probe_main`): every staged kernel runs once on a geometry that selects it,
    beside the kernel it replaces (the same C entry with the bit cleared), and the two results are compared on the device; then
    one pix2pix step (256x256, batch 1 - the workload that uses all of them) runs with the surviving bits, and twelve WGAN-GP
    iterations run with the persistent kernels.  The probe logs begin / end of every check to a file, so a check that takes the
    process down (a memory fault aborts the process; a launch failure poisons its HIP context) or hangs (time limit) is
    attributed: that kernel loses its bit, the probe is started again for the rest.  The training process itself never runs a
    staged kernel that has not come back from the probe.
  * the verdict is cached under the temp directory, keyed by the library's digest and the GPU's name, behind a file lock:
    the ranks of one job and the processes of a test suite probe once.

A kernel that disagrees is taken out of service for the process (its bit is cleared, a RuntimeWarning names it): the call
then runs on the older HIP kernel.  Nothing here touches the CPU - both sides of every comparison are HIP launches; there is
still no CPU path.  The persistent WGAN-GP kernels additionally stay behind their verify-before-use guard
(steps._CriticFusedPlan.verify / _GeneratorFusedPlan.verify: compared with the op-by-op HIP path per training state).

    MIGAN_SELFCHECK=0          skip the check (the bits keep their MIGAN_* defaults)
    MIGAN_SELFCHECK=inproc     run the comparisons inside this process (no isolation; what the probe process itself does)
    MIGAN_SELFCHECK_TIMEOUT    seconds allowed per probe process (default 120); a probe that runs into it is the last one
    selfcheck.report()         {kernel: "ok" | "not run" | "off (...)" | "disabled: ..."} - bench.py prints it,
                               tests/test_zz_staged_gpu.py shows it

Cost: one short-lived process (~10 s) per library build and GPU model and machine; a cached verdict costs a file read."""
import os
import warnings

import torch

from ._lib import lib

BITS = {"thin_conv_wave": 1, "wgrad_reduce_tr": 2, "midk_tile": 4, "norm_small": 8, "smallk_tile16": 16, "pack_transpose": 32,
        "fewpix_conv": 64}
ENV = {"thin_conv_wave": "MIGAN_THIN_WAVE", "wgrad_reduce_tr": "MIGAN_WGRAD_REDUCE_TR", "midk_tile": "MIGAN_MIDK",
       "norm_small": "MIGAN_NORM_SMALL", "smallk_tile16": "MIGAN_SMALLK_PB16", "pack_transpose": "MIGAN_PACK_TR",
       "fewpix_conv": "MIGAN_FEWPIX"}
ALL = 127

PENDING = os.environ.get("MIGAN_SELFCHECK", "1") != "0"   # read by functional.conv2d / norm / weight_cache_scope
# Until the verdict is in, no staged kernel is selectable at all (a hipGraph captured before any eager call would otherwise
# record them unverified): the word the environment asked for is remembered here and the library's word is cleared.
_ASKED = lib.migan_staged(0, 0)
if PENDING:
    lib.migan_staged(ALL, 0)
_REPORT = dict({k: "not run" for k in BITS}, persistent="not run")
_DETAIL = {}


def report():
    """Outcome per staged kernel: "ok", "not run" (self-check skipped or not reached yet), "off (MIGAN_X=0)" or "disabled: why"."""
    return dict(_REPORT)


def detail():
    """Largest relative difference (staged vs replaced kernel) seen per case: {case: {tensor: rel}}."""
    return {k: dict(v) for k, v in _DETAIL.items()}


def _rel(a, b):
    d = (a.double() - b.double()).norm()
    n = b.double().norm().clamp_min(1e-30)
    return d / n   # stays on the device: ONE synchronisation at the end


def _rand(gen, *shape, scale=1.0):
    return torch.randn(*shape, device=gen.device, generator=gen) * scale


# N, Ci, H, W, Co, k, stride, pads, gather, act, bias | tolerance | the staged bits whose kernels the geometry selects (a check of
# one bit runs the cases tagged with it; execution-model launch counts: tests/test_kernels_emu_cpu.py::test_selfcheck_cases_select_the_staged_kernels)
_CONV = [
    ("patchgan_head", (2, 512, 6, 6, 1, 4, 1, (2, 2, 1, 1), 0, 0, True), 1e-4, 1 | 16),     # fwd thin_conv_wave, dgrad smallk_tile<.,16>
    ("patchgan_head_b1", (1, 256, 9, 9, 1, 4, 1, (2, 2, 1, 1), 0, 0, True), 1e-4, 1 | 16),
    ("first_conv_6ch", (1, 6, 32, 32, 64, 4, 2, (1, 1, 1, 1), 0, 0, False), 1e-4, 4),       # midk_tile, K = 96
    ("first_conv_3ch", (2, 3, 24, 24, 64, 3, 1, (1, 1, 1, 1), 0, 0, True), 1e-4, 4),        # midk_tile, K = 27
    ("unet_inner", (1, 512, 4, 4, 512, 4, 2, (1, 1, 1, 1), 0, 0, True), 1e-4, 2 | 32 | 64), # wgrad_reduce_tr (4 M weights), pack_transpose; with bit 64 the fewpix path (4 rows)
    ("unet_mid", (1, 128, 16, 16, 256, 4, 2, (1, 1, 1, 1), 0, 0, False), 1e-4, 2 | 32),     # pack_transpose (512 k), wgrad slabs
    ("fewpix_16px", (1, 128, 8, 8, 512, 4, 2, (1, 1, 1, 1), 0, 1, True), 1e-4, 64),         # fewpix conv: 16 rows, K = 2048, bias + LeakyReLU
    ("fewpix_1px", (1, 256, 2, 2, 512, 4, 2, (1, 1, 1, 1), 0, 0, False), 1e-4, 64),         # fewpix conv: ONE output pixel (d8 of the U-Net)
    ("fewpix_3x3_s1", (2, 256, 2, 2, 512, 3, 1, (1, 1, 1, 1), 0, 2, True), 1e-4, 64),       # fewpix conv: 3x3 stride 1 (VGG19 tail on small crops)
]
_CONVT = [   # N, Cin, H, W, Cout, act, bias: nn.ConvTranspose2d(Cin, Cout, 4, 2, 1)
    ("fewpix_convT_4px", (1, 512, 2, 2, 128, 2, True), 1e-4, 64),                           # fewpix transposed conv: 4 input pixels, ReLU
    ("fewpix_convT_1px", (1, 512, 1, 1, 256, 0, False), 1e-4, 64),                          # u1 of the U-Net: 1 -> 2x2
]
_NORM = [   # N, C, H, W, act, affine, mask, residual
    ("in_4x4", (1, 512, 4, 4, 0, False, False, False), 1e-4, 8),
    ("in_16x16_lrelu_mask", (2, 64, 16, 16, 1, False, True, False), 2e-2, 8),   # an activation kink may flip one element
    ("in_32x32_affine_res", (1, 32, 32, 32, 0, True, False, True), 1e-4, 8),
]


def _run_conv(F, case, gen):
    N, Ci, H, W, Co, k, stride, pads, gather, act, bias = case
    x = _rand(gen, N, Ci, H, W).requires_grad_(True)
    w = _rand(gen, Co, Ci, k, k, scale=0.1).requires_grad_(True)
    b = _rand(gen, Co).requires_grad_(True) if bias else None
    y = F.conv2d(x, w, b, stride, pads, gather, act, 0.2)
    gy = _rand(gen, *y.shape)
    y.backward(gy)
    out = {"y": y.detach(), "dx": x.grad, "dw": w.grad}
    if bias:
        out["db"] = b.grad
    return {k_: torch.Tensor.contiguous(F._plain(v).detach().clone()) for k_, v in out.items()}


def _run_convt(F, case, gen):
    N, Cin, H, W, Cout, act, bias = case
    x = _rand(gen, N, Cin, H, W).requires_grad_(True)
    w = _rand(gen, Cin, Cout, 4, 4, scale=0.1).requires_grad_(True)
    b = _rand(gen, Cout).requires_grad_(True) if bias else None
    y = F.conv_transpose2d(x, w, b, 2, 1, act, 0.2)
    gy = _rand(gen, *y.shape)
    y.backward(gy)
    out = {"y": y.detach(), "dx": x.grad, "dw": w.grad}
    if bias:
        out["db"] = b.grad
    return {k_: torch.Tensor.contiguous(F._plain(v).detach().clone()) for k_, v in out.items()}


def _run_norm(F, case, gen):
    N, C, H, W, act, affine, mask, res = case
    x = _rand(gen, N, C, H, W).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gamma = (_rand(gen, C) * 0.2 + 1.0).requires_grad_(True) if affine else None
    beta = _rand(gen, C).requires_grad_(True) if affine else None
    r = _rand(gen, N, C, H, W).contiguous(memory_format=torch.channels_last) if res else None
    m = None
    if mask:
        m = (torch.rand(N, C, H, W, device=gen.device, generator=gen) > 0.5).float().mul_(2.0).contiguous(memory_format=torch.channels_last)
    y = F.norm(x, gamma, beta, res=r, instance=True, act=act, slope=0.2, mask=m if F.norm_small_takes(x, True) else None)
    if m is not None and not F.norm_small_takes(x, True):
        y = F.mul_mask(y, m)       # what nn.Dropout does behind the three-launch path
    gy = _rand(gen, *y.shape)
    y.backward(gy)
    out = {"y": y.detach(), "dx": x.grad}
    if affine:
        out.update(dgamma=gamma.grad, dbeta=beta.grad)
    return {k_: F._plain(v).detach().clone() for k_, v in out.items()}


def _all_cases(F, word=ALL):
    """(name, tolerance, runner) of the cases that exercise a kernel of `word`"""
    for name, case, tol, tags in _CONV:
        if tags & word:
            yield name, tol, (lambda gen, c=case: _run_conv(F, c, gen))
    for name, case, tol, tags in _CONVT:
        if tags & word:
            yield name, tol, (lambda gen, c=case: _run_convt(F, c, gen))
    for name, case, tol, tags in _NORM:
        if tags & word:
            yield name, tol, (lambda gen, c=case: _run_norm(F, c, gen))


def _compare(F, device, bits_on, seed=1234):
    """Every case with `bits_on` set and with all staged bits cleared -> {case: (tol, {tensor: rel (device scalar)})}."""
    res = {}
    for name, tol, fn in _all_cases(F, bits_on):
        outs = []
        for word in (bits_on, 0):
            lib.migan_staged(ALL, word)
            gen = torch.Generator(device=device)
            gen.manual_seed(seed)
            outs.append(fn(gen))
        res[name] = (tol, {k: _rel(outs[0][k], outs[1][k]) for k in outs[0]})
    return res


def _bad_cases(res):
    bad = {}
    for name, (tol, rels) in res.items():
        vals = {k: float(v) for k, v in rels.items()}     # the host synchronisation
        seen = _DETAIL.setdefault(name, {})
        for k, v in vals.items():
            seen[k] = v if (k not in seen or v != v or v > seen[k]) else seen[k]
        if not all(v <= tol for v in vals.values()):      # a NaN compares False: bad
            bad[name] = vals
    return bad


# ------------------------------------------------------------------------------------------------ the comparisons
def _quiet_scope(F):
    """Run outside any step scope: -> the saved scope state (restore with _restore_scope)."""
    saved = (F._CACHE_SCOPE, F._SCOPE_OWNER, F._INPUT_GRAD_ONLY)
    F._CACHE_SCOPE, F._SCOPE_OWNER, F._INPUT_GRAD_ONLY = None, None, False
    return saved


def _restore_scope(F, saved):
    F._CACHE_SCOPE, F._SCOPE_OWNER, F._INPUT_GRAD_ONLY = saved


def _check_bits(F, device, word):
    """The cases with `word` set against all bits cleared -> None when they agree, else a description.  A HIP error raised by
    a launch counts as disagreement (the message names it)."""
    try:
        with torch.enable_grad():
            bad = _bad_cases(_compare(F, device, word))
    except Exception as ex:   # noqa: BLE001 - a launch the device rejects must take the kernel out, not the caller down
        return "%s: %s" % (type(ex).__name__, str(ex)[:200])
    if bad:
        name, vals = next(iter(bad.items()))
        return "differs from the kernel it replaces on %s (%s)" % (name, ", ".join("%s %.2e" % kv for kv in vals.items()))
    return None


def run_in_process(device, start=None, known_ok=0, log=None):
    """Compare every staged kernel of `start` (default: the library's current word) with the kernel it replaces, one bit at a
    time, then the survivors together; `known_ok` bits are taken as verified (an earlier probe did them).  Sets the library's
    word to the survivors, fills report(), -> the word.  `log(event, name, **kw)` is told begin / end of every check."""
    from . import functional as F

    log = log or (lambda *a, **k: None)
    device = torch.device(device)
    if start is None:
        start = lib.migan_staged(0, 0)
    saved = _quiet_scope(F)
    keep = start & known_ok
    try:
        for k, bit in BITS.items():
            if not start & bit:
                continue
            if known_ok & bit:
                _REPORT[k] = "ok"
                continue
            log("begin", k)
            why = _check_bits(F, device, bit)
            _REPORT[k] = "ok" if why is None else "disabled: " + why
            if why is None:
                keep |= bit
            log("end", k, ok=why is None, why=why)
        if keep & (keep - 1):   # two or more survivors: they must also agree together
            log("begin", "combined")
            why = _check_bits(F, device, keep)
            if why is not None:
                for k, bit in BITS.items():
                    if keep & bit:
                        _REPORT[k] = "disabled: only in combination with the other staged kernels: " + why
                keep = 0
            log("end", "combined", ok=why is None, why=why)
    finally:
        lib.migan_staged(ALL, keep)
        _restore_scope(F, saved)
    return keep


def _workload_pass(device, word):
    """One pix2pix training step (pix2pix.py:141-190 at 256x256, batch 1: the step that launches every staged kernel on the
    shapes of the bench) with the staged kernels of `word`, and the same step from the same weights without them -> None when
    both run and their losses agree, else why not."""
    import numpy as np

    from . import models, steps
    from . import nn as gnn

    outs = []
    for w in (word, 0):
        lib.migan_staged(ALL, w)
        torch.manual_seed(0)
        G, D = models.Pix2pixGenerator(), models.Pix2pixDiscriminator()
        G.apply(models.init_normal_dcgan)
        D.apply(models.init_normal_dcgan)
        G, D = G.to(device), D.to(device)
        s = steps.make_pix2pix_state(G, D, 256)
        rng = np.random.RandomState(555)
        a = torch.from_numpy(rng.uniform(-1, 1, (1, 3, 256, 256)).astype(np.float32)).to(device)
        b = torch.from_numpy(rng.uniform(-1, 1, (1, 3, 256, 256)).astype(np.float32)).to(device)
        gnn.manual_seed(1234)   # the same dropout draws on both sides
        out = None
        for _ in range(2):   # the second step runs on the per-step plans (batched packs and masks) the first one recorded
            out = steps.pix2pix_step(s, a, b)
        outs.append({k: float(v) for k, v in out.items() if "loss" in k})
        del G, D, s, out
    lib.migan_staged(ALL, word)
    for k, v in outs[0].items():
        r = outs[1][k]
        if not (v == v and abs(v - r) <= 2e-2 * max(1.0, abs(r))):   # dropout draws are the same stream; summation orders differ
            return "pix2pix step: %s %.6g with the staged kernels, %.6g without" % (k, v, r)
    return None


def _persistent_pass(device):
    """Twelve WGAN-GP iterations (wgan_gp.py:146-193, batch 64, the reference's MLPs) with the persistent kernels in service
    behind their own verify-before-use guards -> (None | why, {guard: verified})."""
    import numpy as np

    from . import models, steps

    if not steps._K7:
        return None, {}
    torch.manual_seed(0)
    G, D = models.MlpGenerator((1, 32, 32), 100).to(device), models.MlpCritic((1, 32, 32)).to(device)
    s = steps.make_wgan_gp_state(G, D)
    rng = np.random.RandomState(777)
    real = torch.from_numpy(rng.uniform(-1, 1, (64, 1, 32, 32)).astype(np.float32)).to(device)
    out = {}
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for i in range(12):
            z = torch.from_numpy(rng.normal(0, 1, (64, 100)).astype(np.float32)).to(device)
            alpha = torch.from_numpy(rng.random_sample((64, 1, 1, 1)).astype(np.float32)).to(device)
            out.update(steps.wgan_gp_step(s, real, i, z, alpha))
        vals = {k: float(v) for k, v in out.items() if "loss" in k or k == "gp"}
    flags = {"critic": bool(getattr(getattr(s, "_k7_plan", None), "verified", False)),
             "generator_forward": bool(getattr(getattr(s, "_k7_gen_plan", None), "verified", False)),
             "generator_iteration": bool(getattr(getattr(s, "_k7_gen_plan", None), "step_verified", False))}
    if not all(v == v and abs(v) < 1e6 for v in vals.values()):
        return "non-finite WGAN-GP losses with the persistent kernels: %s" % vals, flags
    notes = "; ".join(str(w.message)[:160] for w in caught if "pytorch_gan_amd" in str(w.message))
    if notes:
        flags["notes"] = notes
    return None, flags


# ------------------------------------------------------------------------------------------------ the probe process
def _probe_main(argv):
    """Entry of the probe process: argv = [log path, device index, bits to check, bits already verified, stages]."""
    import json

    path, index, start, known_ok, stages = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), argv[4].split(",")
    fh = open(path, "a")

    def log(event, name, **kw):
        fh.write(json.dumps(dict(kw, event=event, name=name)) + "\n")
        fh.flush()
        os.fsync(fh.fileno())

    global PENDING
    PENDING = False
    torch.cuda.set_device(index)
    device = torch.device("cuda", index)
    log("begin", "device")
    torch.zeros(1, device=device).add_(1.0)
    torch.cuda.synchronize()
    log("end", "device", ok=True, gpu=torch.cuda.get_device_name(index))
    _probe_stages(device, start, known_ok, stages, log, torch.cuda.synchronize)
    log("end", "probe", ok=True)
    fh.close()


def _probe_stages(device, start, known_ok, stages, log, sync):
    """What a probe process does once it has a device; -> the staged word it leaves set."""
    keep = start & known_ok
    lib.migan_staged(ALL, keep)   # (this process starts from the environment's word: nothing unverified below)
    if "bits" in stages:
        keep = run_in_process(device, start, known_ok, log)
    if "workload" in stages and keep:
        log("begin", "workload")
        try:
            why = _workload_pass(device, keep)
        except Exception as ex:   # noqa: BLE001
            why = "%s: %s" % (type(ex).__name__, str(ex)[:200])
        sync()
        log("end", "workload", ok=why is None, why=why)
    if "persistent" in stages:
        log("begin", "persistent")
        try:
            why, flags = _persistent_pass(device)
        except Exception as ex:   # noqa: BLE001
            why, flags = "%s: %s" % (type(ex).__name__, str(ex)[:200]), {}
        sync()
        log("end", "persistent", ok=why is None, why=why, flags=flags)
    return keep


def _spawn_probe(index, start, known_ok, stages, timeout):
    """Run one probe process -> (records, how it ended: "exit 0" | "exit N" | "signal N" | "timeout after S s")."""
    import json
    import subprocess
    import sys
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # the directory that holds pytorch_gan_amd.py
    fd, path = tempfile.mkstemp(prefix="migan_probe_", suffix=".jsonl")
    os.close(fd)
    code = ("import sys; sys.path.insert(0, %r); import pytorch_gan_amd; from pytorch_gan_amd import selfcheck; "
            "selfcheck._probe_main(sys.argv[1:])" % root)
    env = dict(os.environ, MIGAN_SELFCHECK="0", PYTHONWARNINGS="ignore")
    cmd = [sys.executable, "-c", code, path, str(index), str(start), str(known_ok), ",".join(stages)]
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
    except OSError as ex:
        os.unlink(path)
        return [], "could not start (%s)" % ex
    try:
        _, err = p.communicate(timeout=timeout)
        how = "exit %d" % p.returncode if p.returncode >= 0 else "signal %d" % -p.returncode
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, 9)   # its own session: the exact process group we started
        except OSError:
            pass
        _, err = p.communicate()
        how = "timeout after %d s" % timeout
    records = []
    try:
        with open(path) as fh:
            for line in fh:
                try:
                    records.append(json.loads(line))
                except ValueError:
                    pass   # a torn last line
        os.unlink(path)
    except OSError:
        pass
    if how != "exit 0" and err:
        tail = err.decode("utf-8", "replace").strip().splitlines()[-1:]
        how += " (%s)" % (tail[0][:160] if tail else "")
    return records, how


def _digest_records(records):
    """-> ({name: (ok, why, extra)} for finished checks, name of the check that was begun and never ended or None)."""
    done, open_ = {}, None
    for r in records:
        if r.get("event") == "begin":
            open_ = r.get("name")
        elif r.get("event") == "end":
            done[r.get("name")] = (bool(r.get("ok")), r.get("why"), r)
            if open_ == r.get("name"):
                open_ = None
    return done, open_


def probe(index, start, want_persistent, spawn=_spawn_probe, timeout=None, max_spawns=5):
    """The verdict of the probe processes for the staged word `start`: {"bits": surviving word, "report": {kernel: text},
    "persistent": bool}.  A check that ends its process is charged to the kernel it was running and the probe restarts for
    the rest; after `max_spawns` processes whatever is still unverified stays off."""
    timeout = timeout or float(os.environ.get("MIGAN_SELFCHECK_TIMEOUT", "120"))
    rep = {k: ("not run" if start & bit else "off (%s=0)" % ENV[k]) for k, bit in BITS.items()}
    rep["persistent"] = "not run" if want_persistent else "off (MIGAN_K7=0)"
    todo, ok_bits = start, 0
    need_workload, need_persistent = True, bool(want_persistent)
    spawns, reached = 0, True
    while spawns < max_spawns and (todo or (need_workload and ok_bits) or need_persistent):
        spawns += 1
        stages = (["bits"] if todo else []) + (["workload"] if need_workload else []) + (["persistent"] if need_persistent else [])
        records, how = spawn(index, todo | ok_bits, ok_bits, stages, timeout)
        done, died_in = _digest_records(records)
        if how.startswith("timeout"):
            spawns = max_spawns   # a hung check may leave the device slow to recover: what it did not verify stays off
        if "device" not in done:
            # the process never reached the GPU: nothing can be said about any kernel - leave everything unverified off
            for k, bit in BITS.items():
                if todo & bit:
                    rep[k] = "disabled: the probe process did not reach the device (%s)" % how
            if need_persistent:
                rep["persistent"] = "disabled: the probe process did not reach the device (%s)" % how
            todo, need_persistent = 0, False
            reached = reached and spawns > 1
            break
        for k, bit in BITS.items():
            if todo & bit and k in done:
                todo &= ~bit
                if done[k][0]:
                    ok_bits |= bit
                    rep[k] = "ok"
                else:
                    rep[k] = "disabled: " + str(done[k][1])
        if died_in in BITS:
            todo &= ~BITS[died_in]
            rep[died_in] = "disabled: its check ended the probe process (%s)" % how
            continue
        if "combined" in done or died_in == "combined":
            if died_in == "combined" or not done["combined"][0]:
                why = ("their combined check ended the probe process (%s)" % how) if died_in == "combined" else str(done["combined"][1])
                for k, bit in BITS.items():
                    if ok_bits & bit:
                        rep[k] = "disabled: " + why
                ok_bits, need_workload = 0, False
                if died_in == "combined":
                    continue
        if "workload" in done or died_in == "workload":
            need_workload = False
            if died_in == "workload" or not done["workload"][0]:
                why = ("the pix2pix step with the staged kernels ended the probe process (%s)" % how) if died_in == "workload" \
                    else str(done["workload"][1])
                for k, bit in BITS.items():
                    if ok_bits & bit:
                        rep[k] = "disabled: " + why
                ok_bits = 0
                if died_in == "workload":
                    continue
        if "persistent" in done or died_in == "persistent":
            need_persistent = False
            if died_in == "persistent":
                rep["persistent"] = "disabled: the WGAN-GP iterations with the persistent kernels ended the probe process (%s)" % how
            elif not done["persistent"][0]:
                rep["persistent"] = "disabled: " + str(done["persistent"][1])
            else:
                flags = done["persistent"][2].get("flags") or {}
                rep["persistent"] = "ok" + ("" if all(v is True for k, v in flags.items() if k != "notes") else
                                            " (guards: %s)" % ", ".join("%s=%s" % kv for kv in sorted(flags.items())))
            continue
        if died_in is None and "probe" in done:
            continue   # a complete pass: the loop condition decides whether anything is left
        if died_in is not None or "probe" not in done:
            # ended somewhere unattributed (between checks): do not loop on it
            break
    for k, bit in BITS.items():
        if todo & bit and rep[k] == "not run":
            rep[k] = "disabled: not verified (the probe was stopped after an earlier check ended or hung it)"
    if need_workload and ok_bits:
        for k, bit in BITS.items():
            if ok_bits & bit:
                rep[k] = "disabled: the pix2pix step with the staged kernels was not reached (the probe was stopped earlier)"
        ok_bits = 0
    if need_persistent:
        rep["persistent"] = "disabled: not verified (the probe was stopped after an earlier check ended or hung it)"
    # "definitive": every probe process got as far as the device - a verdict worth keeping for this machine (a probe that could
    # not even start, or found no GPU, says nothing about the kernels and is asked again by the next process)
    return {"bits": ok_bits, "report": rep, "persistent": rep["persistent"].startswith("ok"), "definitive": reached}


# ------------------------------------------------------------------------------------------------ verdict cache
def _cache_path(index, start, want_persistent):
    import hashlib
    import tempfile

    from ._lib import LIB_PATH

    h = hashlib.sha256()
    with open(LIB_PATH, "rb") as fh:
        h.update(fh.read())
    h.update(("|%s|%d|%d|v1" % (torch.cuda.get_device_name(index), start, int(want_persistent))).encode())
    return os.path.join(tempfile.gettempdir(), "migan_selfcheck_%s.json" % h.hexdigest()[:20])


def _cached_verdict(index, start, want_persistent, make):
    """The verdict for (library build, GPU model, requested word) from the cache file, or from make() - once per machine: the
    ranks of a job and the processes of a test run meet at the file lock."""
    import fcntl
    import json

    path = _cache_path(index, start, want_persistent)
    with open(path + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            try:
                with open(path) as fh:
                    v = json.load(fh)
                if isinstance(v.get("bits"), int) and isinstance(v.get("report"), dict):
                    v["cached"] = True
                    return v
            except (OSError, ValueError):
                pass
            v = make()
            if v.get("definitive", True):
                tmp = "%s.%d.tmp" % (path, os.getpid())
                with open(tmp, "w") as fh:
                    json.dump(v, fh)
                os.replace(tmp, path)
            return v
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


_LOCK = __import__("threading").Lock()
VERDICT = None   # what ensure() applied: {"bits", "report", "persistent", "cached"?}


def ensure(device=None):
    """Obtain and apply the verdict once per process (no-op afterwards, and while a hipGraph capture is in progress)."""
    global PENDING, VERDICT
    if not PENDING:
        return
    if not torch.cuda.is_available():
        return
    if torch.cuda.is_current_stream_capturing():
        return                                  # stays pending: eager warm-up steps precede every capture of steps.py
    with _LOCK:
        if not PENDING:
            return
        PENDING = False
        from . import steps

        device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if device.type != "cuda":
            return
        index = device.index if device.index is not None else torch.cuda.current_device()
        start = _ASKED
        for k, bit in BITS.items():
            _REPORT[k] = "not run" if start & bit else "off (%s=0)" % ENV[k]
        _REPORT["persistent"] = "not run" if steps._K7 else "off (MIGAN_K7=0)"
        if start == 0 and not steps._K7:
            return
        if os.environ.get("MIGAN_SELFCHECK") == "inproc":
            run_in_process(device, start)
            v = {"bits": lib.migan_staged(0, 0), "report": dict(_REPORT), "persistent": steps._K7}
        else:
            try:
                v = _cached_verdict(index, start, steps._K7, lambda: probe(index, start, steps._K7))
            except Exception as ex:   # noqa: BLE001 - no verdict (temp directory not writable, ...): nothing unverified runs
                v = {"bits": 0, "persistent": False,
                     "report": dict({k: "disabled: no verdict (%s: %s)" % (type(ex).__name__, str(ex)[:120]) for k in BITS},
                                    persistent="disabled: no verdict (%s)" % type(ex).__name__)}
        VERDICT = v
        lib.migan_staged(ALL, int(v["bits"]) & start)
        _REPORT.update(v["report"])
        if steps._K7 and not v.get("persistent", False):
            steps._K7 = False
        off = [k for k in _REPORT if _REPORT[k].startswith("disabled")]
        if off:
            warnings.warn("pytorch_gan_amd: staged kernels taken out of service by the hardware self-check (the kernels they replace "
                          "run instead): " + "; ".join("%s - %s" % (k, _REPORT[k]) for k in off), RuntimeWarning, stacklevel=2)
