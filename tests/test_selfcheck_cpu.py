"""The orchestration of the hardware self-check (pytorch_gan_amd/selfcheck.py) without a GPU: how probe() turns the log of
its probe processes into a verdict - every way a probe process can end - and the verdict cache.  The comparisons themselves run
on the execution model in tests/test_kernels_emu_cpu.py::test_selfcheck_*; on the hardware in tests/test_zz_staged_gpu.py."""
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_gan_amd  # noqa: E402,F401
from pytorch_gan_amd import selfcheck  # noqa: E402

ALL = selfcheck.ALL
NAMES = list(selfcheck.BITS)


def _script(*runs):
    """A spawn() that plays back one scripted probe process per call: each run = (list of (event, name, fields), how)."""
    calls = []

    def spawn(index, start, known_ok, stages, timeout):
        calls.append({"start": start, "known_ok": known_ok, "stages": list(stages)})
        recs, how = runs[len(calls) - 1](start, known_ok, stages)
        return [dict(f, event=e, name=n) for e, n, f in recs], how

    spawn.calls = calls
    return spawn


def _full_run(fail=(), die_in=None, workload=True, persistent=True, flags=None):
    """A probe process that checks what it is asked: `fail` bits disagree, `die_in` names the check that ends the process."""
    def run(start, known_ok, stages):
        recs = [("begin", "device", {}), ("end", "device", {"ok": True, "gpu": "AMD Instinct MI355X"})]
        keep = start & known_ok

        def check(name, ok, **f):
            recs.append(("begin", name, {}))
            if die_in == name:
                raise StopIteration
            recs.append(("end", name, dict(f, ok=ok, why=None if ok else "differs")))

        try:
            if "bits" in stages:
                for k, bit in selfcheck.BITS.items():
                    if start & bit and not known_ok & bit:
                        check(k, k not in fail)
                        if k not in fail:
                            keep |= bit
                if keep & (keep - 1):
                    check("combined", "combined" not in fail)
                    if "combined" in fail:
                        keep = 0
            if "workload" in stages and keep:
                check("workload", workload)
            if "persistent" in stages:
                check("persistent", persistent, flags=flags or {"critic": True, "generator_forward": True, "generator_iteration": True})
            recs.append(("end", "probe", {"ok": True}))
        except StopIteration:
            return recs, "signal 6 (Memory access fault by GPU node-2)"
        return recs, "exit 0"
    return run


def test_clean_probe_gives_every_kernel():
    spawn = _script(_full_run())
    v = selfcheck.probe(0, ALL, True, spawn=spawn)
    assert v["bits"] == ALL and v["persistent"] is True and v["definitive"] is True
    assert v["report"] == dict({k: "ok" for k in NAMES}, persistent="ok")
    assert len(spawn.calls) == 1 and spawn.calls[0]["stages"] == ["bits", "workload", "persistent"]


def test_disagreeing_kernel_loses_its_bit_only():
    spawn = _script(_full_run(fail=("midk_tile",)))
    v = selfcheck.probe(0, ALL, True, spawn=spawn)
    assert v["bits"] == ALL & ~selfcheck.BITS["midk_tile"]
    assert v["report"]["midk_tile"].startswith("disabled: differs") and v["report"]["norm_small"] == "ok"
    assert len(spawn.calls) == 1


def test_a_check_that_ends_the_process_is_charged_and_the_probe_restarts():
    spawn = _script(_full_run(die_in="norm_small"), _full_run())
    v = selfcheck.probe(0, ALL, True, spawn=spawn)
    assert v["bits"] == ALL & ~selfcheck.BITS["norm_small"]
    assert "ended the probe process (signal 6" in v["report"]["norm_small"]
    assert all(v["report"][k] == "ok" for k in NAMES if k != "norm_small") and v["persistent"]
    # the second process re-checks only what was not verified yet (bits behind the one that died), the others are known
    second = spawn.calls[1]
    done_before = selfcheck.BITS["thin_conv_wave"] | selfcheck.BITS["wgrad_reduce_tr"] | selfcheck.BITS["midk_tile"]
    assert second["known_ok"] == done_before and second["start"] == ALL & ~selfcheck.BITS["norm_small"]


def test_workload_crash_takes_all_staged_kernels_out_but_still_probes_the_persistent_ones():
    spawn = _script(_full_run(die_in="workload"), _full_run())
    v = selfcheck.probe(0, ALL, True, spawn=spawn)
    assert v["bits"] == 0 and v["persistent"] is True
    assert all("pix2pix step with the staged kernels ended the probe process" in v["report"][k] for k in NAMES)
    assert spawn.calls[1]["stages"] == ["persistent"] and spawn.calls[1]["start"] == 0


def test_persistent_crash_keeps_the_staged_kernels():
    spawn = _script(_full_run(die_in="persistent"))
    v = selfcheck.probe(0, ALL, True, spawn=spawn)
    assert v["bits"] == ALL and v["persistent"] is False
    assert "ended the probe process" in v["report"]["persistent"] and len(spawn.calls) == 1


def test_persistent_guards_are_reported():
    spawn = _script(_full_run(flags={"critic": True, "generator_forward": True, "generator_iteration": False}))
    v = selfcheck.probe(0, ALL, True, spawn=spawn)
    assert v["persistent"] is True and "generator_iteration=False" in v["report"]["persistent"]


def test_probe_that_never_reaches_the_device_leaves_everything_off():
    spawn = _script(lambda start, known, stages: ([], "exit 1 (RuntimeError: No HIP GPUs are available)"))
    v = selfcheck.probe(0, ALL, True, spawn=spawn)
    assert v["bits"] == 0 and v["persistent"] is False
    assert all("did not reach the device" in t for t in v["report"].values())
    assert len(spawn.calls) == 1 and v["definitive"] is False   # ... and is not worth caching


def test_env_switched_off_bits_are_not_probed():
    start = ALL & ~selfcheck.BITS["pack_transpose"]
    spawn = _script(_full_run())
    v = selfcheck.probe(0, start, False, spawn=spawn)
    assert v["bits"] == start and v["report"]["pack_transpose"] == "off (MIGAN_PACK_TR=0)"
    assert v["report"]["persistent"] == "off (MIGAN_K7=0)" and spawn.calls[0]["stages"] == ["bits", "workload"]


def test_runaway_crashes_are_bounded():
    dies = [_full_run(die_in=k) for k in NAMES]
    spawn = _script(*dies)
    v = selfcheck.probe(0, ALL, True, spawn=spawn, max_spawns=3)
    assert len(spawn.calls) == 3 and v["bits"] == 0 and v["persistent"] is False
    assert all(t.startswith("disabled") for t in v["report"].values())


def test_a_hung_check_is_the_last_probe():
    def hung(start, known_ok, stages):
        recs, _ = _full_run(die_in="norm_small")(start, known_ok, stages)
        return recs, "timeout after 120 s"

    spawn = _script(hung, _full_run())
    v = selfcheck.probe(0, ALL, True, spawn=spawn)
    assert len(spawn.calls) == 1                      # no second process on a device that may still be recovering
    assert v["bits"] == 0 and v["persistent"] is False
    assert "timeout after 120 s" in v["report"]["norm_small"]
    assert all(t.startswith("disabled") for t in v["report"].values())


def test_real_probe_process_without_a_gpu_is_contained():
    """The real child process on this machine (no GPU): it ends before reaching a device; the parent reads that as 'no verdict'."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("this machine has a GPU")
    records, how = selfcheck._spawn_probe(0, ALL, 0, ["bits"], 120)
    assert how.startswith("exit 1"), how
    done, died_in = selfcheck._digest_records(records)
    assert "device" not in done
    v = selfcheck.probe(0, ALL, True, spawn=lambda *a: (records, how))
    assert v["bits"] == 0 and not v["persistent"]


def test_verdict_cache_probes_once(tmp_path, monkeypatch):
    monkeypatch.setattr(selfcheck, "_cache_path", lambda *a: str(tmp_path / "verdict.json"))
    made = []

    def make():
        made.append(1)
        return {"bits": 5, "report": {"x": "ok"}, "persistent": True}

    a = selfcheck._cached_verdict(0, ALL, True, make)
    b = selfcheck._cached_verdict(0, ALL, True, make)
    assert made == [1] and a["bits"] == b["bits"] == 5 and b.get("cached") is True and "cached" not in a
    (tmp_path / "verdict.json").unlink()
    d = selfcheck._cached_verdict(0, ALL, True, lambda: {"bits": 0, "report": {}, "persistent": False, "definitive": False})
    assert d["bits"] == 0 and not (tmp_path / "verdict.json").exists()     # "the probe never saw a GPU" is not kept
    made.clear()
    made.append(1)
    selfcheck._cached_verdict(0, ALL, True, lambda: {"bits": 5, "report": {"x": "ok"}, "persistent": True})
    (tmp_path / "verdict.json").write_text("{ torn")
    c = selfcheck._cached_verdict(0, ALL, True, make)
    assert made == [1, 1] and c["bits"] == 5
    assert json.loads((tmp_path / "verdict.json").read_text())["bits"] == 5
