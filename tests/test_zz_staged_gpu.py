"""The hardware self-check of the kernels written without GPU time (pytorch_gan_amd/selfcheck.py), on the hardware.
Runs last (file name): by then the suite has exercised whatever the verdict left in service."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _verdict():
    import pytorch_gan_amd  # noqa: F401
    from pytorch_gan_amd import functional as F
    from pytorch_gan_amd import selfcheck

    if os.environ.get("MIGAN_SELFCHECK", "1") == "0":
        pytest.skip("MIGAN_SELFCHECK=0")
    x = torch.randn(1, 8, 8, 8, device="cuda")
    w = torch.randn(8, 8, 3, 3, device="cuda")
    F.conv2d(x, w, None, 1, (1, 1, 1, 1))   # any conv call obtains the verdict if nothing has yet
    return selfcheck


def test_every_staged_kernel_has_a_verdict_and_the_library_follows_it():
    from pytorch_gan_amd._lib import lib

    sc = _verdict()
    rep = sc.report()
    print("staged kernels:", rep, "cached" if (sc.VERDICT or {}).get("cached") else "probed by this process")
    assert set(rep) == set(sc.BITS) | {"persistent"}
    assert not any(v == "not run" for v in rep.values()), rep
    word = lib.migan_staged(0, 0)
    for k, bit in sc.BITS.items():
        assert bool(word & bit) == (rep[k] == "ok"), (k, rep[k], word)
    from pytorch_gan_amd import steps

    assert steps._K7 == rep["persistent"].startswith("ok")
    off = {k: v for k, v in rep.items() if v.startswith("disabled")}
    if off:
        pytest.xfail("kernels out of service on this hardware (the kernels they replace ran instead): %s" % off)


def test_staged_kernels_agree_with_the_kernels_they_replace_on_the_device():
    """The comparisons of the probe process once more inside this process, for the kernels the probe left in service."""
    from pytorch_gan_amd._lib import lib

    sc = _verdict()
    word = lib.migan_staged(0, 0)
    if word == 0:
        pytest.skip("no staged kernel is in service: %s" % sc.report())
    keep = sc.run_in_process("cuda", word)
    print("largest relative differences:", sc.detail())
    assert keep == word, sc.report()
    assert lib.migan_staged(0, 0) == word


def test_abi_check_binary_on_the_device():
    """tools/abi_check.bin (built by __graft_entry__.build(): C ABI + HIP runtime only, no torch): every staged kernel against the
    kernel it replaces with the bits forced, the one-launch InstanceNorm and the persistent kernels' forward against host fp64, the
    grid barrier at four grid sizes.  Independent of what the self-check left in service - this is the raw hardware comparison
    behind profiles/r03_abi_check.txt; its table (launch times included) is printed."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "abi_check.bin")
    if not os.path.exists(exe):
        pytest.skip("tools/abi_check.bin is not built (tools/build_abi_check.sh)")
    env = {k: v for k, v in os.environ.items() if not k.startswith("MIGAN_")}
    try:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=240, env=env)
    except subprocess.TimeoutExpired:
        pytest.fail("tools/abi_check.bin did not finish in 240 s")
    print(r.stdout)
    if r.returncode in (126, 127) or "error while loading shared libraries" in r.stderr:
        pytest.skip("tools/abi_check.bin could not start here: %s" % r.stderr[-300:])
    if r.returncode < 0:   # a kernel took the process down: exactly what the self-check's probe process exists to absorb
        pytest.xfail("tools/abi_check.bin died with signal %d after: %s" % (-r.returncode, r.stdout.strip().splitlines()[-1:]))
    bad = [ln for ln in r.stdout.splitlines() if "FAIL" in ln or "TIMED OUT" in ln]
    if bad:
        pytest.xfail("hardware comparison out of bound for (the self-check keeps such kernels out of service): %s" % bad[:4])
    assert r.returncode == 0, r.stdout[-600:] + r.stderr[-300:]
