"""Build libmigan.so in-tree with plain hipcc for gfx950 (no torch headers, no cmake).

    python pytorch-gan_amd/csrc/build.py [--force]

The library has a pure C ABI (include/migan.h), so it is independent of the torch wheel's ROCm version;
it only needs libamdhip64.so.7, which at run time resolves to the copy torch already loaded.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["conv_igemm.hip", "conv_dma.hip", "norm.hip", "eltwise.hip", "reduce_loss_adam.hip", "classify.hip", "skinny_mm.hip", "thin_toeplitz.hip", "rgb_conv.hip", "image_pipeline.hip", "critic_fused.hip", "mlp_fused.hip", "fewpix.hip", "conv_c64.hip"]
HEADERS = ["common.h", "conv_geom.h", "conv_valu_fwd.inc", "conv_wgrad_mfma.inc", "conv_valu_wgrad.inc"]   # .inc: pieces of conv_igemm.hip (same translation unit)
OUT = os.path.join(HERE, "libmigan.so")
STAMP = os.path.join(HERE, ".libmigan.stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-gpu-rdc"]
# extra compiler flags for experiments (e.g. MIGAN_CFLAGS=-save-temps); part of the build digest
FLAGS += os.environ.get("MIGAN_CFLAGS", "").split()


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS + ["build.py"]:
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _up_to_date(dig):
    if os.path.exists(OUT) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            return fh.read().strip() == dig
    return False


def build(force=False, verbose=True):
    """Rebuild libmigan.so when the digest of the sources / flags changed (cheap compare otherwise).  Concurrent callers
    (N ranks of torch.distributed.run on a fresh checkout) serialise on a lock file; objects and the library are written
    under process-unique names and moved into place atomically, so nobody can load a half-written library."""
    import fcntl

    dig = _digest()
    if not force and _up_to_date(dig):
        return OUT
    if not os.path.exists(HIPCC):
        if os.path.exists(OUT):
            # No compiler on this box (e.g. a GPU runner without ROCm dev tools): use the shipped binary - but say so when
            # the sources it was built from cannot be shown to be these (MIGAN_STRICT_BUILD=1 turns that into an error)
            msg = "libmigan.so is used as shipped: no hipcc at %s and %s" % (
                HIPCC, "no build stamp travels with it" if not os.path.exists(STAMP) else "its build stamp differs from the sources")
            if os.environ.get("MIGAN_STRICT_BUILD") == "1":
                raise RuntimeError(msg)
            sys.stderr.write("warning: " + msg + "\n")
            return OUT
        raise RuntimeError("hipcc not found at %s and no prebuilt libmigan.so present" % HIPCC)
    with open(os.path.join(HERE, ".libmigan.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _up_to_date(dig):  # another rank built it while we waited
                return OUT
            tag = ".%d" % os.getpid()
            objs, procs = [], []
            for src in SOURCES:
                obj = os.path.join(HERE, src.replace(".hip", ".o"))
                cmd = [HIPCC] + FLAGS + ["-c", os.path.join(HERE, src), "-o", obj + tag]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
                objs.append(obj)
            failed = None
            for src, p in procs:
                out, _ = p.communicate()
                if p.returncode != 0 and failed is None:
                    sys.stderr.write(out.decode())
                    failed = src
            if failed:
                for o in objs:
                    if os.path.exists(o + tag):
                        os.remove(o + tag)
                raise RuntimeError("hipcc failed on %s" % failed)
            for o in objs:
                os.replace(o + tag, o)
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT + tag] + objs
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            os.replace(OUT + tag, OUT)
            with open(STAMP + tag, "w") as fh:
                fh.write(dig)
            os.replace(STAMP + tag, STAMP)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
