"""TEST INFRASTRUCTURE: build tests/hipemu/_build/libmigan_emu.so - the kernel sources of pytorch-gan_amd/csrc compiled as
plain C++ for the host against the execution-model header in tests/hipemu/include (see its first lines for what the model
covers).  Only tests load the result.

The sources are compiled unchanged except for two textual substitutions no header can express (applied to a copy under
_build/, with #line directives so diagnostics point at the original):
  * `extern __shared__ [attr] T name[];`  ->  `T* name = (T*)hipemu::dyn_lds();`   (dynamic LDS of the launch)
  * `asm volatile(...)` statements: `s_waitcnt vmcnt(N)` becomes `hipemu::waitcnt_vm(N)` (the model's LDS-DMA queue lands data only
    when it is waited for), `s_barrier` becomes `hipemu::barrier_raw()` (a barrier WITHOUT the wait the compiler puts in front of
    __syncthreads()), everything else (lgkmcnt, register pinning) is dropped.

    python tests/hipemu/build_emu.py [--force] [source.hip ...]
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pytorch-gan_amd", "csrc")
BUILD = os.path.join(HERE, "_build")
OUT = os.path.join(BUILD, "libmigan_emu.so")
CXX = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O1", "-fPIC", "-ffp-contract=fast", "-fvisibility=hidden", "-pthread", "-w",
         "-I", os.path.join(HERE, "include"), "-I", CSRC, "-DHIPEMU=1",
         # grid-barrier spin bounds of the persistent kernels: workgroups are OS threads here, possibly on a loaded machine
         "-DCF_SPIN_LIMIT=(1u<<26)", "-DMF_SPIN_LIMIT=(1u<<26)",
         # the kernel sources do not include the public header: here every extern "C" definition is compiled AGAINST its prototype in
         # include/migan.h ("conflicting types" when the two drift) - the header is what INTEGRATION.md tells a maintainer to bind
         "-include", os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "migan.h")]


def sources():
    sys.path.insert(0, CSRC)
    try:
        import build as product_build
        return list(product_build.SOURCES), list(product_build.HEADERS)
    finally:
        sys.path.pop(0)


def _drop_asm(text):
    out, i = [], 0
    pat = re.compile(r"\basm\s+volatile\s*\(")
    while True:
        m = pat.search(text, i)
        if not m:
            out.append(text[i:])
            break
        out.append(text[i:m.start()])
        depth, j = 1, m.end()
        while depth:
            c = text[j]
            if c == '"':  # string literal: skip to its end
                j += 1
                while text[j] != '"':
                    j += 2 if text[j] == "\\" else 1
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
            j += 1
        body = text[m.end():j - 1]
        while text[j] in " \t":
            j += 1
        assert text[j] == ";", "asm statement without ';' near: " + text[m.start():j + 20]
        j += 1
        keep_lines = "\n" * body.count("\n")  # keep the line numbering
        rep = ""
        m_cnt = re.search(r"vmcnt\((%0|\d+)\)", body)
        if m_cnt:   # s_waitcnt vmcnt(N): the LDS-DMA queue of the model honours it
            if m_cnt.group(1) == "%0":
                m_op = re.search(r'"n"\s*\(', body)
                assert m_op, "vmcnt(%0) without an \"n\" operand: " + body
                k, depth = m_op.end(), 1
                while depth:
                    depth += {"(": 1, ")": -1}.get(body[k], 0)
                    k += 1
                rep += "hipemu::waitcnt_vm(%s);" % body[m_op.end():k - 1]
            else:
                rep += "hipemu::waitcnt_vm(%s);" % m_cnt.group(1)
        if "s_barrier" in body:
            rep += "hipemu::barrier_raw();"
        out.append(rep + keep_lines)
        i = j
    return "".join(out)


_EXT_SHARED = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w:]*)\s+(\w+)\s*\[\s*\]\s*;")


_STATIC_SHARED = re.compile(r"^([ \t]*)__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?[A-Za-z_][\w:]*\s+([^;]+);", re.M)


def _poison_static_lds(text):
    """`__shared__ T a[N], b[M];` -> the same + `hipemu::poison_lds(&a, sizeof(a), &seen_a); ...` on the same line (see hipemu.cpp)."""
    def rep(m):
        decl, names = m.group(0), []
        depth, cur = 0, ""
        for ch in m.group(2) + ",":
            if ch == "," and depth == 0:
                nm = re.match(r"\s*(\w+)", cur)
                assert nm, "unparsed __shared__ declarator: " + decl
                names.append(nm.group(1))
                cur = ""
                continue
            depth += {"[": 1, "(": 1, "<": 0, "]": -1, ")": -1}.get(ch, 0)
            cur += ch
        calls = "".join(" { static thread_local unsigned long seen_%s = 0; hipemu::poison_lds(&%s, sizeof(%s), &seen_%s); }" % (n, n, n, n)
                        for n in names)
        return decl + calls
    return _STATIC_SHARED.sub(rep, text)


def transform(text, origin):
    text = _drop_asm(text)
    text = _poison_static_lds(text)
    text = _EXT_SHARED.sub(lambda m: "%s* %s = reinterpret_cast<%s*>(hipemu::dyn_lds());" % (m.group(1), m.group(2), m.group(1)), text)
    assert "extern __shared__" not in text, "unhandled dynamic LDS declaration in " + origin
    return '#line 1 "%s"\n' % origin + text


def _digest(srcs, hdrs):
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, s) for s in srcs + hdrs] + [os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.abspath(__file__)]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, only=None, verbose=False):
    """Returns the path of the emulation library (built when the sources, the model or this script changed).  `only`: list of
    source names - a partial library under its own name (quick iterations on one kernel file)."""
    srcs, hdrs = sources()
    out = OUT
    if only:
        srcs = [s for s in srcs if s in only]
        out = os.path.join(BUILD, "libmigan_emu_%s.so" % "_".join(s.split(".")[0] for s in srcs))
    if not os.path.exists(CXX):
        raise RuntimeError("no host clang++ at %s (HIPEMU_CXX overrides)" % CXX)
    os.makedirs(BUILD, exist_ok=True)
    dig = _digest(srcs, hdrs)
    stamp = out + ".stamp"
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return out
    procs, objs = [], []
    for s in srcs + ["hipemu.cpp"]:
        if s == "hipemu.cpp":
            src = os.path.join(HERE, s)
        else:
            src = os.path.join(BUILD, s.replace(".hip", ".emu.cpp"))
            with open(os.path.join(CSRC, s)) as fh:
                raw = fh.read()
            # `#include "x.inc"`: pieces of the same translation unit (kernels with LDS declarations and asm the transform must see)
            raw = re.sub(r'^#include "(\w+\.inc)"[^\n]*$', lambda m: open(os.path.join(CSRC, m.group(1))).read(), raw, flags=re.M)
            text = transform(raw, os.path.join(CSRC, s))
            with open(src, "w") as fh:
                fh.write(text)
        obj = os.path.join(BUILD, os.path.basename(src) + ".o")
        cmd = [CXX] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = None
    for s, p in procs:
        o, _ = p.communicate()
        if p.returncode != 0 and failed is None:
            sys.stderr.write(o.decode()[-6000:])
            failed = s
    if failed:
        raise RuntimeError("host compile of %s failed" % failed)
    tmp = out + ".%d" % os.getpid()
    subprocess.check_call([CXX, "-shared", "-fPIC", "-pthread", "-o", tmp] + objs)
    os.replace(tmp, out)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    print(build(force="--force" in sys.argv, only=args or None, verbose=True))
