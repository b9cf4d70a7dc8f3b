#!/usr/bin/env python
"""Per-kernel comparison of the gfx950 device code of one HIP source before and after an edit (no GPU needed) - the check behind
"this refactor does not change the kernels":

    python tools/isa_diff.py <old.hip> <new.hip> [--rename 'regex=>replacement' ...]

Both files are compiled device-only with the flags of csrc/build.py, the code objects disassembled (llvm-objdump), every kernel's
instruction stream stripped of addresses / encodings / branch-target annotations and compared under its demangled name.  --rename maps
old demangled names onto new ones when a template parameter list changed.  For kernels that differ, a second comparison with register
numbers masked tells a pure register re-allocation from a change of the instruction stream."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-gpu-rdc", "--cuda-device-only", "-c",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pytorch-gan_amd", "csrc")]


def kernels(src, tmp, tag):
    co, elf = os.path.join(tmp, tag + ".co"), os.path.join(tmp, tag + ".elf")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", co], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--input=" + co, "--output=" + elf], check=True)
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", elf], check=True,
                         capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]* ?<(.+)>:$", line.strip())
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        t = re.sub(r"<[^>]*>", "", line.split("//")[0]).strip()
        if cur is not None and t:
            out[cur].append(t)
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return {d: out[n] for n, d in zip(names, dem)}


def mask(t):
    return re.sub(r"\b([vsa])\d+\b", r"\1#", re.sub(r"\b([vsa])\[\d+:\d+\]", r"\1[#]", t))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("old")
    ap.add_argument("new")
    ap.add_argument("--rename", action="append", default=[], help="'regex=>replacement' applied to the OLD demangled names")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        a, b = kernels(args.old, tmp, "old"), kernels(args.new, tmp, "new")
    for r in args.rename:
        pat, rep = r.split("=>")
        a = {re.sub(pat, rep, k): v for k, v in a.items()}
    both = sorted(set(a) & set(b))
    print("kernels: old %d, new %d, in both %d" % (len(a), len(b), len(both)))
    for k in sorted(set(a) - set(b)):
        print("only old:", k[:140])
    for k in sorted(set(b) - set(a)):
        print("only new:", k[:140])
    changed = 0
    for k in both:
        if a[k] == b[k]:
            continue
        changed += 1
        if len(a[k]) == len(b[k]) and [mask(x) for x in a[k]] == [mask(x) for x in b[k]]:
            print("register renames only (%d of %d lines): %s" % (sum(x != y for x, y in zip(a[k], b[k])), len(a[k]), k[:120]))
        else:
            print("INSTRUCTION STREAM CHANGED (%d -> %d lines): %s" % (len(a[k]), len(b[k]), k[:120]))
    print("identical: %d, changed: %d" % (len(both) - changed, changed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
