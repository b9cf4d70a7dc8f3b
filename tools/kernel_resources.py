#!/usr/bin/env python
"""Register / LDS / occupancy report of the HIP kernels (compile-time, no GPU):
    python tools/kernel_resources.py pytorch-gan_amd/csrc/conv_igemm.hip [-DFLAG ...]
Parses hipcc -Rpass-analysis=kernel-resource-usage; waves/SIMD = min(8, 512 // alloc(VGPR+AGPR)), workgroups/CU also
bounded by 160 KiB LDS."""
import re
import subprocess
import sys


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    seen = set()
    for b in blocks:
        name = b.split("\n")[0].strip().split(" ")[0]   # the remark line ends in " [-Rpass-analysis=...]"
        if name in seen:
            continue
        seen.add(name)

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1

        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(.*", "", dn).replace("void ", "")
        lds = g(r"LDS Size \[bytes/block\]")
        occ = g(r"Occupancy \[waves/SIMD\]")
        wg = min(occ, (160 * 1024) // lds if lds > 0 else 99)
        print("%-96s VGPR %3d AGPR %3d SGPR %3d waves/SIMD %d LDS %6d -> WG/CU %2d scratch %d" % (
            dn[:96], g("VGPRs"), g("AGPRs"), g("SGPRs"), occ, lds, wg, g(r"ScratchSize \[bytes/lane\]")))


if __name__ == "__main__":
    main()
