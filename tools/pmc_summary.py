#!/usr/bin/env python
"""Per-kernel means of rocprofv3 --pmc counter CSVs:  python tools/pmc_summary.py <dir> [<dir> ...]
Every directory holds one pass (*_counter_collection.csv); counters of the same kernel from different passes are
joined by kernel name + grid size (the dispatch sequence of the profiled command is the same in every pass)."""
import collections
import csv
import glob
import os
import re
import sys


def short(nm):
    nm = re.sub(r"\(.*", "", nm).replace("void ", "")
    return nm


def main():
    tab = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                key = (short(r["Kernel_Name"]), r.get("Grid_Size", ""))
                tab[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, grid), ctrs in sorted(tab.items()):
        if not any(s in k for s in ("igemm", "wgrad", "norm_", "thin_")):
            continue
        vals = {c: sum(v[-5:]) / len(v[-5:]) for c, v in ctrs.items()}
        n = max(len(v) for v in ctrs.values())
        print("%s grid=%s (n=%d)" % (k[:90], grid, n))
        print("   " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(vals.items())))
        wc = vals.get("SQ_WAVE_CYCLES")
        if wc:
            parts = ["%s %.1f%%" % (c[3:], 100 * vals[c] / wc) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                                                          "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                                                                          "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM") if c in vals]
            print("   of WAVE_CYCLES: " + ", ".join(parts))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (64 per v_mfma_f32_32x32x2_f32: r01_pmc_igemm128.txt);
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs -> busy fraction of the matrix pipes = BUSY / (1024 * GUI_ACTIVE / 8)
            print("   mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE) = %.3f" % (vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * vals["GRBM_GUI_ACTIVE"])))
        if "SQ_INST_LEVEL_VMEM" in vals and vals.get("SQ_INSTS_VMEM_RD"):
            print("   mean VMEM latency ~ %.0f cycles (INST_LEVEL_VMEM / INSTS_VMEM_RD)" % (vals["SQ_INST_LEVEL_VMEM"] / vals["SQ_INSTS_VMEM_RD"]))
        if "TCC_HIT_sum" in vals:
            print("   L2 hit rate %.3f" % (vals["TCC_HIT_sum"] / max(vals["TCC_HIT_sum"] + vals.get("TCC_MISS_sum", 0), 1)))


if __name__ == "__main__":
    main()
