#!/usr/bin/env python
"""rocprofv3 --pmc passes over `bench.py --workload W --pmc-log <pass_dir>/segments.json`  ->  profiles/<round>_pmc_kernels.json
(what bench.py's `roofline` object reads: HBM-side bytes per launch, MFMA-busy fraction and kernel symbol of every conv-family /
normalisation call of the TRAINING STEP ITSELF, keyed by bench.py's roofline group names).

    python tools/pmc_step.py <round> <pass_dir> [<pass_dir> ...]          (merges into an existing profiles/<round>_pmc_kernels.json)

Each <pass_dir> holds ONE rocprofv3 invocation (`--pmc <counters> --kernel-trace --output-format csv`, never combined with sys / hip
traces).  Attribution needs no marker kernels: bench.py logs, for every call of a roofline group, the ordinals [first, last) of the
library launches it issued (the C ABI's launch counter, counted from process start); the pass's dispatch list, filtered to the
library's kernels (everything that is not an ATen / runtime kernel) and taken in dispatch order, is that same sequence - the tool
refuses a pass whose library-dispatch count differs from the logged total.

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are reported in KiB; FETCH_SIZE counts
64 B per 128 B request on gfx950, so HBM-side read bytes = 2 x FETCH_SIZE x 1024 (checked in every pass that has both on the library's
own streaming kernels, recorded under "_calibration"); WRITE_SIZE x 1024 as is.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the
1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: mfma_busy_frac = BUSY / (128 x GUI_ACTIVE)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FOREIGN = ("at::native", "at::cuda", "rocclr", "rocprim", "hipcub", "at_cuda_detail", "c10::", "Cijk_", "rccl", "nccl")


def short(nm):
    return re.sub(r"\(.*", "", nm).replace("void ", "")[:120]


def is_library(name):
    return not any(f in name for f in FOREIGN)


def read_pass(d):
    log = json.load(open(os.path.join(d, "segments.json")))
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = [r for f in files for r in csv.DictReader(open(f))]
    by_disp = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        e = by_disp.setdefault(int(r["Dispatch_Id"]), {"name": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"]), "ctr": {},
                                                       "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e["ctr"][r["Counter_Name"]] = float(r["Counter_Value"])
    libk = [e for e in by_disp.values() if is_library(e["name"])]
    return log, libk


def main():
    rnd, dirs = sys.argv[1], sys.argv[2:]
    path = os.path.join(ROOT, "profiles", "%s_pmc_kernels.json" % rnd)
    try:
        out = json.load(open(path), object_pairs_hook=collections.OrderedDict)
    except (OSError, ValueError):
        out = collections.OrderedDict()
    groups = collections.OrderedDict()
    sources = out.get("_source", {}).get("passes", [])
    for d in dirs:
        log, libk = read_pass(d)
        if len(libk) != log["total_launches"]:
            print("REFUSED %s: %d library dispatches in the pass, %d launches logged by bench.py" % (d, len(libk), log["total_launches"]))
            continue
        sources.append(os.path.relpath(os.path.abspath(d), ROOT))
        for seg in log["segments"]:
            g = groups.setdefault(seg["group"], {"workload": log["workload"], "calls": {}, "dense": seg["dense"], "executed": seg["executed"],
                                                 "_k": {}, "_c": {}, "_kdir": d, "_cdir": {}})
            g["calls"][d] = g["calls"].get(d, 0) + 1
            for e in libk[seg["first"]:seg["last"]]:
                if g["_kdir"] == d:
                    k = g["_k"].setdefault(e["name"], {"n": 0, "ns": 0})
                    k["n"] += 1
                    k["ns"] += e["t"]
                for cn, cv in e["ctr"].items():
                    if g["_cdir"].setdefault(cn, d) == d:
                        g["_c"][cn] = g["_c"].get(cn, 0.0) + cv
    for name, g in groups.items():
        if not g["_k"]:
            continue
        ncall = float(g["calls"][g["_kdir"]])
        c = {k: v / float(g["calls"][g["_cdir"][k]]) for k, v in g["_c"].items()}
        main_k = max(g["_k"], key=lambda k: g["_k"][k]["ns"])
        ent = collections.OrderedDict(workload=g["workload"], symbol=main_k,
                                      kernels_per_launch={k: round(v["n"] / ncall, 2) for k, v in g["_k"].items()},
                                      launch_us_under_pmc=round(sum(v["ns"] for v in g["_k"].values()) / ncall / 1e3, 1),
                                      where="in-step (bench.py --pmc-log)")
        if g["executed"] >= 0:
            ent["dense_gflop_per_launch"] = round(g["dense"] / 1e9, 3)
            ent["executed_gflop_per_launch"] = round(g["executed"] / 1e9, 3)
        else:
            ent["algorithmic_mb_per_call"] = round(g["dense"] / 1e6, 2)
        if "FETCH_SIZE" in c:
            ent["fetch_size_kb_reported"] = round(c["FETCH_SIZE"], 1)
        if "WRITE_SIZE" in c:
            ent["write_size_kb_reported"] = round(c["WRITE_SIZE"], 1)
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            ent["hbm_bytes_per_launch"] = int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            ent["mfma_busy_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"])
            ent["grbm_gui_active"] = round(c["GRBM_GUI_ACTIVE"])
            ent["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * c["GRBM_GUI_ACTIVE"]), 4)
        if c.get("SQ_WAVE_CYCLES"):
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in c:
                    ent[k.lower() + "_frac"] = round(c[k] / c["SQ_WAVE_CYCLES"], 4)
        if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            ent["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        out[name] = ent
    # calibration: the streaming normalisation calls move a known number of bytes (4 B x elements x passes)
    cal = []
    for name, ent in out.items():
        if name.startswith("norm_apply[") and "hbm_bytes_per_launch" in ent and ent.get("algorithmic_mb_per_call", 0) >= 32:
            cal.append({"call": name, "algorithmic_mb": ent["algorithmic_mb_per_call"],
                        "counted_mb (2 x FETCH + WRITE)": round(ent["hbm_bytes_per_launch"] / 1e6, 2)})
    out["_calibration"] = {"note": "norm_apply reads and writes its tensor once: 2 x FETCH_SIZE KiB + WRITE_SIZE KiB reproduces 8 B per element",
                           "samples": cal[:8]}
    # the library the passes ran on: bench.py compares this digest with the one it loads and flags the counters `stale` otherwise
    stamp = os.path.join(ROOT, "pytorch-gan_amd", "csrc", ".libmigan.stamp")
    digest = open(stamp).read().strip() if os.path.exists(stamp) else None
    prev = out.get("_source", {}) if isinstance(out.get("_source"), dict) else {}
    digests = sorted(set(([digest] if digest else []) + [d for d in prev.get("lib_digests", []) if d]))
    out["_source"] = {"passes": sorted(set(sources) | set(prev.get("passes", []))), "tool": "tools/pmc_step.py",
                      "lib_digest": digest, "lib_digests": digests,
                      "note": "lib_digest = sha256 of the kernel sources + flags (csrc/build.py) of the library the newest passes profiled; "
                              "lib_digests = every digest that contributed groups to this table (more than one = mixed trees)"}
    out.move_to_end("_calibration")
    out.move_to_end("_source")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", os.path.relpath(path, ROOT), "-", len([k for k in out if not k.startswith("_")]), "groups")
    for k, v in out.items():
        if not k.startswith("_") and k in groups:
            print("  %-46s %-44s %7.1f us  hbm %7.1f MB  busy %s  l2 %s" % (k[:46], v["symbol"][:44], v["launch_us_under_pmc"],
                                                                           v.get("hbm_bytes_per_launch", 0) / 1e6, v.get("mfma_busy_frac"), v.get("l2_hit_rate")))


if __name__ == "__main__":
    main()
