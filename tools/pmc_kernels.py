#!/usr/bin/env python
"""rocprofv3 --pmc passes over tools/conv_microbench.py  ->  profiles/<round>_pmc_kernels.json (what bench.py's `roofline` object
reads: HBM bytes per launch and MFMA-busy fraction of the conv-family launches, keyed by bench.py's roofline group names).

    python tools/pmc_kernels.py <round> <pass_dir> [<pass_dir> ...]

Each <pass_dir> holds ONE rocprofv3 invocation (`--pmc <counters> --kernel-trace --output-format csv`, never combined with
sys/hip traces) over `conv_microbench.py --pmc-log <pass_dir>/segments.jsonl ...`.  The microbench launches a 1-element
`axpby_kernel` marker in front of every timed (layer, direction) and logs the segment, so the dispatch list of a pass is
cut at the markers: the counters of all non-ATen kernels of a segment, divided by the number of calls (3 warm-up + iters),
are the per-launch figures of that direction (a wgrad launch = main kernel + split-K reduction, like bench.py counts it).

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are reported in KiB;
FETCH_SIZE counts 64 B per 128 B request on gfx950, so HBM read bytes = 2 x FETCH_SIZE x 1024 (checked in every pass on the
ATen fill / uniform kernels of the microbench's own setup, recorded under "_calibration"); WRITE_SIZE x 1024 as is.
SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs:
mfma_busy_frac = BUSY / (128 x GUI_ACTIVE)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(nm):
    return re.sub(r"\(.*", "", nm).replace("void ", "")[:120]


def group_name(seg):
    """bench.py ConvProfiler group of a microbench segment (same strings)."""
    N, Ci, Co, H, Ho, k, s, d = seg["N"], seg["Ci"], seg["Co"], seg["H"], seg["Ho"], seg["k"], seg["stride"], seg["dir"]
    shape = "[%dx %d->%d k%d%s @%d]" % (N, Ci, Co, k, "s2" if s == 2 else "", Ho)
    if d in ("ufwd", "udgrad", "uwgrad"):
        return "upconv_%s[%dx %d->%d @%d]" % (d[1:], N, Ci, Co, 2 * H)
    if d == "wgrad":
        return "wgrad" + shape
    if d == "rdgrad":
        return "dgrad_reflect1" + shape
    if d in ("tfwd", "twgrad", "tdgrad", "texpand"):
        return "toeplitz_%s%s" % (d[1:], shape)
    return None  # fwd / dgrad carry the tile code, resolved by the caller through libmigan


def main():
    rnd, dirs = sys.argv[1], sys.argv[2:]
    try:
        sys.path.insert(0, ROOT)
        from importlib import import_module

        lib = import_module("pytorch_gan_amd._lib").lib
    except Exception:  # noqa: BLE001 - naming only
        lib = None
    groups = collections.OrderedDict()
    calib = []
    for d in dirs:
        segs = [json.loads(l) for l in open(os.path.join(d, "segments.jsonl"))]
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        rows = [r for f in files for r in csv.DictReader(open(f))]
        by_disp = collections.OrderedDict()
        for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
            e = by_disp.setdefault(int(r["Dispatch_Id"]), {"name": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"]), "ctr": {},
                                                           "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            e["ctr"][r["Counter_Name"]] = float(r["Counter_Value"])
        cur = -1
        for e in by_disp.values():
            if e["name"].startswith("axpby_kernel") and e["grid"] <= 256:
                cur += 1
                continue
            aten = e["name"].startswith("at::native") or "rocclr" in e["name"]
            if aten:
                if cur < 0 and ("FETCH_SIZE" in e["ctr"] or "WRITE_SIZE" in e["ctr"]) and e["grid"] >= 1 << 20:
                    calib.append({"kernel": e["name"][:60], "grid_threads": e["grid"], **{k: v for k, v in e["ctr"].items()}})
                continue
            if cur < 0 or cur >= len(segs) or e["name"].startswith("upconv_pack") or e["name"].startswith("toep_pack"):
                continue
            seg = segs[cur]
            name = group_name(seg)
            if name is None and lib is not None:
                N, Ci, Co, Ho, k, s = seg["N"], seg["Ci"], seg["Co"], seg["Ho"], seg["k"], seg["stride"]
                shape = "[%dx %d->%d k%d%s @%d]" % (N, Ci, Co, k, "s2" if s == 2 else "", Ho)
                if seg["dir"] == "fwd":
                    name = "fwd_igemm_%d%s" % (lib.migan_igemm_tile_code(N * Ho * Ho, Co, Ci, 1), shape)
                else:
                    Hd = seg["H"] + (2 * (k // 2) if seg["gather"] == 1 else 0)
                    maxm = N * ((Hd + s - 1) // s) ** 2
                    name = "dgrad_igemm_%d%s" % (lib.migan_igemm_tile_code(maxm, Ci, Co, s * s), shape)
            name = name or "%s %s" % (seg["layer"], seg["dir"])
            g = groups.setdefault(name, {"layer": seg["layer"], "direction": seg["dir"], "calls": seg["calls"],
                                         "dense_gflop_per_launch": round(seg["dense_flops"] / 1e9, 3),
                                         "executed_gflop_per_launch": round(seg["executed_flops"] / 1e9, 3), "_k": {}, "_c": {},
                                         "_kdir": d, "_cdir": {}})
            if g["_kdir"] == d:  # kernel mix and durations from the first pass that covers the group
                k = g["_k"].setdefault(e["name"], {"n": 0, "ns": 0})
                k["n"] += 1
                k["ns"] += e["t"]
            for cn, cv in e["ctr"].items():
                if g["_cdir"].setdefault(cn, d) == d:  # a counter collected in several passes: keep the first
                    g["_c"][cn] = g["_c"].get(cn, 0.0) + cv
    out = collections.OrderedDict()
    for name, g in groups.items():
        calls = float(g["calls"])
        c = {k: v / calls for k, v in g["_c"].items()}
        main_k = max(g["_k"], key=lambda k: g["_k"][k]["ns"])
        ent = {"layer": g["layer"], "direction": g["direction"], "symbol": main_k,
               "kernels_per_launch": {k: round(v["n"] / calls, 2) for k, v in g["_k"].items()},
               "launch_us_under_pmc": round(sum(v["ns"] for v in g["_k"].values()) / calls / 1e3, 1),
               "dense_gflop_per_launch": g["dense_gflop_per_launch"], "executed_gflop_per_launch": g["executed_gflop_per_launch"]}
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
        if "SQ_INSTS_MFMA" in c:
            ent["mfma_instructions"] = round(c["SQ_INSTS_MFMA"])
        out[name] = ent
    out["_calibration"] = {"note": "ATen setup kernels of the same passes (torch.rand / mul / fill over >= 1M threads): bytes moved are "
                                   "known (4 B per element), FETCH_SIZE KiB x 2 and WRITE_SIZE KiB x 1 reproduce them",
                           "samples": calib[:12]}
    out["_source"] = {"passes": [os.path.relpath(os.path.abspath(d), ROOT) for d in dirs],
                      "tool": "tools/pmc_kernels.py", "counters": sorted({k for g in groups.values() for k in g["_c"]})}
    path = os.path.join(ROOT, "profiles", "%s_pmc_kernels.json" % rnd)
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", os.path.relpath(path, ROOT), "-", len(out) - 2, "groups")
    for k, v in out.items():
        if not k.startswith("_"):
            print("  %-44s %-40s hbm %s MB  busy %s" % (k, v["symbol"][:40], round(v.get("hbm_bytes_per_launch", 0) / 1e6, 1),
                                                        v.get("mfma_busy_frac")))


if __name__ == "__main__":
    main()
