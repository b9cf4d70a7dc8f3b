#!/usr/bin/env python
"""Turn the raw output of tools/final_measure.sh (gpurun_out/final/) into the committed evidence under profiles/.

    python tools/collect_profiles.py [--pmc-only]

profiles/r01_pmc_traffic.json is keyed by bench.py's roofline group names; HBM bytes per launch = 2 x FETCH_SIZE
(gfx950 reports 64 B per 128 B request, calibrated below on the ATen streaming kernels of the same pass) + WRITE_SIZE."""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
KIB = 1024


def load(counter):
    rows = [r for r in csv.DictReader(open(os.path.join(F, "pmc_%s" % counter, "mb_counter_collection.csv")))
            if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = collections.defaultdict(list)
    for r in rows:
        nm = r["Kernel_Name"]
        nm = nm[:nm.index("(")] if "(" in nm else nm
        out[nm.replace("void ", "")].append(float(r["Counter_Value"]))
    return out


def mean_tail(v, n=5):
    v = v[-n:]
    return sum(v) / len(v)


def pmc_table():
    fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
    calib = [k for k in fe if "vectorized_elementwise" in k and "Mul" in k]
    cal = {"fetch_kb": fe[calib[0]], "write_kb": wr[calib[0]]} if calib else {}
    x, dy = 128 * 32 * 32 * 128 * 4, 128 * 64 * 64 * 64 * 4
    src = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace) on tools/conv_microbench.py "
           "--shapes dcgan --match G.conv2 (tools/final_measure.sh); bytes = 2 x FETCH_SIZE (gfx950 note of "
           "MI355X_MICROARCH.md) + WRITE_SIZE; calibration in the same pass, ATen `x*2-1` on 67.1/134.2 MB tensors: "
           "FETCH_SIZE %s KB, WRITE_SIZE %s KB" % ([round(v) for v in cal.get("fetch_kb", [])],
                                                      [round(v) for v in cal.get("write_kb", [])]))
    # the microbench dispatches fwd, dgrad, wgrad, ufwd, udgrad, uwgrad (3 warm-up + 5 timed each): the collapsed
    # launches are the LAST 8 dispatches of a symbol shared with the dense ones
    groups = {
        "upconv_wgrad_64x128[128->64@64]": ("wgrad_inc_kernel<64, 128, true, false>", x + dy + 4 * 64 * 64 * 512 * 4,
                                            "DCGAN G.conv2 weight gradient (phase-collapsed Upsample+Conv3x3 128->64 @64x64, bs 128): "
                                            "reads x 67.1 MB + dy 134.2 MB, writes 4x64 split-K slabs 33.6 MB"),
        "upconv_fwd_igemm_1128064[128->64@64]": ("igemm_pipe_kernel<128, 64, 2, 2, false, true>", x + dy,
                                                 "DCGAN G.conv2 forward (phase-collapsed, 4 classes x 4 taps, tap-inner K order): reads x "
                                                 "67.1 MB, writes y 134.2 MB (tap-outer order: 921 MB fetched)"),
        "upconv_dgrad_igemm_1128128[128->64@64]": ("igemm_pipe_kernel<128, 128, 2, 2, false, false>", x + dy,
                                                   "DCGAN G.conv2 input gradient (phase-collapsed, 16 taps, source stride 2): reads dy "
                                                   "134.2 MB, writes dx 67.1 MB"),
    }
    tab = {}
    for name, (sym, alg, what) in groups.items():
        if sym not in fe:
            print("pmc: no dispatches of", sym, file=sys.stderr)
            continue
        fk, wk = mean_tail(fe[sym]), mean_tail(wr[sym])
        b = int((2 * fk + wk) * KIB)
        tab[name] = {"symbol": sym, "hbm_bytes_per_launch": b, "fetch_size_kb_reported": round(fk, 1),
                     "write_size_kb_reported": round(wk, 1), "algorithmic_bytes_per_launch": alg,
                     "ratio": round(b / alg, 2), "workload": what, "source": src}
    json.dump(tab, open(os.path.join(P, "r01_pmc_traffic.json"), "w"), indent=1)
    for k, v in tab.items():
        print("%-44s %7.1f MB  x%.2f of algorithmic" % (k, v["hbm_bytes_per_launch"] / 1e6, v["ratio"]))


def main():
    pmc_table()
    if "--pmc-only" in sys.argv:
        return
    shutil.copy(os.path.join(F, "bench.json"), os.path.join(P, "r01_final_bench.json"))
    shutil.copy(os.path.join(F, "mfma_loop_probe.txt"), os.path.join(P, "r01_mfma_loop_probe.txt"))
    stats = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"),
                            os.path.join(F, "prof", "dcgan_results.db"), "33"], capture_output=True, text=True).stdout
    open(os.path.join(P, "r01_final_dcgan_kernel_stats.txt"), "w").write(
        "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline  (tools/final_measure.sh)\n" + stats)
    open(os.path.join(P, "r01_conv_microbench.txt"), "w").write(
        "# python tools/conv_microbench.py --shapes dcgan|cyclegan|srgan --iters 10   (one MI355X, final round-1 build; "
        "u* = phase-collapsed Upsample+Conv3x3, TF = algorithmic dense FLOPs / time)\n" + open(os.path.join(F, "conv_microbench.txt")).read())
    open(os.path.join(P, "r01_models.txt"), "w").write(
        "# python tools/bench_models.py --workload <w>   (one MI355X, final round-1 build; hipgraph: whole step replayed as "
        "captured graph(s))\n" + open(os.path.join(F, "models.txt")).read())
    for f in ("pytest_gpu.log", "smoke.log"):
        print(open(os.path.join(F, f)).read().strip().splitlines()[-1])


if __name__ == "__main__":
    main()
