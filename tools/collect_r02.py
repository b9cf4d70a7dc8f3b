#!/usr/bin/env python
"""Assemble the committed round-2 evidence under profiles/ from the raw gpurun output (gpurun_out/r2*/, scratch).

    python tools/collect_r02.py

Every output file starts with the gpurun call(s) it came from (tools/gpu_run<N>.sh wrote gpurun_out/r2<letter>/)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def read(rel):
    p = os.path.join(G, rel)
    if not os.path.exists(p):
        return None
    return "".join(l for l in open(p) if "amdgpu.ids" not in l)


def write(name, parts):
    body = []
    for title, rel in parts:
        txt = rel if title is None else read(rel)
        if txt is None:
            continue
        if title is not None:
            body.append("## %s   [gpurun_out/%s]\n" % (title, rel))
        body.append(txt.rstrip("\n") + "\n\n")
    if body:
        open(os.path.join(P, name), "w").write("".join(body))
        print("wrote", name)


def stats(db, top=40, steps=None):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "rocpd_stats.py"), db, str(top)] + ([str(steps)] if steps else [])
    return subprocess.run(cmd, capture_output=True, text=True).stdout


def main():
    write("r02_conv_microbench.txt", [
        (None, "# tools/conv_microbench.py on MI355X: per-layer launch time (HIP events), ALGORITHMIC (dense) TFLOP/s and its share of the\n"
               "# 157.3 TF fp32 MFMA peak.  u* = phase-collapsed up-conv (executes 4/9 of the dense FLOPs), r* = reflect-1 dgrad,\n"
               "# t* = width-Toeplitz thin-N path (texpand is shared by twgrad and tdgrad).\n"),
        ("forward / input gradient, all BASELINE layer shapes (gpu_run3)", "r2c/mb_fwd_dgrad.txt"),
        ("weight gradient, planner 3 = balance model, default (gpu_run3)", "r2c/mb_wgrad_plan3.txt"),
        ("weight gradient, planner 1 = round-1 rule, for comparison (gpu_run3)", "r2c/mb_wgrad_plan1.txt"),
        ("thin-N image-output convs: direct VALU kernels vs width-Toeplitz expansion (gpu_run6)", "r2f/mb_thin.txt"),
        ("thin-N after the BM=32 weight-gradient tile (final run)", "final/mb_thin.txt"),
        ("all shapes, final build (final run)", "final/conv_microbench.txt"),
    ])
    write("r02_ab.txt", [
        (None, "# Whole-step A/B runs of bench.py (columns: img/s, ms/step median, ms/step min[, hipgraph]).  Each block ran back to back on ONE\n"
               "# box; different blocks ran on different boxes (box-to-box spread of the same build: ~ +-10 %).\n"),
        ("gpu_run1: planner 1 vs 2, reflect1, colsum fusion, wgrad overlap", "r2a/dcgan_ab.txt"),
        ("gpu_run1 cyclegan", "r2a/cyclegan_ab.txt"),
        ("gpu_run2: V4 MFMA loop variants, wgrad OCC=4 (all rejected)", "r2b/dcgan_ab.txt"),
        ("gpu_run2 cyclegan", "r2b/cyclegan_ab.txt"),
        ("gpu_run3: planner 3 (base) vs planner 1; wgrad/dgrad side-stream overlap", "r2c/dcgan_ab.txt"),
        ("gpu_run3 cyclegan", "r2c/cyclegan_ab.txt"),
        ("gpu_run4: conv-epilogue norm statistics (base = on) vs off -> made opt-in", "r2d/dcgan_ab.txt"),
        ("gpu_run4 cyclegan", "r2d/cyclegan_ab.txt"),
        ("gpu_run4 wgan_gp: skinny GEMMs v1", "r2d/wgan_ab.txt"),
        ("gpu_run5 wgan_gp: skinny GEMMs v2 (K-slices) + skinny_tn", "r2e/wgan_ab.txt"),
        ("gpu_run5 dcgan: one dropout-mask launch per step", "r2e/dcgan_ab.txt"),
        ("gpu_run5 other workloads", "r2e/others.txt"),
        ("gpu_run6: width-Toeplitz thin-N convs", "r2f/toep_ab.txt"),
        ("gpu_run6 wgan_gp: skinny GEMMs v3 (one row group per workgroup)", "r2f/wgan.txt"),
        ("final run A/B", "final/ab.txt"),
    ])
    write("r02_tile_sweep.txt", [("MIGAN_IGEMM_TILE sweep over the layer shapes (gpu_run2)", "r2b/sweep_variants.txt")])
    write("r02_wgrad_split_sweep.txt", [("MIGAN_WGRAD_SPLITS sweep (gpu_run2)", "r2b/sweep_splits.txt")])
    write("r02_pmc_conv.txt", [
        (None, "# rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over tools/conv_microbench.py, summarised by\n"
               "# tools/pmc_summary.py.  FETCH_SIZE / WRITE_SIZE in KiB as reported (FETCH_SIZE counts 64 B per 128 B request on gfx950: x2).\n"),
        ("DCGAN G.conv2 (up-conv 128->64 @64x64, batch 128): fwd / dgrad / wgrad kernels (gpu_run2)", "r2b/pmc_G.conv2.txt"),
        ("CycleGAN R256 (reflect 256->256 @64x64, batch 8) (gpu_run2)", "r2b/pmc_R256.txt"),
        ("final build", "final/pmc_conv.txt"),
    ])
    for tag, rel, steps in (("dcgan", "r2d/prof_dcgan/dcgan_results.db", 28), ("wgan", "r2f/prof_wgan/wgan_results.db", None),
                            ("wgan_before_v3", "r2e/prof_wgan/wgan_results.db", None)):
        db = os.path.join(G, rel)
        if os.path.exists(db):
            head = "# rocprofv3 --kernel-trace --stats of bench.py (%s), summarised from the rocpd database by tools/rocpd_stats.py  [gpurun_out/%s]\n" % (tag, rel)
            open(os.path.join(P, "r02_%s_kernel_stats.txt" % tag), "w").write(head + stats(db, 45, steps))
            print("wrote r02_%s_kernel_stats.txt" % tag)
    for f in sorted(glob.glob(os.path.join(G, "final", "*_results.db"))):
        tag = os.path.basename(f).replace("_results.db", "")
        open(os.path.join(P, "r02_final_%s_kernel_stats.txt" % tag), "w").write(
            "# rocprofv3 --kernel-trace --stats of the final build's bench.py --workload %s [gpurun_out/final]\n" % tag + stats(f, 45))
        print("wrote r02_final_%s_kernel_stats.txt" % tag)


if __name__ == "__main__":
    main()
