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


def stats(db, top=40, steps=None, by_grid=False):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "rocpd_stats.py"), db, str(top)] + ([str(steps)] if steps else []) \
        + (["--by-grid"] if by_grid else [])
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
        ("128x64 tile at 5 workgroups/CU vs 4 (gpu_run8; MIGAN_IGEMM_OCC5)", "r2h/mb_occ5.txt"),
        ("XCD-contiguous tile order vs plain (gpu_run12; MIGAN_IGEMM_XCD)", "r2l/mb_xcd.txt"),
        ("s_setprio around the MFMA stream vs none (gpu_run19; MIGAN_MFMA_PRIO)", "r2r/mb_prio.txt"),
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
        ("gpu_run7 pix2pix: mask plan (box check after a slow-box reading)", "r2g/pix2pix_ab.txt"),
        ("gpu_run7 esrgan first line", "r2g/esrgan.txt"),
        ("gpu_run8: igemm 128x64 tile at 5 workgroups/CU", "r2h/occ5_ab.txt"),
        ("gpu_run10: pack plan, every weight planned", "r2j/packs_ab.txt"),
        ("gpu_run11: pack plan with the 1M-element cap (default), off, and uncapped", "r2k/packs_ab.txt"),
        ("gpu_run12: XCD-contiguous tile order of the forward/dgrad implicit GEMM (rejected, opt-in)", "r2l/xcd_ab.txt"),
        ("gpu_run13: BatchNorm+PReLU fusion (srgan lines: fused, then MIGAN_NO_PRELU_FUSE=1; twice)", "r2m/prelu_ab.txt"),
        ("gpu_run16: dcgan after the 1024-element pack chunks (slow box)", "r2p/dcgan.txt"),
        ("gpu_run17: PixelShuffle as the norm launches' index map (srgan, fused / MIGAN_NO_SHUFFLE_FUSE=1, twice)", "r2q/shuffle_ab.txt"),
        ("gpu_run19: s_setprio 1 around the MFMA stream of igemm_pipe (MIGAN_MFMA_PRIO), first box", "r2r/prio_ab.txt"),
        ("gpu_run20: the same, second box, order reversed and repeated", "r2s/prio_ab2.txt"),
        ("gpu_run21: reflect-1 ring correction on a side stream under the wgrad (cyclegan, on / off, twice)", "r2t/ring_ab.txt"),
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
        steps = {"dcgan": 25, "cyclegan": 5, "srgan": 5}.get(tag)   # timed + warm-up steps of tools/round_measure.sh
        head = ("# rocprofv3 --kernel-trace --stats -- python bench.py --workload %s --steps K --warmup 2 --no-graph (tools/round_measure.sh,\n"
                "# final build; eager launches so that every kernel is a dispatch), summarised by tools/rocpd_stats.py --by-grid: one row per\n"
                "# (symbol, launch grid) = per layer shape.  [gpurun_out/final/%s_results.db]\n" % (tag, tag))
        open(os.path.join(P, "r02_final_%s_kernel_stats.txt" % tag), "w").write(head + stats(f, 60, steps, True))
        print("wrote r02_final_%s_kernel_stats.txt" % tag)
    b = os.path.join(G, "final", "bench.json")
    if os.path.exists(b):
        import json

        line = open(b).readline()
        json.dump(json.loads(line), open(os.path.join(P, "r02_final_bench.json"), "w"), indent=1)
        print("wrote r02_final_bench.json")
    o = os.path.join(G, "final", "bench_others.jsonl")
    if os.path.exists(o):
        import json

        json.dump([json.loads(l) for l in open(o) if l.strip()], open(os.path.join(P, "r02_final_bench_others.json"), "w"), indent=1)
        print("wrote r02_final_bench_others.json")
    t = os.path.join(G, "final", "pytest_gpu.log")
    if os.path.exists(t):
        open(os.path.join(P, "r02_final_pytest_gpu.txt"), "w").write("# python -m pytest tests -m gpu -q on the GPU box (tools/round_measure.sh)\n"
                                                                      + "".join(open(t).readlines()[-6:]))
        print("wrote r02_final_pytest_gpu.txt")

if __name__ == "__main__":
    main()
