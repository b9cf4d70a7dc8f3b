#!/usr/bin/env python
"""Step timing of the other BASELINE.json configurations on one MI355X (bench.py holds the headline DCGAN run).

    python tools/bench_models.py --workload cyclegan|srgan|pix2pix|wgan_gp [--steps K] [--warmup W] [--batch B]

Prints one JSON line: images/s, ms/step, algorithmic TFLOP/s of the whole step vs the 157.3 TF fp32-MFMA peak
(FLOPs per image from SURVEY.md §8d).  Synthetic inputs resident in HBM, eager launches (the CycleGAN step has
host-side replay-buffer logic between its generator and discriminator phases)."""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK = 157.3e12
GFLOP_PER_IMG = {"cyclegan": 2097.99, "srgan": 541.43, "pix2pix": 65.52, "wgan_gp": 0.0219}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cyclegan")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--graph", type=int, default=-1, help="1: replay the step as a hipGraph, 0: eager launches; "
                    "default: graph for the launch-bound workloads (pix2pix, wgan_gp)")
    args = ap.parse_args()
    from pytorch_gan_amd import graph as gmod
    from pytorch_gan_amd import models, steps

    torch.manual_seed(0)
    np.random.seed(0)
    random.seed(0)
    dev = "cuda:0"
    w = args.workload
    if w == "cyclegan":
        B = args.batch or 8
        shape = (3, 256, 256)
        nets = [models.CycleGenerator(shape, 9), models.CycleGenerator(shape, 9), models.CycleDiscriminator(shape),
                models.CycleDiscriminator(shape)]
        for n in nets:
            n.apply(models.init_normal_cyclegan)
        s = steps.make_cyclegan_state(*[n.to(dev) for n in nets])
        a = (torch.rand(B, *shape) * 2 - 1).to(dev)
        b = (torch.rand(B, *shape) * 2 - 1).to(dev)
        fn = lambda i: steps.cyclegan_step(s, a, b)  # noqa: E731
    elif w == "srgan":
        B = args.batch or 16
        G, D, V = models.SrganGenerator(), models.SrganDiscriminator((3, 384, 384)), models.SrganFeatureExtractor()
        s = steps.make_srgan_state(G.to(dev), D.to(dev), V.to(dev))
        lr, hr = torch.randn(B, 3, 96, 96).to(dev), torch.randn(B, 3, 384, 384).to(dev)
        fn = lambda i: steps.srgan_step(s, lr, hr)  # noqa: E731
    elif w == "pix2pix":
        B = args.batch or 1
        G, D = models.Pix2pixGenerator(), models.Pix2pixDiscriminator()
        G.apply(models.init_normal_dcgan)
        D.apply(models.init_normal_dcgan)
        s = steps.make_pix2pix_state(G.to(dev), D.to(dev), 256)
        a = (torch.rand(B, 3, 256, 256) * 2 - 1).to(dev)
        b = (torch.rand(B, 3, 256, 256) * 2 - 1).to(dev)
        fn = lambda i: steps.pix2pix_step(s, a, b)  # noqa: E731
    elif w == "wgan_gp":
        B = args.batch or 64
        G, D = models.MlpGenerator((1, 32, 32), 100), models.MlpCritic((1, 32, 32))
        s = steps.make_wgan_gp_state(G.to(dev), D.to(dev))
        real = (torch.rand(B, 1, 32, 32) * 2 - 1).to(dev)
        zs = torch.randn(64, B, 100).to(dev)
        alphas = torch.rand(64, B, 1, 1, 1).to(dev)
        fn = lambda i: steps.wgan_gp_step(s, real, i, zs[i % 64], alphas[i % 64])  # noqa: E731
    else:
        raise SystemExit("unknown workload " + w)
    use_graph = args.graph == 1 or (args.graph < 0 and w in ("pix2pix", "wgan_gp"))
    graphed, capture_error = False, None
    if use_graph:
        # static input buffers + captured step(s); WGAN-GP has two step shapes (critic only / critic + generator)
        if w == "wgan_gp":
            z_s, a_s = zs[0].clone(), alphas[0].clone()
            runners = {True: gmod.StepRunner(lambda: steps.wgan_gp_step(s, real, 0, z_s, a_s), s.dp),
                       False: gmod.StepRunner(lambda: steps.wgan_gp_step(s, real, 1, z_s, a_s), s.dp)}
            for r in runners.values():
                r.prepare()

            def fn(i):  # noqa: F811
                z_s.copy_(zs[i % 64])
                a_s.copy_(alphas[i % 64])
                return runners[i % s.n_critic == 0].run()
            graphed = all(r.graphed for r in runners.values())
            capture_error = next((r.capture_error for r in runners.values() if r.capture_error), None)
        else:
            eager_fn = fn
            runner = gmod.StepRunner(lambda: eager_fn(0), s.dp).prepare()
            fn = lambda i: runner.run()  # noqa: E731
            graphed, capture_error = runner.graphed, runner.capture_error
    for i in range(args.warmup):
        out = fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = fn(args.warmup + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    losses = {k: float(v) for k, v in out.items() if "loss" in k}
    assert all(np.isfinite(v) for v in losses.values()), losses
    ips = B * args.steps / el
    print(json.dumps({"workload": w, "batch": B, "images_per_s": round(ips, 3), "ms_per_step": round(1e3 * el / args.steps, 3),
                      "step_tflops": round(ips * GFLOP_PER_IMG[w] * 1e9 / 1e12, 2),
                      "step_mfma_frac": round(ips * GFLOP_PER_IMG[w] * 1e9 / PEAK, 4), "steps": args.steps,
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "losses": losses,
                      "hipgraph": graphed, **({"hipgraph_error": capture_error[:160]} if capture_error else {})}))


if __name__ == "__main__":
    main()
