#!/usr/bin/env python
"""Stand-alone timing of the normalisation launchers (C ABI called directly, HIP events on the launch stream).

    python tools/norm_microbench.py [--shapes srgan,dcgan,cyclegan] [--iters 20] [--repeat 3] [--rotate 8]

One line per (tensor, pass): time and the algorithmic bytes it moves per second.  --rotate R walks R copies of the tensors (R x bytes above
the 256 MB Infinity Cache = every pass finds its operands in HBM); --rotate 1 = operands still in the caches, as behind the producing conv.
Used for tuning the streaming passes (profiles/r06_ab.txt); test infrastructure only.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

# name, G (1 = BatchNorm, N = InstanceNorm), pixels per group, channels, PReLU behind it
SHAPES = {
    "srgan": [("trunk BN 16x96x96x64 +PReLU", 1, 16 * 96 * 96, 64, True), ("trunk BN 16x96x96x64", 1, 16 * 96 * 96, 64, False),
              ("D BN 16x192x192x128", 1, 16 * 192 * 192, 128, False)],
    "dcgan": [("G BN 128x64x64x64", 1, 128 * 64 * 64, 64, False), ("G BN 128x32x32x128", 1, 128 * 32 * 32, 128, False)],
    "cyclegan": [("R256 IN 8 x 64x64x256", 8, 64 * 64, 256, False), ("R256 IN 1 x 64x64x256", 1 * 1, 64 * 64, 256, False),
                 ("u64 IN 8 x 256x256x64", 8, 256 * 256, 64, False)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="srgan,dcgan,cyclegan")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--rotate", type=int, default=1)
    args = ap.parse_args()
    import importlib

    F = importlib.import_module("pytorch_gan_amd.functional")
    lib, check = F.lib, F.check
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for fam in args.shapes.split(","):
        for name, G, P, C, prelu in SHAPES[fam]:
            R = args.rotate
            x = [torch.randn(G * P * C, device=dev) for _ in range(R)]
            dy = [torch.randn(G * P * C, device=dev) for _ in range(R)]
            y = [torch.empty(G * P * C, device=dev) for _ in range(R)]
            mean, invstd = torch.zeros(G * C, device=dev), torch.ones(G * C, device=dev)
            gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            dg, db, dp = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(1, device=dev)
            pw = torch.full((1,), 0.25, device=dev)
            nb = lib.migan_norm_workspace(G, P, C)
            ws = torch.empty(nb // 4 + 1, device=dev)
            nbp = lib.migan_norm_workspace_prelu(G, P, C)
            wsp = torch.empty(nbp // 4 + 1, device=dev)
            aff = G == 1
            gp, bp = (gamma.data_ptr(), beta.data_ptr()) if aff else (None, None)
            i = [0]

            def nxt():
                i[0] = (i[0] + 1) % R
                return i[0]

            calls = {}
            calls["stats"] = (1, lambda: lib.migan_norm_stats(x[nxt()].data_ptr(), mean.data_ptr(), invstd.data_ptr(), None, None, None, 0.1, 1e-5,
                                                              G, P, C, ws.data_ptr(), nb, st))
            if prelu:
                def apply():
                    k = nxt()
                    return lib.migan_norm_apply_prelu(x[k].data_ptr(), y[k].data_ptr(), mean.data_ptr(), invstd.data_ptr(), gp, bp, None,
                                                      pw.data_ptr(), G, P, C, 0, 0, st)

                def bwd():
                    k = nxt()
                    return lib.migan_norm_bwd_prelu(x[k].data_ptr(), dy[k].data_ptr(), mean.data_ptr(), invstd.data_ptr(), gp, bp, pw.data_ptr(),
                                                    y[k].data_ptr(), dg.data_ptr(), db.data_ptr(), dp.data_ptr(), G, P, C, wsp.data_ptr(), nbp,
                                                    0, 0, None, 0, 0, st)
            else:
                def apply():
                    k = nxt()
                    return lib.migan_norm_apply(x[k].data_ptr(), y[k].data_ptr(), mean.data_ptr(), invstd.data_ptr(), gp, bp, None, G, P, C,
                                                1, 0.2, st)

                def bwd():
                    k = nxt()
                    return lib.migan_norm_bwd(x[k].data_ptr(), dy[k].data_ptr(), mean.data_ptr(), invstd.data_ptr(), gp, bp, y[k].data_ptr(),
                                              dg.data_ptr() if aff else None, db.data_ptr() if aff else None, G, P, C, 1, 0.2, ws.data_ptr(),
                                              nb, 0, None, st)
            calls["apply"] = (2, apply)
            calls["bwd"] = (5, bwd)   # sums pass reads x, dy; apply pass reads x, dy and writes dx
            for d, (passes, fn) in calls.items():
                for _ in range(3):
                    check(fn(), d)
                torch.cuda.synchronize()
                times = []
                for _rep in range(max(1, args.repeat)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    times.append(e0.elapsed_time(e1) / args.iters)
                times.sort()
                nbytes = passes * G * P * C * 4
                print("%-30s %-6s %8.1f us  %6.2f TB/s  (%.1f MB, median %.1f us)" % (name, d, times[0] * 1e3, nbytes / (times[0] * 1e-3) / 1e12,
                                                                                     nbytes / 1e6, times[len(times) // 2] * 1e3), flush=True)


if __name__ == "__main__":
    main()
