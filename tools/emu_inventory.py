#!/usr/bin/env python3
"""Launch inventory of one training step on the HOST execution model of the kernels (tests/hipemu - no GPU needed): which
kernels a `bench.py` workload launches per step, how often, and with how many workgroups.  This is where the launch counts in
DESIGN.md come from (pix2pix 337 -> 249, wgan_gp ~4 per iteration, dcgan 129 at batch 4).  Counts are exact; times are not
modelled.

    python tools/emu_inventory.py pix2pix            # full BASELINE shape where the model can afford it (pix2pix: ~40 s per step)
    python tools/emu_inventory.py wgan_gp --steps 6
    python tools/emu_inventory.py dcgan --batch 4     # big convs are slow on the model: shrink the batch, the launch list is the same
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--steps", type=int, default=1, help="steps counted (after the builder's own warm-up and one more step)")
    ap.add_argument("--max-wg", type=int, default=0, help="list only launches of fewer than this many workgroups")
    args = ap.parse_args()
    if os.environ.get("_EMU_INVENTORY_CHILD") != "1":   # the counters print from C: run as a child and post-process its output
        env = dict(os.environ, _EMU_INVENTORY_CHILD="1")
        out = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
        sys.stderr.write(out.stderr[-3000:] if out.returncode else "")
        rows, agg = [], collections.Counter()
        for ln in out.stdout.splitlines():
            m = re.match(r"\s*(\d+)\s+(.*) wg=(\d+) x(\d+)", ln)
            if m:
                rows.append((int(m.group(3)), int(m.group(1)), m.group(2), int(m.group(4))))
                agg[re.sub(r"<.*", "", m.group(2)).strip("( ")] += int(m.group(1))
        total = sum(r[1] for r in rows)
        print("%d launches in %d step(s) = %.1f per step" % (total, args.steps, total / max(args.steps, 1)))
        for k, v in agg.most_common():
            print("%6d  %s" % (v, k))
        if args.max_wg:
            print("-- launches of fewer than %d workgroups" % args.max_wg)
            for wg, n, k, th in sorted(rows):
                if wg < args.max_wg:
                    print("%5d wg x%4d  %3d calls  %s" % (wg, th, n, k[:100]))
        return out.returncode
    import torch

    import bench
    import hipemu
    import hipemu.host
    from pytorch_gan_amd.dp import LocalStepper

    a = types.SimpleNamespace(batch=args.batch, no_graph=True, global_batch=0, sync_bn=False)
    with hipemu.host.emulated_device() as emu:
        w = bench.BUILDERS[args.workload](LocalStepper(), 0, torch.device("cpu"), a, args.steps + 2)
        w.run(0)
        emu.hipemu_count_grids(1)
        emu.hipemu_reset_counts()
        for i in range(1, args.steps + 1):
            w.run(i)
        emu.hipemu_print_counts()
    return 0


if __name__ == "__main__":
    sys.exit(main())
