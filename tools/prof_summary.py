"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats): per-kernel totals per step and the launch
sequence of the last step.  usage: python tools/prof_summary.py <results.db> <steps_in_db> [--seq]"""
import re
import sqlite3
import sys


def short(nm):
    nm = re.sub(r"\(.*", "", nm).replace("void ", "").replace("at::native::", "")
    return re.sub(r"vectorized_elementwise_kernel<4, ", "vec<", nm)[:70]


def main():
    db, steps = sys.argv[1], int(sys.argv[2])
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
    n = len(rows) // steps
    last = rows[-n:]
    agg = {}
    for r in rows[-n * (steps - 3):]:  # skip the eager warm-up / capture passes at the front
        d = agg.setdefault(short(r[0]), [0, 0.0])
        d[0] += 1
        d[1] += (r[2] - r[1]) / 1000.0
    k = float(steps - 3)
    tot = sum(v[1] for v in agg.values()) / k
    print("# per step (mean of the last %d steps): %d launches, kernel time %.1f us" % (steps - 3, n, tot))
    print("# kernel | launches/step | us/step | share")
    for name, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %6.1f %9.1f us %5.1f%%" % (name, v[0] / k, v[1] / k, 100 * v[1] / k / tot))
    if "--seq" in sys.argv:
        t0 = last[0][1]
        for r in last:
            print("%9.1f %8.1f %s grid=(%d,%d,%d)" % ((r[1] - t0) / 1000, (r[2] - r[1]) / 1000, short(r[0]), r[3], r[4], r[5]))


if __name__ == "__main__":
    main()
