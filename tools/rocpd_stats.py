#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd database (`*_results.db`): the same table
`--stats` prints, for runs whose CSV output was not requested.
usage: rocpd_stats.py <db> [top_n] [divide_by_steps] [--by-grid] [--per-step SYMBOL=COUNT]
--per-step adam_kernel=3 counts the steps IN THE TRACE (launches of a kernel that runs COUNT times per training step, e.g. the
three optimisers of a CycleGAN step) instead of trusting divide_by_steps: a trace of bench.py also holds the untimed warm-up
steps (50+ for the CycleGAN replay buffers), so dividing by --steps alone overstates the per-step header.
--by-grid splits a symbol by launch grid (threads x,y,z): one row per layer shape, which is what bench.py's roofline groups
time (e.g. wgrad_inc_kernel<64,128,true> serves both DCGAN up-conv layers)."""
import sqlite3
import sys


def main():
    by_grid = "--by-grid" in sys.argv
    argv = [a for a in sys.argv if a != "--by-grid"]
    per_step = None
    if "--per-step" in argv:
        i = argv.index("--per-step")
        sym, cnt = argv[i + 1].rsplit("=", 1)
        per_step = (sym, float(cnt))
        del argv[i:i + 2]
    db = sqlite3.connect(argv[1])
    top = int(argv[2]) if len(argv) > 2 else 40
    steps = float(argv[3]) if len(argv) > 3 else 0
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else cols[0]
    if by_grid and "grid_x" in cols:
        key = "%s || ' grid=' || grid_x || 'x' || grid_y || 'x' || grid_z" % name
    else:
        key = name
    rows = db.execute("select %s, count(*), sum(end - start), avg(end - start) from kernels group by 1 order by 3 desc" % key).fetchall()
    tot = sum(r[2] for r in rows)
    how = ""
    if per_step:
        calls = sum(r[1] for r in rows if per_step[0] in r[0])
        if calls == 0:
            raise SystemExit("--per-step: no kernel matches %r" % per_step[0])
        steps = calls / per_step[1]
        how = " (%g steps in the trace: %d launches of %s / %g per step, warm-up included)" % (steps, calls, per_step[0], per_step[1])
    print("total kernel time %.3f ms over %d launches%s" % (tot / 1e6, sum(r[1] for r in rows),
                                                          ((" = %.1f us, %.1f launches per step" % (tot / 1e3 / steps, sum(r[1] for r in rows) / steps)) if steps else "") + how))
    print("%-110s %7s %10s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "share"))
    for n, c, t, a in rows[:top]:
        if by_grid and " grid=" in n:  # keep the grid visible when the symbol is long
            sym, grid = n.rsplit(" grid=", 1)
            n = sym[:110 - len(grid) - 6] + " grid=" + grid
        print("%-110s %7d %10.1f %9.2f %5.1f%%" % (n[:110], c, t / 1e3, a / 1e3, 100.0 * t / tot))


if __name__ == "__main__":
    main()
