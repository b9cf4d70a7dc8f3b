#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd database (`*_results.db`): the same table
`--stats` prints, for runs whose CSV output was not requested.  usage: rocpd_stats.py <db> [top_n] [divide_by_steps]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 0
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else cols[0]
    rows = db.execute("select %s, count(*), sum(end - start), avg(end - start) from kernels group by %s order by 3 desc" % (name, name)).fetchall()
    tot = sum(r[2] for r in rows)
    print("total kernel time %.3f ms over %d launches%s" % (tot / 1e6, sum(r[1] for r in rows),
                                                          (" = %.1f us, %.1f launches per step" % (tot / 1e3 / steps, sum(r[1] for r in rows) / steps)) if steps else ""))
    print("%-110s %7s %10s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "share"))
    for n, c, t, a in rows[:top]:
        print("%-110s %7d %10.1f %9.2f %5.1f%%" % (n[:110], c, t / 1e3, a / 1e3, 100.0 * t / tot))


if __name__ == "__main__":
    main()
