#!/usr/bin/env python3
"""Timeline of the last launches in a rocprofv3 rocpd database (`*_results.db`): one CSV line per kernel - start and end relative to
the first dumped launch (us), queue / stream columns as the database has them, grid, symbol - for working out what ran beside what
(the two-stream step bodies: which chain is the critical path, where the chip idles).
usage: rocpd_timeline.py <db> [last_n_launches]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    print("# columns of `kernels`:", ",".join(cols))
    want = [c for c in ("start", "end", "queue_id", "stream_id", "tid", "grid_x", "grid_y", "grid_z", "workgroup_x", "name") if c in cols]
    rows = db.execute("select %s from kernels order by start desc limit %d" % (",".join(want), n)).fetchall()[::-1]
    t0 = rows[0][0]
    print(",".join(want))
    for r in rows:
        d = dict(zip(want, r))
        d["start"] = "%.2f" % ((d["start"] - t0) / 1e3)
        d["end"] = "%.2f" % ((d["end"] - t0) / 1e3)
        d["name"] = str(d["name"]).split("(")[0].replace("void ", "")[:70]
        print(",".join(str(d[c]) for c in want))


if __name__ == "__main__":
    main()
