#!/usr/bin/env python
"""Idle time between kernels of a recorded step in a rocprofv3 --kernel-trace CSV:
    python tools/trace_gaps.py <kernel_trace.csv> <launches per step> [steps to analyse, default 10]
Takes the last steps*launches kernels of the trace (replays), cuts them into windows of one step's launch count, and prints per window the
span, the union of the kernels' busy intervals over all streams and the idle remainder; then the gaps that recur (same kernel before and
after) with their median length - what a step loses to node hand-over and to waits on another stream."""
import collections
import csv
import statistics
import sys


def main():
    path, per = sys.argv[1], int(sys.argv[2])
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    rows = list(csv.DictReader(open(path)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda k: k[0])
    ks = ks[-per * nsteps:]
    recur = collections.defaultdict(list)
    spans, idles = [], []
    for w in range(nsteps):
        win = ks[w * per:(w + 1) * per]
        cur_s, cur_e, busy, last = win[0][0], win[0][1], 0, win[0]
        for s, e, n in win[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                recur[(last[2][:48], n[:48])].append(s - cur_e)
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
            if e >= cur_e:
                last = (s, e, n)
        busy += cur_e - cur_s
        span = win[-1][1] - win[0][0] if w == nsteps - 1 else ks[(w + 1) * per][0] - win[0][0]
        spans.append(span)
        idles.append(span - busy)
    print("per step: span median %.1f us, idle median %.1f us (%.1f %%), min idle %.1f us" % (
        statistics.median(spans) / 1e3, statistics.median(idles) / 1e3, 100.0 * statistics.median(idles) / statistics.median(spans),
        min(idles) / 1e3))
    tot = sorted(((statistics.median(v) * len(v) / nsteps, statistics.median(v), len(v) / nsteps, k) for k, v in recur.items()), reverse=True)
    print("recurring gaps: us per step | median us | per step | after -> before")
    for t, m, c, (a, b) in tot[:25]:
        print("  %7.1f  %6.1f  %4.1f  %s -> %s" % (t / 1e3, m / 1e3, c, a, b))
    small = [g for v in recur.values() for g in v if g <= 10000]
    print("gaps <= 10 us: %.1f us per step over %.1f gaps per step" % (sum(small) / 1e3 / nsteps, len(small) / nsteps))


if __name__ == "__main__":
    main()
