#!/bin/bash
set -u
O=gpurun_out/r2i
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in srgan cyclegan esrgan; do
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$w -o $w -- python $R/bench.py --workload $w --steps 3 --warmup 1 --min-seconds 0 --no-roofline --no-cpu-baseline > $R/$O/prof_$w.log 2>&1)
python tools/rocpd_stats.py $O/prof_$w/*.db 30 > $O/stats_$w.txt
head -34 $O/stats_$w.txt
done
