#!/bin/bash
# per-kernel times of whole steps (eager launches), rocprofv3 --kernel-trace --stats -> gpurun_out/$1/
O=gpurun_out/${1:-r3prof}
shift
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in "$@"; do
  k=3; [ $w = dcgan ] && k=20; [ $w = pix2pix ] && k=20; [ $w = wgan_gp ] && k=50
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$w -o $w -- python $R/bench.py --workload $w --steps $k --warmup 2 \
     --min-seconds 0 --no-graph --no-roofline --no-cpu-baseline --no-extra > $R/$O/prof_$w.log 2>&1)
  db=$(ls $O/prof_$w/*/${w}_results.db $O/prof_$w/${w}_results.db 2>/dev/null | head -1)
  python tools/rocpd_stats.py $db 150 $((k+2)) --by-grid > $O/${w}_kernel_stats.txt 2>&1
  head -3 $O/${w}_kernel_stats.txt
done
