#!/bin/bash
set -u
O=gpurun_out/r2d
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_fullsize_gpu.py > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
for v in "base" "MIGAN_CONV_STATS=0" "MIGAN_COLSUM_FUSE=0"; do
  echo "== dcgan $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'])"
done > $O/dcgan_ab.txt 2>&1
cat $O/dcgan_ab.txt
for v in "base" "MIGAN_SKINNY=0"; do
  echo "== wgan_gp $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --workload wgan_gp --steps 100 --warmup 10 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'], d['config']['hipgraph'])"
done > $O/wgan_ab.txt 2>&1
cat $O/wgan_ab.txt
for v in "base" "MIGAN_CONV_STATS=0"; do
  echo "== cyclegan $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --workload cyclegan --steps 3 --warmup 1 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'])"
done > $O/cyclegan_ab.txt 2>&1
cat $O/cyclegan_ab.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_wgan -o wgan -- python $R/bench.py --workload wgan_gp --steps 50 --warmup 5 --min-seconds 0 --no-roofline --no-graph > $R/$O/prof_wgan.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_dcgan -o dcgan -- python $R/bench.py --steps 20 --warmup 5 --min-seconds 0 --no-roofline --no-cpu-baseline --no-extra > $R/$O/prof_dcgan.log 2>&1)
ls $O
