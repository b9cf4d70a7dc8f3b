#!/bin/bash
set -u
O=gpurun_out/r2e
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_fullsize_gpu.py > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["config"].get("hipgraph"))'
for v in "base" "MIGAN_SKINNY=0"; do
  echo "== wgan_gp $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --workload wgan_gp --steps 100 --warmup 10 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
done > $O/wgan_ab.txt 2>&1
cat $O/wgan_ab.txt
for v in "base" "MIGAN_BATCH_MASKS=0"; do
  echo "== dcgan $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "$J"
done > $O/dcgan_ab.txt 2>&1
cat $O/dcgan_ab.txt
for w in cyclegan srgan pix2pix; do
  echo "== $w"
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
done > $O/others.txt 2>&1
cat $O/others.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_wgan -o wgan -- python $R/bench.py --workload wgan_gp --steps 50 --warmup 5 --min-seconds 0 --no-roofline --no-graph > $R/$O/prof_wgan.log 2>&1)
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r2e/prof_wgan/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:14]: print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
ls $O
