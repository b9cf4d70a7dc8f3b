#!/bin/bash
set -u
O=gpurun_out/r2j
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_fullsize_gpu.py -x > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["config"].get("hipgraph"))'
for v in "MIGAN_BATCH_PACKS=1" "MIGAN_BATCH_PACKS=0"; do
  echo "== dcgan $v"
  env $v timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "$J"
  echo "== pix2pix $v"
  env $v timeout 300 python bench.py --workload pix2pix --steps 30 --warmup 5 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
  for w in esrgan srgan; do
  echo "== $w $v"
  env $v timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
  done
done > $O/packs_ab.txt 2>&1
cat $O/packs_ab.txt
