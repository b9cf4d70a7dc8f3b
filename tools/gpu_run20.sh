#!/bin/bash
set -u
O=gpurun_out/r2s
mkdir -p $O
export TMPDIR=/tmp
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"])'
for v in "MIGAN_MFMA_PRIO=0" "MIGAN_MFMA_PRIO=1" "MIGAN_MFMA_PRIO=0" "MIGAN_MFMA_PRIO=1"; do
  echo "== cyclegan $v"
  env $v timeout 300 python bench.py --workload cyclegan --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
  echo "== dcgan $v"
  env $v timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "$J"
  echo "== srgan $v"
  env $v timeout 300 python bench.py --workload srgan --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
done > $O/prio_ab2.txt 2>&1
cat $O/prio_ab2.txt
