#!/bin/bash
set -u
O=gpurun_out/r2q
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_steps_gpu.py -m gpu -q --tb=short -x > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log | cut -c1-200
for v in "X=1" "MIGAN_NO_SHUFFLE_FUSE=1" "X=2" "MIGAN_NO_SHUFFLE_FUSE=1"; do
  echo "== srgan $v"
  env $v timeout 300 python bench.py --workload srgan --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"])'
done > $O/shuffle_ab.txt 2>&1
cat $O/shuffle_ab.txt
