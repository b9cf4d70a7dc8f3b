#!/bin/bash
set -u
O=gpurun_out/r2t
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_steps_gpu.py tests/test_fullsize_gpu.py -m gpu -q --tb=short -x -k "cyclegan or reflect or conv2d_fwd_bwd" > $O/pytest_sel.log 2>&1; tail -4 $O/pytest_sel.log | cut -c1-200
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"])'
for v in "MIGAN_RING_OVERLAP=1" "MIGAN_RING_OVERLAP=0" "MIGAN_RING_OVERLAP=1" "MIGAN_RING_OVERLAP=0"; do
  echo "== cyclegan $v"
  env $v timeout 300 python bench.py --workload cyclegan --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
done > $O/ring_ab.txt 2>&1
cat $O/ring_ab.txt
