#!/bin/bash
set -u
O=gpurun_out/r2g
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_fullsize_gpu.py > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["config"].get("hipgraph"))'
for v in "base" "MIGAN_BATCH_MASKS=0"; do
  echo "== pix2pix $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --workload pix2pix --steps 30 --warmup 5 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
done > $O/pix2pix_ab.txt 2>&1
cat $O/pix2pix_ab.txt
echo "== esrgan" > $O/esrgan.txt
timeout 400 python bench.py --workload esrgan --steps 3 --warmup 1 --min-seconds 1 --no-roofline --no-cpu-baseline 2>$O/esrgan.err | python -c "$J" >> $O/esrgan.txt
cat $O/esrgan.txt; tail -3 $O/esrgan.err
