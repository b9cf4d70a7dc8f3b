#!/bin/bash
set -u
O=gpurun_out/r2m
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_steps_gpu.py -m gpu -q --tb=short -x > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["config"].get("hipgraph"))'
for i in 1 2; do
for w in srgan cyclegan; do
  echo "== $w"
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
done
echo "== srgan fusion off"
python - <<'PY'
import subprocess, sys, json, os
env = dict(os.environ, MIGAN_NO_PRELU_FUSE="1")
out = subprocess.run([sys.executable, "bench.py", "--workload", "srgan", "--steps", "5", "--warmup", "2", "--min-seconds", "1", "--no-roofline", "--no-cpu-baseline"], capture_output=True, text=True, env=env).stdout
d = json.loads(out.splitlines()[-1]); print(d["value"], d["ms_per_step"])
PY
done > $O/prelu_ab.txt 2>&1
cat $O/prelu_ab.txt
