#!/bin/bash
set -u
O=gpurun_out/r2l
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -m gpu -q --tb=short -x > $O/pytest_sel.log 2>&1; tail -4 $O/pytest_sel.log
for v in "MIGAN_IGEMM_XCD=1" "MIGAN_IGEMM_XCD=0"; do
  echo "## $v"
  for s in srgan cyclegan dcgan; do env $v python tools/conv_microbench.py --shapes $s --only fwd,dgrad 2>&1 | grep -v amdgpu; done
done > $O/mb_xcd.txt
python - <<'PY'
import re
rows={}
cur=None
for l in open('gpurun_out/r2l/mb_xcd.txt'):
    if l.startswith('##'): cur=l.split()[1]; continue
    m=re.match(r'(.{28}) (\S+)\s+([\d.]+) us',l)
    if m: rows.setdefault((m.group(1).strip(),m.group(2)),{})[cur]=float(m.group(3))
for k,v in rows.items():
    a,b=v.get('MIGAN_IGEMM_XCD=1'),v.get('MIGAN_IGEMM_XCD=0')
    if a and b: print('%-30s %-7s xcd %8.1f  base %8.1f  %+5.1f%%'%(k[0],k[1],a,b,100*(a-b)/b))
PY
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["config"].get("hipgraph"))'
for v in "MIGAN_IGEMM_XCD=1" "MIGAN_IGEMM_XCD=0"; do
  for w in srgan cyclegan; do
  echo "== $w $v"
  env $v timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
  done
  echo "== dcgan $v"
  env $v timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "$J"
done > $O/xcd_ab.txt 2>&1
cat $O/xcd_ab.txt
