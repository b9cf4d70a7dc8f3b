#!/bin/bash
set -u
O=gpurun_out/r2h
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_steps_gpu.py tests/test_ops_gpu.py -m gpu -q --tb=short -k "esrgan or conv2d_fwd_bwd or upconv" > $O/pytest_sel.log 2>&1; tail -4 $O/pytest_sel.log
for v in "MIGAN_IGEMM_OCC5=1" "MIGAN_IGEMM_OCC5=0"; do
  echo "## $v"
  env $v python tools/conv_microbench.py --shapes srgan --only fwd,dgrad 2>&1 | grep -v amdgpu
  env $v python tools/conv_microbench.py --shapes cyclegan --only fwd,dgrad 2>&1 | grep -v amdgpu
  env $v python tools/conv_microbench.py --shapes dcgan --only fwd,dgrad 2>&1 | grep -v amdgpu
done > $O/mb_occ5.txt
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["config"].get("hipgraph"))'
for v in "MIGAN_IGEMM_OCC5=1" "MIGAN_IGEMM_OCC5=0"; do
  for w in srgan cyclegan esrgan; do
  echo "== $w $v"
  env $v timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
  done
  echo "== dcgan $v"
  env $v timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "$J"
done > $O/occ5_ab.txt 2>&1
cat $O/occ5_ab.txt
python - <<'PY'
import re
rows={}
cur=None
for l in open('gpurun_out/r2h/mb_occ5.txt'):
    if l.startswith('##'): cur=l.split()[1]; continue
    m=re.match(r'(.{28}) (\S+)\s+([\d.]+) us',l)
    if m: rows.setdefault((m.group(1).strip(),m.group(2)),{})[cur]=float(m.group(3))
for k,v in rows.items():
    a,b=v.get('MIGAN_IGEMM_OCC5=1'),v.get('MIGAN_IGEMM_OCC5=0')
    if a and b and abs(a-b)/b>0.03: print('%-30s %-7s occ5 %8.1f  base %8.1f  %+5.1f%%'%(k[0],k[1],a,b,100*(a-b)/b))
PY
