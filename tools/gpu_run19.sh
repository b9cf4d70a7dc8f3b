#!/bin/bash
set -u
O=gpurun_out/r2r
mkdir -p $O
export TMPDIR=/tmp
for v in "MIGAN_MFMA_PRIO=1" "MIGAN_MFMA_PRIO=0"; do
  echo "## $v"
  for s in srgan cyclegan dcgan; do env $v python tools/conv_microbench.py --shapes $s --only fwd,dgrad 2>&1 | grep -v amdgpu; done
done > $O/mb_prio.txt
python - <<'PY'
import re
rows={}
cur=None
for l in open('gpurun_out/r2r/mb_prio.txt'):
    if l.startswith('##'): cur=l.split()[1]; continue
    m=re.match(r'(.{28}) (\S+)\s+([\d.]+) us',l)
    if m: rows.setdefault((m.group(1).strip(),m.group(2)),{})[cur]=float(m.group(3))
tot=[0,0]
for k,v in rows.items():
    a,b=v.get('MIGAN_MFMA_PRIO=1'),v.get('MIGAN_MFMA_PRIO=0')
    if a and b and b>100:
        print('%-30s %-7s prio %8.1f  base %8.1f  %+5.1f%%'%(k[0],k[1],a,b,100*(a-b)/b)); tot[0]+=a; tot[1]+=b
print("sum", tot, "%+.2f%%"%(100*(tot[0]-tot[1])/tot[1]))
PY
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"])'
for v in "MIGAN_MFMA_PRIO=1" "MIGAN_MFMA_PRIO=0"; do
  for w in srgan cyclegan; do
  echo "== $w $v"
  env $v timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
  done
  echo "== dcgan $v"
  env $v timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "$J"
done > $O/prio_ab.txt 2>&1
cat $O/prio_ab.txt
