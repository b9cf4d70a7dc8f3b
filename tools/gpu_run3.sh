#!/bin/bash
# Round-2 pass 3: full GPU test suite on the new code (F1/F2, planner), plan A/B, WGAN-GP kernel profile.  -> gpurun_out/r2c/
set -u
O=gpurun_out/r2c
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_fullsize_gpu.py 2>&1 | tail -30 > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
MB="python $R/tools/conv_microbench.py"
for plan in 1 3; do
  for s in dcgan cyclegan srgan; do MIGAN_WGRAD_PLAN=$plan timeout 300 $MB --shapes $s --iters 10 --only wgrad 2>&1 | grep -v amdgpu.ids; done > $O/mb_wgrad_plan$plan.txt
done
paste -d'|' <(cut -c1-62 $O/mb_wgrad_plan1.txt) <(cut -c29-62 $O/mb_wgrad_plan3.txt)
for s in cyclegan srgan; do timeout 300 $MB --shapes $s --iters 10 --only fwd,dgrad 2>&1 | grep -v amdgpu.ids; done > $O/mb_fwd_dgrad.txt
grep -E "D 512|R256" $O/mb_fwd_dgrad.txt
for v in "base" "MIGAN_WGRAD_PLAN=1" "MIGAN_WGRAD_OVERLAP=0"; do
  echo "== cyclegan $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --workload cyclegan --steps 3 --warmup 1 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'])"
done > $O/cyclegan_ab.txt 2>&1
cat $O/cyclegan_ab.txt
for v in "base" "MIGAN_WGRAD_PLAN=1" "MIGAN_WGRAD_OVERLAP=0"; do
  echo "== dcgan $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'])"
done > $O/dcgan_ab.txt 2>&1
cat $O/dcgan_ab.txt
for w in srgan wgan_gp pix2pix; do
  timeout 300 python bench.py --workload $w --steps $([ $w = srgan ] && echo 4 || echo 100) --warmup 3 --min-seconds 1 --no-cpu-baseline 2>/dev/null > $O/bench_$w.json; python -c "import sys,json; d=json.load(open('$O/bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d['config']['hipgraph'], d.get('roofline',{}).get('kernel'))"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_wgan -o wgan -- python $R/bench.py --workload wgan_gp --steps 50 --warmup 5 --min-seconds 0 --no-roofline --no-graph > $R/$O/prof_wgan.log 2>&1)
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
ls $O
