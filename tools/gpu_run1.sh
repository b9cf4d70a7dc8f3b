#!/bin/bash
# Round-2 measurement pass 1 (run through gpurun from the repo root); everything lands in gpurun_out/r2a/.
set -u
O=gpurun_out/r2a
mkdir -p $O
export TMPDIR=/tmp
python - > $O/env.txt 2>&1 <<'PY'
import os, torch, psutil
print("cpus", os.cpu_count(), "mem_gb", psutil.virtual_memory().total / 2**30, "gpus", torch.cuda.device_count(), torch.cuda.get_device_name(0))
PY
cat $O/env.txt
export MIGAN_TEST_ERRLOG=$PWD/$O/test_errors.log
rm -f $MIGAN_TEST_ERRLOG
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_fullsize_gpu.py 2>&1 | tail -15 > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q --durations=10 2>&1 | tail -25 > $O/pytest_full.log; tail -6 $O/pytest_full.log
unset MIGAN_TEST_ERRLOG
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; head -c 600 $O/bench.json; echo
for plan in 1 2; do
  for s in dcgan cyclegan srgan; do MIGAN_WGRAD_PLAN=$plan timeout 300 python tools/conv_microbench.py --shapes $s --iters 10 --only wgrad 2>&1 | grep -v amdgpu.ids; done > $O/mb_wgrad_plan$plan.txt
done
for s in dcgan cyclegan srgan; do timeout 300 python tools/conv_microbench.py --shapes $s --iters 10 --only fwd,dgrad 2>&1 | grep -v amdgpu.ids; done > $O/mb_fwd_dgrad.txt
grep -E "R256|rdgrad|fold" $O/mb_fwd_dgrad.txt
for v in "base" "MIGAN_WGRAD_OVERLAP=1" "MIGAN_REFLECT1=0" "MIGAN_WGRAD_PLAN=1" "MIGAN_COLSUM_FUSE=0"; do
  echo "== cyclegan $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --workload cyclegan --steps 3 --warmup 1 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'])"
done > $O/cyclegan_ab.txt 2>&1
cat $O/cyclegan_ab.txt
for v in "base" "MIGAN_WGRAD_PLAN=1" "MIGAN_COLSUM_FUSE=0"; do
  echo "== dcgan $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'])"
done > $O/dcgan_ab.txt 2>&1
cat $O/dcgan_ab.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_cyc -o cyc -- python $GRAFT_REPO_ROOT/bench.py --workload cyclegan --steps 2 --warmup 1 --min-seconds 0 --no-roofline > $GRAFT_REPO_ROOT/$O/prof_cyc.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_dcgan -o dcgan -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --min-seconds 0 --no-roofline --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/prof_dcgan.log 2>&1)
ls $O $O/prof_cyc | head -40
