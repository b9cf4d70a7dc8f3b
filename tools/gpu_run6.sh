#!/bin/bash
set -u
O=gpurun_out/r2f
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_dp_gpu.py -m gpu -q --tb=short -k "toeplitz or skinny or rccl or conv2d_fwd_bwd or linear" > $O/pytest_sel.log 2>&1; tail -6 $O/pytest_sel.log
python tools/conv_microbench.py --shapes srgan --match conv3 > $O/mb_thin.txt 2>&1
python tools/conv_microbench.py --shapes cyclegan --match c7s1-3 >> $O/mb_thin.txt 2>&1
cat $O/mb_thin.txt
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["config"].get("hipgraph"))'
for v in "base" "MIGAN_TOEPLITZ=0"; do
  for w in srgan cyclegan; do
  echo "== $w $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
  done
done > $O/toep_ab.txt 2>&1
cat $O/toep_ab.txt
echo "== wgan_gp" > $O/wgan.txt
timeout 300 python bench.py --workload wgan_gp --steps 100 --warmup 10 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J" >> $O/wgan.txt
cat $O/wgan.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_wgan -o wgan -- python $R/bench.py --workload wgan_gp --steps 50 --warmup 5 --min-seconds 0 --no-roofline --no-graph > $R/$O/prof_wgan.log 2>&1)
python tools/rocpd_stats.py $O/prof_wgan/*.db 12
