#!/bin/bash
# In-step vs stand-alone gap of the DCGAN up-conv kernels: PMC passes (separate, --kernel-trace only) over the eager bench
# step and over the microbench of the same layers -> gpurun_out/r3pmc/
O=gpurun_out/r3pmc
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BENCH="python $R/bench.py --workload dcgan --steps 6 --warmup 2 --min-seconds 0 --no-graph --no-roofline --no-cpu-baseline --no-extra"
MICRO="python $R/tools/conv_microbench.py --shapes dcgan --match G.conv --dirs ufwd,udgrad,uwgrad --iters 6"
pass() {  # name, target cmd, counters...
  local name=$1; shift
  local cmd=$1; shift
  mkdir -p $R/$O/$name
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$O/$name -o p -- $cmd > $R/$O/$name.log 2>&1)
}
for tgt in bench micro; do
  cmd="$BENCH"; [ $tgt = micro ] && cmd="$MICRO"
  pass ${tgt}_sq "$cmd" GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
  pass ${tgt}_l2 "$cmd" TCC_HIT_sum TCC_MISS_sum
  pass ${tgt}_fetch "$cmd" FETCH_SIZE
  pass ${tgt}_write "$cmd" WRITE_SIZE
done
find $O -name "*.csv" | head -20
f=$(find $O/bench_sq -name "*counter_collection.csv" | head -1); head -3 $f
f=$(find $O/bench_sq -name "*kernel_trace.csv" | head -1); head -3 $f
for tgt in bench micro; do
  python tools/pmc_summary.py $O/${tgt}_sq $O/${tgt}_l2 $O/${tgt}_fetch $O/${tgt}_write > $O/${tgt}_summary.txt 2>&1
done
grep -A4 "igemm_dma_kernel<128, 64\|wgrad_dma_kernel<64, 128, 32, true" $O/bench_summary.txt | head -40
echo ---- micro
grep -A4 "igemm_dma_kernel<128, 64\|wgrad_dma_kernel<64, 128, 32, true" $O/micro_summary.txt | head -40
