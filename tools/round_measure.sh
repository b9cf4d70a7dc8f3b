#!/bin/bash
# Round-end measurement pass on the GPU box (run through gpurun from the repo root); everything lands in gpurun_out/final/ and
# is turned into the committed evidence under profiles/ by tools/collect_r02.py + tools/pmc_kernels.py.
#   gpurun --timeout 2400 -- 'bash tools/round_measure.sh'
set -u
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# 1. parity: the whole GPU suite incl. the BASELINE-size step tests (oracle on the host cores)
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
# 2. HBM traffic and MFMA-busy of the conv-family launches: separate --pmc passes (--kernel-trace only) over the microbench,
#    cut into (layer, direction) segments by marker launches (tools/pmc_kernels.py)
pmc() {  # name, counters..., -- microbench args
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  mkdir -p $R/$O/pmc_$name
  (cd /tmp && timeout 400 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $R/$O/pmc_$name -o p -- \
     python $R/tools/conv_microbench.py --iters 3 --pmc-log $R/$O/pmc_$name/segments.jsonl "$@" > $R/$O/pmc_$name.log 2>&1)
}
DC="--shapes dcgan --match G.conv --dirs ufwd,udgrad,uwgrad"
CY="--shapes cyclegan --dirs fwd,rdgrad,wgrad,ufwd,udgrad,uwgrad,tfwd,texpand,twgrad"
SR="--shapes srgan --dirs fwd,dgrad,wgrad,tfwd,texpand,twgrad,tdgrad"
for set in DC CY SR; do
  pmc ${set}_fetch FETCH_SIZE -- ${!set}
  pmc ${set}_write WRITE_SIZE -- ${!set}
  pmc ${set}_mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA -- ${!set}
done
python tools/pmc_kernels.py r02 $O/pmc_DC_fetch $O/pmc_DC_write $O/pmc_DC_mfma $O/pmc_CY_fetch $O/pmc_CY_write $O/pmc_CY_mfma \
    $O/pmc_SR_fetch $O/pmc_SR_write $O/pmc_SR_mfma > $O/pmc_kernels.log 2>&1; tail -45 $O/pmc_kernels.log
cp profiles/r02_pmc_kernels.json $O/r02_pmc_kernels.json
# 3. per-kernel times of the bench command itself (eager launches: a graph replay shows up as one node)
for w in dcgan cyclegan srgan; do
  k=3; [ $w = dcgan ] && k=20
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$w -o $w -- python $R/bench.py --workload $w --steps $k --warmup 2 \
     --min-seconds 0 --no-graph --no-roofline --no-cpu-baseline --no-extra > $R/$O/prof_$w.log 2>&1)
  cp $O/prof_$w/${w}_results.db $O/${w}_results.db 2>/dev/null
done
# 4. the bench line the driver will take (default flags) + the other workloads
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo
for w in pix2pix esrgan; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1
done > $O/bench_others.jsonl
# 5. stand-alone layer timings of the final build
for s in dcgan cyclegan srgan; do python tools/conv_microbench.py --shapes $s --iters 10 2>&1 | grep -v amdgpu.ids; done > $O/conv_microbench.txt
ls $O | head -60
