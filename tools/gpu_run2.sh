#!/bin/bash
# Round-2 pass 2: counters on the dominant conv kernels + sweeps of the A/B knobs.  Output: gpurun_out/r2b/.
set -u
O=gpurun_out/r2b
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py tests/test_models_gpu.py -m gpu -x -q 2>&1 | tail -12 > $O/pytest.log; tail -3 $O/pytest.log
MB="python $R/tools/conv_microbench.py"
# ---- PMC passes (counter collection serialises kernels; timings inside these passes are not comparable)
pmc() { # name, counters..., then -- command
  n=$1; shift; ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $R/$O/pmc_$n -o p -- "$@" > $R/$O/pmc_$n.log 2>&1)
}
for shp in "cyclegan R256" "dcgan G.conv2"; do
  set -- $shp; tag=$2
  pmc ${tag}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS -- $MB --shapes $1 --match $2 --iters 3
  pmc ${tag}_sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM -- $MB --shapes $1 --match $2 --iters 3
  pmc ${tag}_grbm GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_SALU TCC_HIT_sum TCC_MISS_sum -- $MB --shapes $1 --match $2 --iters 3
  pmc ${tag}_fetch FETCH_SIZE -- $MB --shapes $1 --match $2 --iters 3
  pmc ${tag}_write WRITE_SIZE -- $MB --shapes $1 --match $2 --iters 3
  python tools/pmc_summary.py $O/pmc_${tag}_sq1 $O/pmc_${tag}_sq2 $O/pmc_${tag}_grbm $O/pmc_${tag}_fetch $O/pmc_${tag}_write > $O/pmc_${tag}.txt 2>&1
done
head -50 $O/pmc_R256.txt
# ---- sweeps (un-profiled)
{
for sp in 8 16 21 24 32 40 42 48 64; do echo "== R256/d256/vgg wgrad SPLITS=$sp"; MIGAN_WGRAD_SPLITS=$sp $MB --shapes cyclegan --only wgrad --iters 10 2>&1 | grep -E "R256|d256|d128|D.c3"; MIGAN_WGRAD_SPLITS=$sp $MB --shapes srgan --only wgrad --iters 5 2>&1 | grep -E "vgg 256|res 64"; done
for sp in 16 32 48 64 96 128; do echo "== uwgrad SPLITS=$sp"; MIGAN_WGRAD_SPLITS=$sp $MB --shapes dcgan --only wgrad --iters 10 2>&1 | grep -E "uwgrad"; done
} > $O/sweep_splits.txt 2>&1
{
echo "== base"; for s in dcgan cyclegan srgan; do $MB --shapes $s --iters 10 2>&1 | grep -v amdgpu.ids; done
echo "== MIGAN_MFMA_V4=3"; for s in dcgan cyclegan srgan; do MIGAN_MFMA_V4=3 $MB --shapes $s --iters 10 2>&1 | grep -v amdgpu.ids; done
echo "== MIGAN_WGRAD_OCC=4"; for s in cyclegan srgan; do MIGAN_WGRAD_OCC=4 $MB --shapes $s --iters 10 --only wgrad 2>&1 | grep -v amdgpu.ids; done
for t in 128128 128064 64064; do echo "== MIGAN_IGEMM_TILE=$t"; for s in dcgan cyclegan srgan; do MIGAN_IGEMM_TILE=$t $MB --shapes $s --iters 10 --only fwd,dgrad 2>&1 | grep -v amdgpu.ids; done; done
} > $O/sweep_variants.txt 2>&1
for v in "base" "MIGAN_MFMA_V4=1" "MIGAN_MFMA_V4=2" "MIGAN_MFMA_V4=3" "MIGAN_WGRAD_OVERLAP=1" "MIGAN_COLSUM_FUSE=0"; do
  echo "== dcgan $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'])"
done > $O/dcgan_ab.txt 2>&1
cat $O/dcgan_ab.txt
for v in "base" "MIGAN_MFMA_V4=3" "MIGAN_WGRAD_OVERLAP=1" "MIGAN_WGRAD_OCC=4"; do
  echo "== cyclegan $v"
  env $( [ "$v" = base ] && echo X=1 || echo $v ) timeout 300 python bench.py --workload cyclegan --steps 3 --warmup 1 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'])"
done > $O/cyclegan_ab.txt 2>&1
cat $O/cyclegan_ab.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_dcgan -o dcgan -- python $R/bench.py --steps 20 --warmup 5 --min-seconds 0 --no-roofline --no-cpu-baseline --no-extra > $R/$O/prof_dcgan.log 2>&1)
ls $O | head -50
