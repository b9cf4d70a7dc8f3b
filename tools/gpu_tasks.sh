#!/bin/bash
# Every gpurun call of round 3 as ONE parameterised script:  gpurun --timeout T -- "bash tools/gpu_tasks.sh <task> [args]"
# Tasks write under gpurun_out/<dir>/ (scratch); what is quoted in DESIGN.md was copied into profiles/r03_*.
# Each task starts with tools/gpu_sanity.py: a box whose first HIP call aborts must not burn the call's time limit.
set -u
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 120 python tools/gpu_sanity.py || { echo "bad box, giving up"; exit 3; }

# parity of the LDS-DMA igemm kernels + first A/B against the register-staged kernels
task_parity() {
  # round-3 GPU call 1: parity of the LDS-DMA igemm kernels + A/B microbench against the register-staged kernels
  mkdir -p gpurun_out/r3a
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "conv2d_fwd_bwd or conv_transpose or upconv or reflect_pad1 or conv_double or dropout2d_fused or toeplitz" > gpurun_out/r3a/pytest_conv.txt 2>&1
  echo "pytest rc=$?" >> gpurun_out/r3a/pytest_conv.txt
  tail -5 gpurun_out/r3a/pytest_conv.txt
  for shp in dcgan cyclegan srgan; do
    for dma in 0 1; do
      echo "== $shp MIGAN_DMA=$dma" >> gpurun_out/r3a/mb.txt
      MIGAN_DMA=$dma timeout 300 python tools/conv_microbench.py --shapes $shp --only fwd,dgrad >> gpurun_out/r3a/mb.txt 2>&1
    done
  done
  for t in 128128 128064 256064 64064; do
    echo "== cyclegan R256/u128 MIGAN_DMA_TILE=$t" >> gpurun_out/r3a/mb_tile.txt
    MIGAN_DMA_TILE=$t timeout 300 python tools/conv_microbench.py --shapes cyclegan --only fwd,dgrad --match "256" >> gpurun_out/r3a/mb_tile.txt 2>&1
    echo "== srgan MIGAN_DMA_TILE=$t" >> gpurun_out/r3a/mb_tile.txt
    MIGAN_DMA_TILE=$t timeout 300 python tools/conv_microbench.py --shapes srgan --only fwd,dgrad --match "64" >> gpurun_out/r3a/mb_tile.txt 2>&1
  done
  for dma in 0 1; do
    echo "== bench dcgan MIGAN_DMA=$dma" >> gpurun_out/r3a/bench.txt
    MIGAN_DMA=$dma timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra >> gpurun_out/r3a/bench.txt 2>&1
  done
  cat gpurun_out/r3a/mb.txt | tail -70
}

# tile / BK sweep of the LDS-DMA igemm kernels (MIGAN_DMA_TILE), profiles/r03_dma_tile_sweep.txt
task_tile_sweep() {
  # round-3 GPU call 2: tile / BK sweep of the LDS-DMA igemm kernels (forced tiles, 3 repeats, min reported)
  mkdir -p gpurun_out/r3b
  D="--dirs fwd,dgrad,rdgrad,ufwd,udgrad --repeat 3 --iters 10"
  for round in 1 2; do
  for t in 0 32128128 32256064 32128064 32064064 16128128 16256064 16128064 16064064; do
    echo "== round $round dcgan MIGAN_DMA_TILE=$t" >> gpurun_out/r3b/sweep.txt
    MIGAN_DMA_TILE=$t timeout 120 python tools/conv_microbench.py --shapes dcgan $D --match "G.conv" 2>&1 | grep -v "^/opt" | grep -v "conv3" >> gpurun_out/r3b/sweep.txt
    echo "== round $round cyclegan MIGAN_DMA_TILE=$t" >> gpurun_out/r3b/sweep.txt
    MIGAN_DMA_TILE=$t timeout 120 python tools/conv_microbench.py --shapes cyclegan $D 2>&1 | grep -v "^/opt" | grep -v "c7s1\|D.c1" >> gpurun_out/r3b/sweep.txt
    echo "== round $round srgan MIGAN_DMA_TILE=$t" >> gpurun_out/r3b/sweep.txt
    MIGAN_DMA_TILE=$t timeout 120 python tools/conv_microbench.py --shapes srgan $D 2>&1 | grep -v "^/opt" | grep -v "conv3" >> gpurun_out/r3b/sweep.txt
  done
  done
  tail -30 gpurun_out/r3b/sweep.txt
}

# LDS-DMA wgrad parity + A/B, automatic tile check, whole-step A/B (profiles/r03_dma_vs_regstage.txt)
task_wgrad_ab() {
  # round-3 GPU call 3: LDS-DMA wgrad parity + A/B, auto tile selection check, whole-step A/B
  mkdir -p gpurun_out/r3c
  O=gpurun_out/r3c
  timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv or upconv or reflect_pad1 or toeplitz or bias_grad or direct_grad" > $O/pytest_conv.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest_conv.txt
  tail -5 $O/pytest_conv.txt
  for shp in dcgan cyclegan srgan; do
    for dma in 0 1; do
      echo "== $shp MIGAN_DMA_WGRAD=$dma" >> $O/mb_wgrad.txt
      MIGAN_DMA_WGRAD=$dma timeout 300 python tools/conv_microbench.py --shapes $shp --dirs wgrad,uwgrad,twgrad --repeat 3 --iters 10 2>&1 | grep -v "^/opt" >> $O/mb_wgrad.txt
    done
    echo "== $shp auto" >> $O/mb_auto.txt
    timeout 300 python tools/conv_microbench.py --shapes $shp --dirs fwd,dgrad,rdgrad,ufwd,udgrad --repeat 3 --iters 10 2>&1 | grep -v "^/opt" >> $O/mb_auto.txt
  done
  for wl in dcgan cyclegan srgan; do
    for cfg in "MIGAN_DMA=0 MIGAN_DMA_WGRAD=0" "MIGAN_DMA=1 MIGAN_DMA_WGRAD=0" "MIGAN_DMA=1 MIGAN_DMA_WGRAD=1"; do
      echo "== bench $wl $cfg" >> $O/bench.txt
      env $cfg timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | grep -v "^/opt" | cut -c1-300 >> $O/bench.txt
    done
  done
  cat $O/mb_wgrad.txt; cat $O/bench.txt | cut -c1-200
}

# wgrad BK / planner-occupancy / split sweeps (profiles/r03_wgrad_dma.txt)
task_wgrad_sweep() {
  # round-3 GPU call 4: wgrad LDS-DMA BK 16 vs 32, planner occupancy, split sweep on R256
  mkdir -p gpurun_out/r3d
  O=gpurun_out/r3d
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "conv or upconv or reflect_pad1 or toeplitz or bias_grad or direct_grad" > $O/pytest_conv.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest_conv.txt
  tail -3 $O/pytest_conv.txt
  W="--dirs wgrad,uwgrad,twgrad --repeat 3 --iters 10"
  for shp in dcgan cyclegan srgan; do
    for cfg in "MIGAN_DMA_WGRAD_BK=32 MIGAN_WGRAD_OCCS=3,4,7" "MIGAN_DMA_WGRAD_BK=32 MIGAN_WGRAD_OCCS=2,3,5" "MIGAN_DMA_WGRAD_BK=16 MIGAN_WGRAD_OCCS=3,4,7" "MIGAN_DMA_WGRAD_BK=16 MIGAN_WGRAD_OCCS=4,6,8"; do
      echo "== $shp $cfg" >> $O/mb_wgrad.txt
      env $cfg timeout 300 python tools/conv_microbench.py --shapes $shp $W 2>&1 | grep -v "^/opt" | grep -v "G.conv3\|D.conv\|c7s1-64\|D.c1\| wgrad .*up2" >> $O/mb_wgrad.txt
    done
  done
  for s in 8 12 14 16 21 24 28 32 42 56 64; do
    echo "== splits $s" >> $O/mb_splits.txt
    MIGAN_WGRAD_SPLITS=$s timeout 300 python tools/conv_microbench.py --shapes cyclegan --dirs wgrad,uwgrad --repeat 3 --iters 10 --match "256" 2>&1 | grep -v "^/opt" | grep -v " wgrad .*up2" >> $O/mb_splits.txt
    MIGAN_WGRAD_SPLITS=$s timeout 300 python tools/conv_microbench.py --shapes srgan --dirs wgrad --repeat 3 --iters 10 --match "6" 2>&1 | grep -v "^/opt" >> $O/mb_splits.txt
  done
  for wl in dcgan cyclegan srgan; do
    echo "== bench $wl" >> $O/bench.txt
    timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | grep -v "^/opt" | cut -c1-300 >> $O/bench.txt
  done
  cat $O/bench.txt | cut -c1-200
}

# split-K small-GEMM path: parity, pix2pix / D-conv microbench, whole steps (profiles/r03_splitk.txt)
task_splitk() {
  # round-3 GPU call 5: deep-pipeline + split-K small-GEMM path: parity, pix2pix / dcgan-D microbench A/B, whole steps
  mkdir -p gpurun_out/r3e
  O=gpurun_out/r3e
  timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv or upconv or reflect_pad1 or toeplitz or bias_grad or direct_grad or splitk" > $O/pytest_conv.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest_conv.txt
  tail -5 $O/pytest_conv.txt
  D="--dirs fwd,dgrad --repeat 3 --iters 20"
  for cfg in "MIGAN_DMA_DEEP=0 MIGAN_SPLITK=0" "MIGAN_DMA_DEEP=1 MIGAN_SPLITK=0" "MIGAN_DMA_DEEP=1 MIGAN_SPLITK=1"; do
    echo "== pix2pix $cfg" >> $O/mb_small.txt
    env $cfg timeout 300 python tools/conv_microbench.py --shapes pix2pix $D 2>&1 | grep -v "^/opt" >> $O/mb_small.txt
    echo "== dcgan $cfg" >> $O/mb_small.txt
    env $cfg timeout 300 python tools/conv_microbench.py --shapes dcgan $D --match "D.conv" 2>&1 | grep -v "^/opt" >> $O/mb_small.txt
    echo "== cyclegan $cfg" >> $O/mb_small.txt
    env $cfg timeout 300 python tools/conv_microbench.py --shapes cyclegan $D --match "D.c" 2>&1 | grep -v "^/opt" >> $O/mb_small.txt
  done
  for wl in dcgan pix2pix; do
    for cfg in "MIGAN_DMA_DEEP=0 MIGAN_SPLITK=0" "MIGAN_DMA_DEEP=1 MIGAN_SPLITK=0" "MIGAN_DMA_DEEP=1 MIGAN_SPLITK=1"; do
      echo "== bench $wl $cfg" >> $O/bench.txt
      env $cfg timeout 600 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | grep -v "^/opt" | cut -c1-300 >> $O/bench.txt
    done
  done
  cat $O/bench.txt | cut -c1-200
}

# rocprofv3 --kernel-trace --stats of eager bench steps: gpu_tasks.sh prof <outdir> <workload>...
task_prof() {
  # per-kernel times of whole steps (eager launches), rocprofv3 --kernel-trace --stats -> gpurun_out/$1/
  O=gpurun_out/${1:-r3prof}
  shift
  mkdir -p $O
  R=$GRAFT_REPO_ROOT
  for w in "$@"; do
    k=3; [ $w = dcgan ] && k=20; [ $w = pix2pix ] && k=20; [ $w = wgan_gp ] && k=50
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$w -o $w -- python $R/bench.py --workload $w --steps $k --warmup 2 \
       --min-seconds 0 --no-graph --no-roofline --no-cpu-baseline --no-extra > $R/$O/prof_$w.log 2>&1)
    db=$(ls $O/prof_$w/*/${w}_results.db $O/prof_$w/${w}_results.db 2>/dev/null | head -1)
    # steps counted from the trace (optimiser launches per step), not from --steps: warm-up steps are traced too
    a=2; [ $w = cyclegan ] && a=3; [ $w = wgan_gp ] && a=1.2
    python tools/rocpd_stats.py $db 150 --by-grid --per-step adam_kernel=$a > $O/${w}_kernel_stats.txt 2>&1
    head -3 $O/${w}_kernel_stats.txt
  done
}

# paired D pass: step parity tests + bench A/B (profiles/r03_pair_d_bench.txt)
task_pair_d() {
  # round-3 GPU call 7: paired D pass (BatchNorm groups) parity + A/B, split-K rule check
  mkdir -p gpurun_out/r3g
  O=gpurun_out/r3g
  timeout 900 python -m pytest tests/test_steps_gpu.py tests/test_dp_gpu.py -q -x -k "dcgan or bench_config or two_ranks_on_one_gpu or cross_replica or rccl or acgan or pix2pix" > $O/pytest_steps.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest_steps.txt
  tail -15 $O/pytest_steps.txt
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "batchnorm or norm or conv2d_fwd_bwd or splitk" > $O/pytest_ops.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest_ops.txt
  tail -3 $O/pytest_ops.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
  for cfg in "MIGAN_PAIR_D=0" "MIGAN_PAIR_D=1" "MIGAN_PAIR_D=1 MIGAN_SPLITK=0"; do
    echo "== bench dcgan $cfg" >> $O/bench.txt
    env $cfg timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | grep -v "^/opt" | cut -c1-300 >> $O/bench.txt
  done
  echo "== bench pix2pix" >> $O/bench.txt
  timeout 600 python bench.py --workload pix2pix --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | grep -v "^/opt" | cut -c1-300 >> $O/bench.txt
  cat $O/bench.txt | cut -c1-200
}

# F2 clone tests, N1 trajectory tests, whole ops / data files
task_new_tests() {
  # round-3 GPU call 8: F2 clone tests + N1 trajectory tests + whole ops file
  mkdir -p gpurun_out/r3h
  O=gpurun_out/r3h
  timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_steps_gpu.py -q -k "clone or trajectory or pullaway" --durations=15 > $O/pytest_new.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest_new.txt
  tail -40 $O/pytest_new.txt
  timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_data_gpu.py -q > $O/pytest_ops.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest_ops.txt
  tail -4 $O/pytest_ops.txt
}

# PMC passes over the eager DCGAN step and the microbench of the same layers (in-step vs stand-alone gap; not completed this round)
task_pmc_gap() {
  # In-step vs stand-alone gap of the DCGAN up-conv kernels: PMC passes (separate, --kernel-trace only) over the eager bench
  # step and over the microbench of the same layers -> gpurun_out/r3pmc/
  O=gpurun_out/r3pmc
  mkdir -p $O
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
}

# the driver's round-end command: full pytest -m gpu after a GPU sanity check (profiles/r03_final_pytest_gpu.txt)
task_suite() {
  # full GPU suite on the current tree (the driver's round-end command), after a GPU sanity check
  O=gpurun_out/r3j
  mkdir -p $O
  timeout 1300 python -m pytest tests -m gpu -q --timeout=700 --durations=25 > $O/pytest_gpu.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest_gpu.txt
  grep -v "^  \|^$" $O/pytest_gpu.txt | tail -60
}

t=${1:-}; shift || true
# kernels written at the end of round 3 with no GPU time left (verified on the host execution model only; ON by default):
# parity with each one switched off and on, then the pix2pix step A/B (where all three act) and CycleGAN (PatchGAN head)
#   MIGAN_THIN_WAVE        one-wave-per-pixel thin conv for PatchGAN heads (conv_igemm.hip thin_conv_wave_kernel)
#   MIGAN_WGRAD_REDUCE_TR  transposing wgrad slab reduction for >= 1 M-element weights (wgrad_reduce_tr_kernel)
#   MIGAN_PACK_TR          LDS-tiled OHWI / IHWO weight packs (eltwise.hip pack_transpose_kernel)
#   MIGAN_MIDK             direct conv for the 3 / 6-channel first layers (conv_igemm.hip midk_tile_kernel)
task_staged() {
  mkdir -p gpurun_out/staged
  # torch-free: every staged kernel against the kernel it replaces / host fp64, with launch times (seconds)
  [ -x tools/abi_check.bin ] || tools/build_abi_check.sh
  timeout 120 ./tools/abi_check.bin > gpurun_out/staged/abi_check.txt 2>&1; echo "abi_check rc=$?"; tail -3 gpurun_out/staged/abi_check.txt
  # the fused WGAN-GP kernels as ONE persistent launch (152 / 63 us in round 3) against the default one-launch-per-phase form above
  for s in critic mlp; do MIGAN_K7_PERSIST=1 timeout 60 ./tools/abi_check.bin $s >> gpurun_out/staged/abi_check_persist.txt 2>&1; done
  grep -E "critic_fused|mlp_fused" gpurun_out/staged/abi_check.txt gpurun_out/staged/abi_check_persist.txt
  # then: what does the hardware self-check say, and how long does the probe take (cold, then from the cached verdict)
  for i in 1 2; do
    ( time timeout 300 python -c "
import torch, pytorch_gan_amd
from pytorch_gan_amd import selfcheck
selfcheck.ensure()
print(selfcheck.report()); print('cached' if (selfcheck.VERDICT or {}).get('cached') else 'probed'); print(selfcheck.detail())" ) > gpurun_out/staged/selfcheck_$i.txt 2>&1
  done
  cat gpurun_out/staged/selfcheck_1.txt
  MIGAN_SELFCHECK=inproc timeout 300 python -m pytest tests/test_zz_staged_gpu.py -q -rxs > gpurun_out/staged/selfcheck_inproc.txt 2>&1
  tail -5 gpurun_out/staged/selfcheck_inproc.txt
  timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -q -x \
    -k "conv or pix2pix or cyclegan or srgan or patch" > gpurun_out/staged/pytest.txt 2>&1
  echo "pytest rc=$?" >> gpurun_out/staged/pytest.txt
  timeout 900 python -m pytest tests/test_steps_gpu.py -q -x -k "pix2pix or cyclegan_step or srgan_step" >> gpurun_out/staged/pytest.txt 2>&1
  echo "steps pytest rc=$?" >> gpurun_out/staged/pytest.txt
  grep -E "passed|failed|rc=" gpurun_out/staged/pytest.txt
  OFF="MIGAN_THIN_WAVE=0 MIGAN_WGRAD_REDUCE_TR=0 MIGAN_PACK_TR=0 MIGAN_MIDK=0 MIGAN_NORM_SMALL=0 MIGAN_SMALLK_PB16=0 MIGAN_DROPOUT_FUSE=0 MIGAN_FEWPIX=0"
  for k7 in 0 1; do
    echo "== wgan_gp [MIGAN_K7=$k7]" >> gpurun_out/staged/bench.txt
    env MIGAN_K7=$k7 timeout 300 python bench.py --workload wgan_gp --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-roofline \
      2>>gpurun_out/staged/bench.err | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'])" >> gpurun_out/staged/bench.txt
  done
  timeout 600 python -m pytest tests/test_steps_gpu.py tests/test_fullsize_gpu.py -q -x -k "wgan or critic" >> gpurun_out/staged/pytest.txt 2>&1
  echo "wgan pytest rc=$?" >> gpurun_out/staged/pytest.txt
  for w in pix2pix cyclegan; do
    for f in "$OFF" "MIGAN_WGRAD_REDUCE_TR=0 MIGAN_PACK_TR=0" "MIGAN_THIN_WAVE=0 MIGAN_PACK_TR=0" "MIGAN_THIN_WAVE=0 MIGAN_WGRAD_REDUCE_TR=0" "MIGAN_MIDK=0" "MIGAN_NORM_SMALL=0" "MIGAN_SMALLK_PB16=0" "MIGAN_FEWPIX=0" "MIGAN_X=1"; do
      echo "== $w [$f]" >> gpurun_out/staged/bench.txt
      env $f timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-roofline \
        2>>gpurun_out/staged/bench.err | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'])" >> gpurun_out/staged/bench.txt
    done
  done
  cat gpurun_out/staged/bench.txt
}

case "$t" in
  parity) task_parity "$@" ;;
  tile_sweep) task_tile_sweep "$@" ;;
  wgrad_ab) task_wgrad_ab "$@" ;;
  wgrad_sweep) task_wgrad_sweep "$@" ;;
  splitk) task_splitk "$@" ;;
  prof) task_prof "$@" ;;
  pair_d) task_pair_d "$@" ;;
  new_tests) task_new_tests "$@" ;;
  pmc_gap) task_pmc_gap "$@" ;;
  staged) task_staged "$@" ;;
  suite) task_suite "$@" ;;
  *) echo "usage: gpu_tasks.sh {parity|tile_sweep|wgrad_ab|wgrad_sweep|splitk|prof|pair_d|new_tests|pmc_gap|staged|suite} [args]"; exit 2 ;;
esac
