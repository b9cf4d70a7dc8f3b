#!/bin/bash
# Round-5 gpurun calls:  gpurun --timeout T -- "bash tools/gpu_r05.sh <task> [args]"
# Tasks write under gpurun_out/<dir>/ (scratch); what DESIGN.md quotes is copied into profiles/r05_*.
# ab_base/ = `git archive c6c64d7` (the round-start tree) + its built library: git-ignored, travels with the snapshot.
set -u
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 120 python tools/gpu_sanity.py || { echo "bad box, giving up"; exit 3; }

line() { python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['timing']['ms_per_step_min'], 'graph' if r['config'].get('hipgraph') else 'eager', r['config'].get('hipgraph_error',''))"; }

# bench one workload:  bl <outfile> <workload> <steps> [ENV=VAL ... | --flag ...]  -> "img/s ms/step min-block"
bl() {
  local out=$1 w=$2 k=$3; shift 3
  local envs=() flags=()
  for a in "$@"; do case $a in --*|[0-9]*) flags+=("$a") ;; *) envs+=("$a") ;; esac; done
  echo "== $w $*" >> $out
  env "${envs[@]}" timeout 300 python bench.py --workload $w --steps $k --warmup 2 --no-cpu-baseline --no-extra --no-roofline "${flags[@]}" 2>>$out.err | line >> $out
}

# whole-step A/B on ONE box against the round-start tree:  ab <outfile> <workload> <steps> [reps]
ab() {
  local out=$1 w=$2 k=$3 reps=${4:-2}
  for r in $(seq $reps); do
    (cd ab_base && echo "== base $w" >> $R/$out && timeout 300 python bench.py --workload $w --steps $k --warmup 2 --no-cpu-baseline --no-extra --no-roofline 2>>$R/$out.err | line >> $R/$out)
    bl $out $w $k
  done
}

# rocprofv3 --kernel-trace --stats of bench steps:  prof <outdir> <workload>[:graph][@batch] ...
task_prof() {
  local O=gpurun_out/${1:-r5prof}; shift
  mkdir -p $O
  for spec in "$@"; do
    local b=; case $spec in *@*) b=${spec##*@}; spec=${spec%@*} ;; esac
    w=${spec%%:*}; mode=eager; flag="--no-graph --no-overlap"; [ "$spec" != "$w" ] && { mode=graph; flag=; }
    k=3; [ $w = dcgan ] && k=20; [ $w = pix2pix ] && k=20; [ $w = wgan_gp ] && k=50
    tag=$w; [ -n "$b" ] && { flag="$flag --batch $b"; tag=${w}_bs$b; k=10; }
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_${tag}_$mode -o $w -- python $R/bench.py --workload $w --steps $k --warmup 2 \
       --min-seconds 0 $flag --no-roofline --no-cpu-baseline --no-extra > $R/$O/prof_${tag}_$mode.log 2>&1)
    db=$(ls $O/prof_${tag}_$mode/*/${w}_results.db $O/prof_${tag}_$mode/${w}_results.db 2>/dev/null | head -1)
    a=2; [ $w = cyclegan ] && a=3; [ $w = wgan_gp ] && a=1.2
    python tools/rocpd_stats.py $db 150 --by-grid --per-step adam_kernel=$a > $O/${tag}_${mode}_kernel_stats.txt 2>&1
    head -4 $O/${tag}_${mode}_kernel_stats.txt
    rm -rf $O/prof_${tag}_$mode
  done
}

task_pmcstep() {
  local w=$1 k=$2; shift 2
  local O=gpurun_out/r5pmcstep_$w
  mkdir -p $O
  for name in "$@"; do
    case $name in
      sq) ctr="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" ;;
      l2) ctr="TCC_HIT_sum TCC_MISS_sum" ;;
      fetch) ctr="FETCH_SIZE" ;;
      write) ctr="WRITE_SIZE" ;;
    esac
    mkdir -p $R/$O/$name
    (cd /tmp && timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/$O/$name -o p -- \
       python $R/bench.py --workload $w --steps $k --warmup 1 --pmc-log $R/$O/$name/segments.json > $R/$O/$name.log 2>&1)
    tail -1 $O/$name.log | cut -c1-200
  done
  python tools/pmc_step.py r05 $(for n in "$@"; do echo $O/$n; done) > $O/summary.txt 2>&1
  head -30 $O/summary.txt | cut -c1-200
  find $O -name "*.csv" -size +6M -delete
  cp profiles/r05_pmc_kernels.json gpurun_out/r05_pmc_kernels.json 2>/dev/null
}

task_bench() {
  local O=gpurun_out/r5bench; mkdir -p $O
  timeout 900 python bench.py "$@" > $O/bench_default.json 2> $O/bench_default.err
  echo "bench rc=$?"; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r5bench/bench_default.json').read().strip().splitlines()[-1])
print({k:r[k] for k in ('value','ms_per_step')})
rf=r.get('roofline',{}); print({k:rf.get(k) for k in ('kernel','frac','traffic','mfma_busy_frac','symbol','avg_launch_ms')})
print('cpu', r.get('cpu_baseline'))
for k,v in r.get('extra',{}).items(): print(k, {a:v.get(a) for a in ('images_per_s','ms_per_step','hipgraph','eager_ms_per_step','library_launches_per_step','error','hipgraph_error')}, (v.get('cpu_baseline') or {}).get('value'))
PY
  tail -5 $O/bench_default.err
}

task_suite() {
  local O=gpurun_out/r5suite; mkdir -p $O
  timeout 1500 python -m pytest tests -m gpu -q --timeout=700 --durations=25 -rxs > $O/pytest_gpu.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest_gpu.txt
  grep -v "^  \|^$" $O/pytest_gpu.txt | tail -60
}

task_first() {   # call 1: what the CPU cannot check - the recorded CycleGAN step, one image per GPU, both dp orders, the SRGAN fork fix
  local O=gpurun_out/r5a; mkdir -p $O
  timeout 900 python -m pytest tests/test_steps_gpu.py tests/test_fullsize_gpu.py tests/test_dp_gpu.py -q -x \
     -k "cyclegan_recorded or cyclegan_256_bs1 or both_step_orders or two_ranks or second_stream or srgan_step or cyclegan_steps or world1" --durations=8 > $O/pytest.txt 2>&1
  tail -15 $O/pytest.txt
  bl $O/bench.txt cyclegan 10 --batch 1
  bl $O/bench.txt cyclegan 10 --batch 1 --no-graph
  bl $O/bench.txt cyclegan 4
  bl $O/bench.txt cyclegan 4 --no-graph
  ab $O/bench.txt srgan 4 1
  cat $O/bench.txt
  tail -5 $O/bench.txt.err 2>/dev/null
}

task_probe() {   # call 2: which fork of the CycleGAN step body the capture does not survive (segfault in hipStreamEndCapture, call 1)
  local O=gpurun_out/r5b; mkdir -p $O
  for cfg in "0 0 0" "0 1 0" "1 0 0" "0 0 1" "1 1 0" "1 1 1"; do
    timeout 120 python tools/capture_probe.py $cfg >> $O/probe.txt 2>&1
    echo "  rc=$?" >> $O/probe.txt
  done
  PROBE_WGRAD_MIN=4096 timeout 120 python tools/capture_probe.py 0 0 1 >> $O/probe.txt 2>&1; echo "  rc=$? (wgrad min 4096)" >> $O/probe.txt
  PROBE_WGRAD_MIN=4096 timeout 120 python tools/capture_probe.py 1 1 1 >> $O/probe.txt 2>&1; echo "  rc=$? (wgrad min 4096)" >> $O/probe.txt
  timeout 200 python tools/capture_probe.py 1 1 1 256 9 1 >> $O/probe.txt 2>&1; echo "  rc=$? (256, 9 blocks, bs 1)" >> $O/probe.txt
  timeout 200 python tools/capture_probe.py 0 0 0 256 9 1 >> $O/probe.txt 2>&1; echo "  rc=$? (256, 9 blocks, bs 1)" >> $O/probe.txt
  grep -v "amdgpu.ids\|^  File\|Extension modules" $O/probe.txt | cut -c1-220
}

task_probe2() {   # call 3: the crash needs 256x256 / 9 blocks / batch 1 with the forks on - which fork, which kernel family; native backtrace
  local O=gpurun_out/r5c; mkdir -p $O
  pr() { local tag=$1; shift; env "$@" > /dev/null 2>&1; }
  run() { local note=$1; shift; timeout 200 "$@" >> $O/probe.txt 2>&1; echo "  rc=$? ($note)" >> $O/probe.txt; }
  run "256/9/1" python tools/capture_probe.py 1 0 0 256 9 1
  run "256/9/1" python tools/capture_probe.py 0 1 0 256 9 1
  run "256/9/1" python tools/capture_probe.py 0 0 1 256 9 1
  run "256/9/1" python tools/capture_probe.py 1 1 0 256 9 1
  run "256/2/1" python tools/capture_probe.py 1 1 1 256 2 1
  run "128/9/1" python tools/capture_probe.py 1 1 1 128 9 1
  run "256/9/1 no toeplitz" env MIGAN_TOEPLITZ=0 python tools/capture_probe.py 1 1 1 256 9 1
  run "256/9/1 no split-K" env MIGAN_SPLITK=0 python tools/capture_probe.py 1 1 1 256 9 1
  run "256/9/1 record_stream instead of held references" env MIGAN_CAPTURE_REFS=0 python tools/capture_probe.py 1 1 1 256 9 1
  grep -v "amdgpu.ids\|^  File\|Extension modules" $O/probe.txt | cut -c1-220
  timeout 300 /opt/rocm/bin/rocgdb -batch -ex run -ex bt -ex "info threads" --args python tools/capture_probe.py 1 1 1 256 9 1 > $O/gdb.txt 2>&1
  grep -n "SIGSEGV" -A 40 $O/gdb.txt | cut -c1-200 | head -80
}

task_fourth() {   # call 4: the capture with per-stream pack copies; then call 1's list
  local O=gpurun_out/r5d; mkdir -p $O
  timeout 200 python tools/capture_probe.py 1 1 1 256 9 1 > $O/probe.txt 2>&1; echo "  rc=$?" >> $O/probe.txt
  timeout 200 python tools/capture_probe.py 1 1 1 256 9 8 >> $O/probe.txt 2>&1; echo "  rc=$?" >> $O/probe.txt
  grep -v "amdgpu.ids\|^  File\|Extension modules" $O/probe.txt | cut -c1-220
  task_first
}

task_fifth() {   # call 5: the tests of call 1 again (test fixed), kernel traces of the one-image CycleGAN step, eager and recorded
  local O=gpurun_out/r5e; mkdir -p $O
  timeout 900 python -m pytest tests/test_steps_gpu.py tests/test_fullsize_gpu.py tests/test_dp_gpu.py -q \
     -k "cyclegan_recorded or cyclegan_256_bs1 or both_step_orders or two_ranks or second_stream or srgan_step or cyclegan_steps or world1" --durations=8 > $O/pytest.txt 2>&1
  tail -15 $O/pytest.txt
  task_prof r5e cyclegan@1 cyclegan:graph@1
}

task_sixth() {   # call 6: split-K reach (256 tiles x >= 64 K-tiles) and the reflect-1 ring with a workspace: parity, one image per GPU, batch 8 A/B
  local O=gpurun_out/r5f; mkdir -p $O
  timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py tests/test_fullsize_gpu.py -q \
     -k "reflect or splitk or cyclegan_recorded or cyclegan_256_bs1 or conv2d_fwd_bwd or geometry_selects" --durations=5 > $O/pytest.txt 2>&1
  tail -6 $O/pytest.txt
  bl $O/bench.txt cyclegan 10 --batch 1
  bl $O/bench.txt cyclegan 10 --batch 1 MIGAN_SPLITK=0
  bl $O/bench.txt cyclegan 10 --batch 1 --no-graph
  ab $O/bench.txt cyclegan 4 1
  bl $O/bench.txt cyclegan 4 --no-graph
  ab $O/bench.txt pix2pix 50 1
  cat $O/bench.txt
}

task_rgb() {   # call 8: the image-input kernels (csrc/rgb_conv.hip): parity on the hardware, stand-alone times, SRGAN / CycleGAN whole steps
  local O=gpurun_out/r5h; mkdir -p $O
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py tests/test_fullsize_gpu.py -q -k "rgb_conv or srgan or test_conv2d_fwd_bwd" --durations=5 > $O/pytest.txt 2>&1
  tail -6 $O/pytest.txt
  timeout 200 python tools/conv_microbench.py --shapes srgan --match "3->64" --iters 20 --repeat 3 > $O/micro.txt 2>&1
  timeout 200 python tools/conv_microbench.py --shapes cyclegan --match "c7s1-64" --iters 20 --repeat 3 >> $O/micro.txt 2>&1
  cat $O/micro.txt
  for r in 1 2; do
    bl $O/bench.txt srgan 4 MIGAN_RGB=0
    bl $O/bench.txt srgan 4
  done
  bl $O/bench.txt cyclegan 4 MIGAN_RGB=0 --no-graph
  bl $O/bench.txt cyclegan 4 --no-graph
  cat $O/bench.txt
}

task_toep() {   # call 10: strip-walking Toeplitz weight gradient, 8 x 16 M blocks, Co' = 32: parity, stand-alone times, whole steps
  local O=gpurun_out/r5i; mkdir -p $O
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py -q -k "toeplitz or srgan_step or test_conv2d_fwd_bwd or cyclegan_steps" --durations=5 > $O/pytest.txt 2>&1
  tail -5 $O/pytest.txt
  for cfg in "MIGAN_M2D=0 MIGAN_TOEP_RING=0" "MIGAN_M2D=1 MIGAN_TOEP_RING=1" "MIGAN_M2D=2 MIGAN_TOEP_RING=1"; do
    echo "== $cfg" >> $O/micro.txt
    env $cfg timeout 200 python tools/conv_microbench.py --shapes srgan --match "conv3" --iters 10 --repeat 3 2>&1 | grep "tfwd\|twgrad\|tdgrad\|texpand" >> $O/micro.txt
    env $cfg timeout 200 python tools/conv_microbench.py --shapes cyclegan --match "c7s1-3" --iters 10 --repeat 3 2>&1 | grep "tfwd\|twgrad\|tdgrad\|texpand" >> $O/micro.txt
  done
  for cfg in "MIGAN_M2D=1" "MIGAN_M2D=2"; do
    echo "== $cfg" >> $O/micro.txt
    env $cfg timeout 300 python tools/conv_microbench.py --shapes srgan --match "vgg" --only fwd,dgrad --iters 10 --repeat 3 >> $O/micro.txt 2>&1
    env $cfg timeout 300 python tools/conv_microbench.py --shapes srgan --match "up 64" --only fwd,dgrad --iters 10 --repeat 3 >> $O/micro.txt 2>&1
    env $cfg timeout 300 python tools/conv_microbench.py --shapes srgan --match "res 64" --only fwd,dgrad --iters 10 --repeat 3 >> $O/micro.txt 2>&1
  done
  cat $O/micro.txt
  bl $O/bench.txt srgan 4 MIGAN_M2D=0 MIGAN_TOEP_RING=0
  bl $O/bench.txt srgan 4
  bl $O/bench.txt srgan 4 MIGAN_M2D=2
  bl $O/bench.txt srgan 4 MIGAN_M2D=0 MIGAN_TOEP_RING=0
  bl $O/bench.txt srgan 4
  bl $O/bench.txt srgan 4 MIGAN_M2D=2
  bl $O/bench.txt cyclegan 4 --no-graph
  bl $O/bench.txt cyclegan 4 --no-graph MIGAN_M2D=2
  cat $O/bench.txt
}

task_toep2() {   # call 11: the four-wave strip-walking weight gradient
  local O=gpurun_out/r5j; mkdir -p $O
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py -q -k "toeplitz or srgan_step" --durations=3 > $O/pytest.txt 2>&1
  tail -4 $O/pytest.txt
  for cfg in "MIGAN_TOEP_RING=0" "MIGAN_TOEP_RING=1"; do
    echo "== $cfg" >> $O/micro.txt
    env $cfg timeout 200 python tools/conv_microbench.py --shapes srgan --match "conv3" --dirs twgrad --iters 10 --repeat 3 2>&1 | grep "twgrad" >> $O/micro.txt
    env $cfg timeout 200 python tools/conv_microbench.py --shapes cyclegan --match "c7s1-3" --dirs twgrad --iters 10 --repeat 3 2>&1 | grep "twgrad" >> $O/micro.txt
  done
  cat $O/micro.txt
  ab $O/bench.txt srgan 4 2
  cat $O/bench.txt
}

task_upw() {   # call 12: 64 x 256 weight-gradient tiles of the Upsample+Conv3x3 layers with 64 output channels; ring kernel with fetch distance two
  local O=gpurun_out/r5k; mkdir -p $O
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py -q -k "upconv or toeplitz or dcgan_steps" --durations=3 > $O/pytest.txt 2>&1
  tail -4 $O/pytest.txt
  for bn in 128 256 128 256; do
    echo "== MIGAN_UPW_BN=$bn" >> $O/micro.txt
    MIGAN_UPW_BN=$bn timeout 100 python tools/conv_microbench.py --shapes dcgan --match "G.conv2" --dirs uwgrad --iters 20 --repeat 3 2>&1 | grep uwgrad >> $O/micro.txt
    MIGAN_UPW_BN=$bn timeout 100 python tools/conv_microbench.py --shapes cyclegan --match "u64" --dirs uwgrad --iters 20 --repeat 3 2>&1 | grep uwgrad >> $O/micro.txt
  done
  echo "== ring, fetch distance two" >> $O/micro.txt
  timeout 200 python tools/conv_microbench.py --shapes srgan --match "conv3" --dirs twgrad --iters 10 --repeat 3 2>&1 | grep "twgrad" >> $O/micro.txt
  timeout 200 python tools/conv_microbench.py --shapes cyclegan --match "c7s1-3" --dirs twgrad --iters 10 --repeat 3 2>&1 | grep "twgrad" >> $O/micro.txt
  cat $O/micro.txt
  for r in 1 2; do
    bl $O/bench.txt dcgan 50 MIGAN_UPW_BN=128
    bl $O/bench.txt dcgan 50 MIGAN_UPW_BN=256
  done
  bl $O/bench.txt cyclegan 4 MIGAN_UPW_BN=128 --no-graph
  bl $O/bench.txt cyclegan 4 MIGAN_UPW_BN=256 --no-graph
  cat $O/bench.txt
}

task_bs1() {   # call 14: one image per GPU: 128 x 128 tiles cut along K for the trunk
  local O=gpurun_out/r5l; mkdir -p $O
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py tests/test_fullsize_gpu.py -q -k "reflect or test_conv2d_fwd_bwd or cyclegan_recorded or cyclegan_256_bs1 or splitk" --durations=3 > $O/pytest.txt 2>&1
  tail -4 $O/pytest.txt
  for v in 0 1; do
    echo "== MIGAN_DMA_WIDE_SK=$v" >> $O/micro.txt
    MIGAN_DMA_WIDE_SK=$v timeout 100 python tools/conv_microbench.py --shapes cyclegan --match "R256 bs1" --dirs fwd,rdgrad,wgrad --iters 30 --repeat 3 2>&1 | grep "R256" >> $O/micro.txt
  done
  cat $O/micro.txt
  for r in 1 2; do
    bl $O/bench.txt cyclegan 10 --batch 1 MIGAN_DMA_WIDE_SK=0
    bl $O/bench.txt cyclegan 10 --batch 1
  done
  bl $O/bench.txt cyclegan 10 --batch 1 --no-graph
  bl $O/bench.txt pix2pix 50 MIGAN_DMA_WIDE_SK=0
  bl $O/bench.txt pix2pix 50
  cat $O/bench.txt
}

task_mid() {   # call 15: the default bench line of the tree + kernel traces of the workloads the round worked on
  task_bench
  cp gpurun_out/r5bench/bench_default.json gpurun_out/r5bench/bench_call15.json
  task_prof r5mid srgan cyclegan:graph@1
}

task_final() {   # closing pass: default bench line, kernel traces of all workloads, PMC passes over the steps
  task_bench
  cp gpurun_out/r5bench/bench_default.json gpurun_out/r5bench/bench_final.json
  task_prof r5final dcgan dcgan:graph cyclegan srgan wgan_gp:graph pix2pix:graph cyclegan:graph@1
  task_pmcstep dcgan 3 sq l2 fetch write
  task_pmcstep cyclegan 1 sq fetch write
  task_pmcstep srgan 1 sq fetch write
}

task_thin() {   # call 16: thin-output 3x3 on the MFMA units (dcgan.py:62 forward; input gradients of the image-input layers), its input gradient
  local O=gpurun_out/r5m; mkdir -p $O
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py tests/test_fullsize_gpu.py -q -k "rgb_conv or thin_output or dcgan or srgan_step or srgan_96" --durations=3 > $O/pytest.txt 2>&1
  tail -4 $O/pytest.txt
  timeout 200 python tools/conv_microbench.py --shapes dcgan --match "G.conv3" --iters 30 --repeat 3 > $O/micro.txt 2>&1
  timeout 200 python tools/conv_microbench.py --shapes srgan --match "D.c1" --only dgrad --iters 20 --repeat 3 >> $O/micro.txt 2>&1
  cat $O/micro.txt
  for r in 1 2; do
    bl $O/bench.txt dcgan 50 MIGAN_RGB=0
    bl $O/bench.txt dcgan 50
  done
  ab $O/bench.txt dcgan 50 1
  bl $O/bench.txt dcgan_ch3 50 MIGAN_RGB=0
  bl $O/bench.txt dcgan_ch3 50
  bl $O/bench.txt srgan 4 MIGAN_RGB=0
  bl $O/bench.txt srgan 4
  ab $O/bench.txt srgan 4 1
  cat $O/bench.txt
}

task_ab_all() {   # call 17: the tree against the round-start tree on one box, all workloads; norm statistics pass with eight loads in flight
  local O=gpurun_out/r5n; mkdir -p $O
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py -q -k "norm or thin_output or rgb_conv or dcgan_steps or srgan_step" --durations=3 > $O/pytest.txt 2>&1
  tail -4 $O/pytest.txt
  ab $O/bench.txt dcgan 50 2
  ab $O/bench.txt srgan 4 2
  ab $O/bench.txt cyclegan 4 1
  ab $O/bench.txt pix2pix 50 1
  ab $O/bench.txt dcgan_ch3 50 1
  cat $O/bench.txt
}

task_dcgan_bisect() {   # call 18: which change costs the DCGAN step 2.7 % (call 17)
  local O=gpurun_out/r5o; mkdir -p $O
  for r in 1 2; do
    (cd ab_base && echo "== base dcgan" >> $R/$O/bench.txt && timeout 300 python bench.py --workload dcgan --steps 50 --warmup 2 --no-cpu-baseline --no-extra --no-roofline 2>>$R/$O/bench.txt.err | line >> $R/$O/bench.txt)
    bl $O/bench.txt dcgan 50 MIGAN_RGB=0
    bl $O/bench.txt dcgan 50
  done
  cat $O/bench.txt
}

task_stats() {   # call 21: norm statistics from the conv epilogue of the LDS-DMA kernels (MIGAN_CONV_STATS=1) against the statistics pass
  local O=gpurun_out/r5p; mkdir -p $O
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "conv_epilogue_statistics or test_conv2d_fwd_bwd or upconv" --durations=3 > $O/pytest.txt 2>&1
  tail -3 $O/pytest.txt
  for r in 1 2; do
    bl $O/bench.txt dcgan 50 MIGAN_CONV_STATS=0
    bl $O/bench.txt dcgan 50 MIGAN_CONV_STATS=1
  done
  for r in 1 2; do
    bl $O/bench.txt srgan 4 MIGAN_CONV_STATS=0
    bl $O/bench.txt srgan 4 MIGAN_CONV_STATS=1
  done
  bl $O/bench.txt cyclegan 4 MIGAN_CONV_STATS=0
  bl $O/bench.txt cyclegan 4 MIGAN_CONV_STATS=1
  bl $O/bench.txt pix2pix 50 MIGAN_CONV_STATS=0
  bl $O/bench.txt pix2pix 50 MIGAN_CONV_STATS=1
  cat $O/bench.txt
}

task_rgb3() {   # call 22: image-input forward with a straight-line epilogue (no wait in front of every store)
  local O=gpurun_out/r5q; mkdir -p $O
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py -q -k "rgb_conv or thin_output or srgan_step" --durations=3 > $O/pytest.txt 2>&1
  tail -3 $O/pytest.txt
  timeout 200 python tools/conv_microbench.py --shapes srgan --match "3->64" --dirs fwd,xfwd,xwgrad --iters 20 --repeat 3 > $O/micro.txt 2>&1
  timeout 200 python tools/conv_microbench.py --shapes cyclegan --match "c7s1-64" --dirs fwd,xfwd,xwgrad --iters 20 --repeat 3 >> $O/micro.txt 2>&1
  cat $O/micro.txt
  ab $O/bench.txt srgan 4 2
  bl $O/bench.txt cyclegan 4 --no-graph
  cat $O/bench.txt
}

task_epi() {   # call 23: igemm_dma_kernel epilogue without a wait in front of every store (bias / decode parameters in registers)
  local O=gpurun_out/r5r; mkdir -p $O
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "conv2d_fwd_bwd or upconv or conv_transpose or reflect or splitk or strided" --durations=3 > $O/pytest.txt 2>&1
  tail -3 $O/pytest.txt
  for tree in ab_base . ab_base .; do
    echo "== tree $tree" >> $O/micro.txt
    (cd $tree && for sh in dcgan srgan cyclegan; do timeout 200 python tools/conv_microbench.py --shapes $sh --only fwd,dgrad --iters 20 --repeat 3 2>&1 | grep -v "^#" ; done) >> $O/micro.txt 2>&1
  done
  cat $O/micro.txt
  ab $O/bench.txt dcgan 50 2
  ab $O/bench.txt srgan 4 2
  ab $O/bench.txt cyclegan 4 1
  ab $O/bench.txt pix2pix 50 1
  bl $O/bench.txt cyclegan 10 --batch 1
  cat $O/bench.txt
}

task_tile192() {   # call 24: 192x64 tiles for launches of <= 3 such tiles per CU (SRGAN trunk), norm statistics pass cut into more blocks
  local O=gpurun_out/r5s; mkdir -p $O
  timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "conv2d_fwd_bwd or geometry_selects" --durations=3 > $O/pytest.txt 2>&1
  tail -3 $O/pytest.txt
  for v in 0 1 0 1; do
    echo "== MIGAN_DMA_192=$v" >> $O/micro.txt
    MIGAN_DMA_192=$v timeout 100 python tools/conv_microbench.py --shapes srgan --match "res 64" --dirs fwd,dgrad --iters 30 --repeat 3 2>&1 | grep "res 64" >> $O/micro.txt
  done
  cat $O/micro.txt
  bl $O/bench.txt srgan 4 MIGAN_DMA_192=0
  bl $O/bench.txt srgan 4 MIGAN_DMA_192=1
  bl $O/bench.txt srgan 4 MIGAN_DMA_192=1 MIGAN_NORM_BLOCKS=2048
  bl $O/bench.txt srgan 4 MIGAN_DMA_192=1 MIGAN_NORM_BLOCKS=4096
  bl $O/bench.txt srgan 4 MIGAN_DMA_192=0
  bl $O/bench.txt srgan 4 MIGAN_DMA_192=1
  bl $O/bench.txt cyclegan 4 MIGAN_NORM_BLOCKS=1024
  bl $O/bench.txt cyclegan 4 MIGAN_NORM_BLOCKS=2048
  bl $O/bench.txt dcgan 50 MIGAN_NORM_BLOCKS=1024
  bl $O/bench.txt dcgan 50 MIGAN_NORM_BLOCKS=2048
  cat $O/bench.txt
}

task_cap() {   # call 25: workgroups per CU capped below the kernel's occupancy (unused dynamic LDS) so that the last round of a launch is not one lone workgroup per CU
  local O=gpurun_out/r5t; mkdir -p $O
  for v in 0 3 5 6 -1; do
    echo "== MIGAN_DMA_CAP=$v" >> $O/micro.txt
    MIGAN_DMA_CAP=$v timeout 100 python tools/conv_microbench.py --shapes srgan --only fwd,dgrad --iters 20 --repeat 3 2>&1 | grep " us " | grep -v "9x9\|3->64" >> $O/micro.txt
  done
  for v in 0 -1; do
    echo "== MIGAN_DMA_CAP=$v" >> $O/micro2.txt
    for sh in dcgan cyclegan; do MIGAN_DMA_CAP=$v timeout 100 python tools/conv_microbench.py --shapes $sh --only fwd,dgrad --iters 20 --repeat 3 2>&1 | grep " us " >> $O/micro2.txt; done
  done
  cat $O/micro.txt $O/micro2.txt
  for r in 1 2; do
    bl $O/bench.txt srgan 4 MIGAN_DMA_CAP=0
    bl $O/bench.txt srgan 4 MIGAN_DMA_CAP=-1
  done
  bl $O/bench.txt cyclegan 4 MIGAN_DMA_CAP=0
  bl $O/bench.txt cyclegan 4 MIGAN_DMA_CAP=-1
  bl $O/bench.txt dcgan 50 MIGAN_DMA_CAP=0
  bl $O/bench.txt dcgan 50 MIGAN_DMA_CAP=-1
  cat $O/bench.txt
}

task_probe3() {   # call 26 (probe, kernel hack not committed): what a launch of igemm_dma_kernel costs besides its K loop
  local O=gpurun_out/r5u; mkdir -p $O
  for v in 0 -1001 -1002 -1003; do
    echo "== MIGAN_MB_SLOPE=$v (0 normal, -1001 no epilogue, -1002 two K-tiles, -1003 two K-tiles and no epilogue)" >> $O/micro.txt
    MIGAN_MB_SLOPE=$v timeout 100 python tools/conv_microbench.py --shapes srgan --dirs fwd --iters 20 --repeat 3 2>&1 | grep " us " | grep -v "9x9\|3->64" >> $O/micro.txt
    MIGAN_MB_SLOPE=$v timeout 100 python tools/conv_microbench.py --shapes cyclegan --dirs fwd --match "R256" --iters 20 --repeat 3 2>&1 | grep " us " >> $O/micro.txt
  done
  cat $O/micro.txt
}

task_probe4() {   # call 27 (probe, kernel hack not committed): half of the first-round workgroups start late - do the rounds of a launch run in lockstep?
  local O=gpurun_out/r5v; mkdir -p $O
  for v in 0 -2010 -2020 -2040 -2070 -2100; do
    echo "== MIGAN_MB_SLOPE=$v (0 normal; -2000 - d: odd half of the first round starts d us late)" >> $O/micro.txt
    MIGAN_MB_SLOPE=$v timeout 100 python tools/conv_microbench.py --shapes srgan --dirs fwd --iters 20 --repeat 3 2>&1 | grep " us " | grep -v "9x9\|3->64" >> $O/micro.txt
    MIGAN_MB_SLOPE=$v timeout 100 python tools/conv_microbench.py --shapes cyclegan --dirs fwd --match "R256" --iters 20 --repeat 3 2>&1 | grep " us " >> $O/micro.txt
  done
  cat $O/micro.txt
}

# whole-step A/B on ONE box against the tree of the previous commit (ab_prev/ = git archive HEAD + built library):  abp <outfile> <workload> <steps> [reps]
abp() {
  local out=$1 w=$2 k=$3 reps=${4:-2}
  for r in $(seq $reps); do
    (cd ab_prev && echo "== prev $w" >> $R/$out && timeout 300 python bench.py --workload $w --steps $k --warmup 2 --no-cpu-baseline --no-extra --no-roofline 2>>$R/$out.err | line >> $R/$out)
    bl $out $w $k
  done
}

task_epi2() {   # call 28: igemm_dma_kernel epilogue with channel quads per lane (weights as the MFMA row operand): 16 dwordx4 stores instead of 64 dword stores per lane
  local O=gpurun_out/r5w; mkdir -p $O
  timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py -q -x -k "conv2d_fwd_bwd or upconv or conv_transpose or reflect or splitk or strided or dropout or relu_backward or toeplitz or geometry_selects or dcgan_step or srgan_step" --durations=3 > $O/pytest.txt 2>&1
  tail -3 $O/pytest.txt
  for tree in ab_prev . ab_prev .; do
    echo "== tree $tree" >> $O/micro.txt
    (cd $tree && for sh in dcgan srgan cyclegan; do timeout 200 python tools/conv_microbench.py --shapes $sh --only fwd,dgrad --iters 20 --repeat 3 2>&1 | grep " us " ; done) >> $O/micro.txt 2>&1
  done
  abp $O/bench.txt dcgan 50 2
  abp $O/bench.txt srgan 4 2
  abp $O/bench.txt cyclegan 4 1
  abp $O/bench.txt pix2pix 50 1
  cat $O/bench.txt
}

task_probe5() {   # call 29 (probe, not committed): cache policy of the output stores of igemm_dma_kernel (0 global, 1 buffer, 2 nt, 3 sc1, 4 sc1 nt)
  local O=gpurun_out/r5x; mkdir -p $O
  for v in 0 1 2 3 4; do
    echo "== MIGAN_ST_MODE=$v" >> $O/micro.txt
    MIGAN_ST_MODE=$v timeout 100 python tools/conv_microbench.py --shapes srgan --only fwd,dgrad --iters 20 --repeat 3 2>&1 | grep " us " | grep -v "9x9\|3->64" >> $O/micro.txt
    MIGAN_ST_MODE=$v timeout 100 python tools/conv_microbench.py --shapes dcgan --dirs ufwd,udgrad --iters 20 --repeat 3 2>&1 | grep " us " >> $O/micro.txt
  done
  for v in 0 2 3 4; do bl $O/bench.txt srgan 4 MIGAN_ST_MODE=$v; done
  for v in 0 3; do bl $O/bench.txt dcgan 50 MIGAN_ST_MODE=$v; done
  cat $O/bench.txt
}

task_final2() {   # call 31, closing pass of the final tree with 3.9 GPU-minutes left: kernel traces of the DCGAN and SRGAN steps, then the default bench line (PMC tables stay those of call 20)
  task_prof r5final2 dcgan dcgan:graph srgan
  task_bench
  cp gpurun_out/r5bench/bench_default.json gpurun_out/r5bench/bench_final2.json
}

t=${1:-}; shift || true
case "$t" in
  first) task_first "$@" ;;
  prof) task_prof "$@" ;;
  pmcstep) task_pmcstep "$@" ;;
  bench) task_bench "$@" ;;
  suite) task_suite "$@" ;;
  *) if declare -F "task_$t" > /dev/null; then "task_$t" "$@"; else echo "usage: gpu_r05.sh <task> [args]"; exit 2; fi ;;
esac
