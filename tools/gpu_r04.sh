#!/bin/bash
# Round-4 gpurun calls:  gpurun --timeout T -- "bash tools/gpu_r04.sh <task> [args]"
# Tasks write under gpurun_out/<dir>/ (scratch); what DESIGN.md quotes is copied into profiles/r04_*.
set -u
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 120 python tools/gpu_sanity.py || { echo "bad box, giving up"; exit 3; }

# two gloo ranks on ONE GPU, cyclegan --global-batch 2 (tests/test_steps_gpu.py::test_bench_cyclegan_strong_scaling_two_ranks_on_one_gpu):
#   two_rank <tag> <limit_s> <dump_s> [ENV=VAL ...]   -> gpurun_out/hang/<tag>.{out,err,rc}
two_rank() { python tools/two_rank.py "$@"; }

task_hang() {
  rm -f /tmp/migan_selfcheck_*
  # A: no self-check, default kernels
  two_rank A_noselfcheck 100 50 MIGAN_SELFCHECK=0
  rcA=$?
  # B: self-check on, cold verdict cache (what a run of this test ALONE sees)
  rm -f /tmp/migan_selfcheck_*
  two_rank B_selfcheck_cold 200 120
  # C: self-check on, cached verdict (what the test sees inside the suite)
  two_rank C_selfcheck_cached 100 50
  if [ $rcA -ne 0 ]; then
    two_rank D_noselfcheck_regstage 100 50 MIGAN_SELFCHECK=0 MIGAN_DMA=0 MIGAN_DMA_WGRAD=0
    two_rank E_noselfcheck_nostaged 100 50 MIGAN_SELFCHECK=0 MIGAN_THIN_WAVE=0 MIGAN_WGRAD_REDUCE_TR=0 MIGAN_PACK_TR=0 MIGAN_MIDK=0 MIGAN_NORM_SMALL=0 MIGAN_SMALLK_PB16=0 MIGAN_DROPOUT_FUSE=0 MIGAN_FEWPIX=0
  fi
  cat gpurun_out/hang/*.rc
  for f in gpurun_out/hang/*.err; do echo "== $f"; grep -v "^/opt\|Warning\|warn" $f | tail -60; done
}

# rocprofv3 --kernel-trace --stats of bench steps:  prof <outdir> <workload>[:graph] ...
task_prof() {
  local O=gpurun_out/${1:-r4prof}; shift
  mkdir -p $O
  for spec in "$@"; do
    # eager traces: one stream (--no-overlap), so a kernel's duration is its own; graph traces: the step as it is timed
    w=${spec%%:*}; mode=eager; flag="--no-graph --no-overlap"; [ "$spec" != "$w" ] && { mode=graph; flag=; }
    k=3; [ $w = dcgan ] && k=20; [ $w = pix2pix ] && k=20; [ $w = wgan_gp ] && k=50
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_${w}_$mode -o $w -- python $R/bench.py --workload $w --steps $k --warmup 2 \
       --min-seconds 0 $flag --no-roofline --no-cpu-baseline --no-extra > $R/$O/prof_${w}_$mode.log 2>&1)
    db=$(ls $O/prof_${w}_$mode/*/${w}_results.db $O/prof_${w}_$mode/${w}_results.db 2>/dev/null | head -1)
    a=2; [ $w = cyclegan ] && a=3; [ $w = wgan_gp ] && a=1.2
    python tools/rocpd_stats.py $db 150 --by-grid --per-step adam_kernel=$a > $O/${w}_${mode}_kernel_stats.txt 2>&1
    head -4 $O/${w}_${mode}_kernel_stats.txt
    rm -rf $O/prof_${w}_$mode   # the database is large; the table is what is kept
  done
}

# PMC passes (separate, --kernel-trace only) over the eager step of a workload and over the microbench of the same layers
#   pmc <workload> <microbench-shapes> <match> <dirs>
task_pmc() {
  local w=$1 shapes=$2 match=$3 dirs=$4
  local O=gpurun_out/r4pmc_$w
  mkdir -p $O
  local k=6; [ $w = cyclegan ] && k=2; [ $w = srgan ] && k=2
  BENCH="python $R/bench.py --workload $w --steps $k --warmup 2 --min-seconds 0 --no-graph --no-roofline --no-cpu-baseline --no-extra"
  MICRO="python $R/tools/conv_microbench.py --shapes $shapes --match $match --dirs $dirs --iters 6"
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
  for tgt in bench micro; do
    python tools/pmc_summary.py $O/${tgt}_sq $O/${tgt}_l2 $O/${tgt}_fetch $O/${tgt}_write > $O/${tgt}_summary.txt 2>&1
  done
  # keep the per-kernel summaries; drop the raw CSVs beyond what the 64 MiB merge limit allows
  find $O -name "*.csv" -size +8M -delete
  head -40 $O/bench_summary.txt
}

# the default bench line (what the driver runs)
task_bench() {
  local O=gpurun_out/r4bench; mkdir -p $O
  timeout 600 python bench.py "$@" > $O/bench_default.json 2> $O/bench_default.err
  echo "bench rc=$?"; cut -c1-1500 $O/bench_default.json
}

task_suite() {
  local O=gpurun_out/r4suite; mkdir -p $O
  timeout 1300 python -m pytest tests -m gpu -q --timeout=700 --durations=25 -rxs > $O/pytest_gpu.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest_gpu.txt
  grep -v "^  \|^$" $O/pytest_gpu.txt | tail -60
}

# PMC passes over the training step itself: bench.py --pmc-log (launch ordinals per roofline group) joined by tools/pmc_step.py
#   pmcstep <workload> <steps> <pass>...      pass = sq | l2 | fetch | write
task_pmcstep() {
  local w=$1 k=$2; shift 2
  local O=gpurun_out/r4pmcstep_$w
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
  python tools/pmc_step.py r04 $(for n in "$@"; do echo $O/$n; done) > $O/summary.txt 2>&1
  head -30 $O/summary.txt | cut -c1-200
  find $O -name "*.csv" -size +6M -delete
  cp profiles/r04_pmc_kernels.json gpurun_out/r04_pmc_kernels.json
}

task_second() {
  local O=gpurun_out/r4b; mkdir -p $O
  timeout 120 ./tools/abi_check.bin floor > $O/abi_check.txt 2>&1
  for s in critic mlp norm; do timeout 120 ./tools/abi_check.bin $s >> $O/abi_check.txt 2>&1; done
  grep -v "^migan" $O/abi_check.txt | cut -c1-220
  # item 1: the two-rank CycleGAN launch with the collective-free buffer warm-up, as the suite runs it
  timeout 300 python -m pytest tests/test_steps_gpu.py -q -x -k "two_ranks_on_one_gpu" --durations=5 > $O/pytest_two_rank.txt 2>&1
  tail -8 $O/pytest_two_rank.txt
  port=$((20000 + RANDOM % 20000))
  MIGAN_DP_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
     tools/gloo_bucket_probe.py > $O/gloo_probe.txt 2>&1
  grep "gloo all_reduce" $O/gloo_probe.txt
  # which kernel ran (launch counters) + parity of the cases that used to sit behind run-time switches
  timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py -q -x -k "geometry_selects or fewpix or fused_dropout or fused_wgan or wgan_gp_steps" > $O/pytest_counters.txt 2>&1
  tail -4 $O/pytest_counters.txt
  task_pmcstep dcgan 3 sq l2 fetch write
  task_pmcstep cyclegan 1 sq fetch write
  task_pmcstep srgan 1 sq fetch write
}

task_third() {
  local O=gpurun_out/r4c; mkdir -p $O
  for s in floor critic mlp; do timeout 120 ./tools/abi_check.bin $s >> $O/abi_check.txt 2>&1; done
  grep -v "^migan" $O/abi_check.txt | cut -c1-220
  timeout 900 python -m pytest tests/test_steps_gpu.py tests/test_ops_gpu.py tests/test_fullsize_gpu.py -q -x -k "wgan or adam or splitk or upconv or bias_grad or critic" --durations=5 > $O/pytest.txt 2>&1
  tail -5 $O/pytest.txt
  for k7 in 1 0; do
    echo "== wgan_gp MIGAN_K7=$k7" >> $O/bench.txt
    MIGAN_K7=$k7 timeout 300 python bench.py --workload wgan_gp --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-roofline 2>>$O/bench.err | cut -c1-400 >> $O/bench.txt
  done
  cat $O/bench.txt
  task_prof r4c wgan_gp:graph
  timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
  echo "bench rc=$?"; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r4c/bench_default.json').read().strip().splitlines()[-1])
print({k:r[k] for k in ('value','ms_per_step')})
rf=r['roofline']; print({k:rf.get(k) for k in ('kernel','frac','traffic','mfma_busy_frac','symbol','avg_launch_ms')})
for k,v in r['extra'].items(): print(k, {a:v.get(a) for a in ('images_per_s','ms_per_step','error')})
PY
}

task_fourth() {
  local O=gpurun_out/r4d; mkdir -p $O
  for s in critic mlp; do timeout 120 ./tools/abi_check.bin $s >> $O/abi_check.txt 2>&1; done
  grep -v "^migan" $O/abi_check.txt | cut -c1-220
  timeout 900 python -m pytest tests/test_steps_gpu.py tests/test_ops_gpu.py tests/test_fullsize_gpu.py -q -x -k "wgan or adam or splitk or critic" --durations=5 > $O/pytest.txt 2>&1
  tail -4 $O/pytest.txt
  echo "== wgan_gp" >> $O/bench.txt
  timeout 300 python bench.py --workload wgan_gp --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-roofline 2>>$O/bench.err | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'])" >> $O/bench.txt
  task_prof r4d wgan_gp:graph
  for w in dcgan cyclegan srgan; do
    k=4; [ $w = dcgan ] && k=50
    for ns in 2 3 2 3; do
      echo "== $w MIGAN_DMA_NS=$ns" >> $O/bench.txt
      MIGAN_DMA_NS=$ns timeout 300 python bench.py --workload $w --steps $k --warmup 2 --no-cpu-baseline --no-extra --no-roofline 2>>$O/bench.err | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['timing']['ms_per_step_min'])" >> $O/bench.txt
    done
  done
  cat $O/bench.txt
}

task_fifth() {
  local O=gpurun_out/r4e; mkdir -p $O
  for s in critic mlp; do timeout 120 ./tools/abi_check.bin $s >> $O/abi_check.txt 2>&1; done
  grep -v "^migan" $O/abi_check.txt | cut -c1-220
  timeout 900 python -m pytest tests/test_steps_gpu.py tests/test_fullsize_gpu.py -q -x -k "wgan or critic" --durations=5 > $O/pytest.txt 2>&1
  tail -4 $O/pytest.txt
  echo "== wgan_gp" >> $O/bench.txt
  timeout 300 python bench.py --workload wgan_gp --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-roofline 2>>$O/bench.err | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'])" >> $O/bench.txt
  cat $O/bench.txt
  task_prof r4e wgan_gp:graph
}

# bench one workload:  bl <outfile> <workload> <steps> [ENV=VAL ...]  -> "img/s ms/step min-block"
bl() {
  local out=$1 w=$2 k=$3; shift 3
  echo "== $w $*" >> $out
  env "$@" timeout 300 python bench.py --workload $w --steps $k --warmup 2 --no-cpu-baseline --no-extra --no-roofline 2>>$out.err | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['timing']['ms_per_step_min'])" >> $out
}

# whole-step A/B on ONE box against the tree of the round's start (ab_base/ = git archive c76fc2b + its built library; git-ignored,
# travels with the snapshot):  ab <outfile> <workload> <steps> [reps]   -> alternating base / head lines
ab() {
  local out=$1 w=$2 k=$3 reps=${4:-2}
  for r in $(seq $reps); do
    (cd ab_base && echo "== base $w" >> $R/$out && timeout 300 python bench.py --workload $w --steps $k --warmup 2 --no-cpu-baseline --no-extra --no-roofline 2>>$R/$out.err | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['timing']['ms_per_step_min'])" >> $R/$out)
    bl $out $w $k
  done
}

task_fifteenth() {   # priority of the discriminator half's stream
  local O=gpurun_out/r4o; mkdir -p $O
  for w in dcgan pix2pix srgan cyclegan; do
    k=4; [ $w = dcgan ] && k=50; [ $w = pix2pix ] && k=50
    for pr in 0 -1 0 -1; do
      bl $O/bench.txt $w $k MIGAN_D_PRIORITY=$pr
    done
  done
  cat $O/bench.txt
}

task_timeline() {   # kernel timeline of the captured DCGAN step (which stream is the critical path)
  local O=gpurun_out/r4p; mkdir -p $O
  for mode in overlap nooverlap; do
    flag=; [ $mode = nooverlap ] && flag=--no-overlap
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/$O/tl_$mode -o dcgan -- python $R/bench.py --workload dcgan --steps 20 --warmup 2 --min-seconds 0 \
       --no-roofline --no-cpu-baseline --no-extra $flag > $R/$O/tl_$mode.log 2>&1)
    db=$(ls $O/tl_$mode/*/dcgan_results.db $O/tl_$mode/dcgan_results.db 2>/dev/null | head -1)
    python tools/rocpd_timeline.py $db 420 > $O/timeline_$mode.csv 2>&1
    rm -rf $O/tl_$mode
    head -3 $O/timeline_$mode.csv | cut -c1-200
  done
}

task_sixteenth() {   # split-K reach for the under-filled DCGAN discriminator convs
  local O=gpurun_out/r4q; mkdir -p $O
  for cfg in "0 0" "256 8" "512 8" "256 4" "0 0" "256 8"; do
    set -- $cfg
    echo "== MIGAN_DMA_SK_T=$1 MIGAN_DMA_SK_KT=$2" >> $O/micro.txt
    MIGAN_DMA_SK_T=$1 MIGAN_DMA_SK_KT=$2 timeout 200 python tools/conv_microbench.py --shapes dcgan --match "D" --only fwd,dgrad --iters 20 --repeat 3 2>&1 | grep "conv3\|conv4" >> $O/micro.txt
  done
  cat $O/micro.txt
  for cfg in "0 0" "256 8" "0 0" "256 8"; do
    set -- $cfg
    bl $O/bench.txt dcgan 50 MIGAN_DMA_SK_T=$1 MIGAN_DMA_SK_KT=$2
  done
  cat $O/bench.txt
}

task_seventeenth() {   # cyclegan: the two halves of the generators' forward / backward on two streams
  local O=gpurun_out/r4r; mkdir -p $O
  timeout 900 python -m pytest tests/test_steps_gpu.py tests/test_fullsize_gpu.py tests/test_models_gpu.py -q -x -k "cyclegan or second_stream" --durations=3 > $O/pytest.txt 2>&1
  tail -4 $O/pytest.txt
  for r in 1 2; do
    bl $O/bench.txt cyclegan 4 MIGAN_CHAINS=0
    bl $O/bench.txt cyclegan 4 MIGAN_CHAINS=1
  done
  cat $O/bench.txt
}

task_syncbn() {   # call 32: the recorded step under cross-replica BatchNorm (segments cut at the collectives), two ranks on the one GPU
  local O=gpurun_out/r4t; mkdir -p $O
  timeout 240 python -m pytest tests/test_dp_gpu.py -q -x -k "cross_replica" --durations=3 > $O/pytest.txt 2>&1
  tail -25 $O/pytest.txt
}

task_syncbn_bench() {   # call 33: two ranks on the one GPU (gloo, host-staged collectives): DCGAN --sync-bn, segments vs eager launches
  local O=gpurun_out/r4t; mkdir -p $O
  export MIGAN_DP_BACKEND=gloo MIGAN_DP_SINGLE_DEVICE=1
  for v in graph nograph; do
    local f=""; [ $v = nograph ] && f="--no-graph"
    timeout 110 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 \
      --sync-bn --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-extra $f > $O/syncbn_$v.json 2> $O/syncbn_$v.err
    tail -c 1500 $O/syncbn_$v.json; tail -3 $O/syncbn_$v.err
  done
}

task_reduce_ab() {   # call 34: slab / chunk reductions with all of a round's loads in flight (ab_base/ = the tree one commit earlier)
  local O=gpurun_out/r4u; mkdir -p $O
  timeout 130 python -m pytest tests/test_ops_gpu.py tests/test_steps_gpu.py -q -x -k "upconv or conv2d_fwd_bwd or batchnorm or instancenorm or norm_bwd_column or colsum or bias_grad or norm_double or epilogue_statistics or conv_transpose2d or dcgan_steps or splitk or direct_grad" > $O/pytest.txt 2>&1
  tail -3 $O/pytest.txt
  for tree in ab_base .; do
    echo "== tree $tree" >> $O/micro.txt
    (cd $tree && timeout 60 python tools/conv_microbench.py --shapes dcgan --match "G.conv" --dirs uwgrad --iters 20 --repeat 3 2>&1 | grep "G.conv") >> $O/micro.txt
  done
  cat $O/micro.txt
  ab $O/bench.txt dcgan 50 2
  cat $O/bench.txt
}

task_closing() {   # the round's closing call: parity of what changed after the full suite of call 19, then the default bench line
  local O=gpurun_out/r4s; mkdir -p $O
  timeout 600 python -m pytest tests/test_steps_gpu.py tests/test_fullsize_gpu.py -q -x -k "second_stream or srgan or pix2pix_step or dcgan_steps or two_ranks" --durations=3 > $O/pytest.txt 2>&1
  tail -4 $O/pytest.txt
  task_bench
  cp gpurun_out/r4bench/bench_default.json $O/bench_closing.json
}

task_final() {   # the round's last measurement pass on the final tree: default bench line, kernel traces, PMC passes over the steps
  task_bench
  cp gpurun_out/r4bench/bench_default.json gpurun_out/r4bench/bench_final.json
  task_prof r4final dcgan dcgan:graph cyclegan srgan wgan_gp:graph pix2pix:graph
  task_pmcstep dcgan 3 sq l2 fetch write
  task_pmcstep cyclegan 1 sq fetch write
  task_pmcstep srgan 1 sq fetch write
}

task_thirteenth() {   # srgan: the frozen VGG passes on a third stream
  local O=gpurun_out/r4m; mkdir -p $O
  timeout 600 python -m pytest tests/test_steps_gpu.py tests/test_fullsize_gpu.py -q -x -k "srgan or second_stream" --durations=3 > $O/pytest.txt 2>&1
  tail -3 $O/pytest.txt
  for r in 1 2; do
    echo "== srgan --no-overlap" >> $O/bench.txt
    timeout 300 python bench.py --workload srgan --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-roofline --no-overlap 2>>$O/bench.txt.err | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['timing']['ms_per_step_min'])" >> $O/bench.txt
    bl $O/bench.txt srgan 4
  done
  ab $O/bench.txt srgan 4 1
  cat $O/bench.txt
}

task_twelfth() {   # weight-gradient stream only for gradients of >= 6 M elements
  local O=gpurun_out/r4l; mkdir -p $O
  timeout 600 python -m pytest tests/test_steps_gpu.py -q -x -k "second_stream" > $O/pytest.txt 2>&1
  tail -2 $O/pytest.txt
  for w in dcgan srgan cyclegan pix2pix; do
    k=4; [ $w = dcgan ] && k=50; [ $w = pix2pix ] && k=50
    echo "== $w --no-overlap" >> $O/bench.txt
    timeout 300 python bench.py --workload $w --steps $k --warmup 2 --no-cpu-baseline --no-extra --no-roofline --no-overlap 2>>$O/bench.txt.err | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['timing']['ms_per_step_min'])" >> $O/bench.txt
    bl $O/bench.txt $w $k
  done
  cat $O/bench.txt
}

task_eleventh() {   # weight gradients on their own stream (deferred join), batched norm statistics loads
  local O=gpurun_out/r4k; mkdir -p $O
  timeout 900 python -m pytest tests/test_steps_gpu.py tests/test_dp_gpu.py tests/test_ops_gpu.py -q -x -k "second_stream or dcgan or cyclegan_steps or srgan_step or pix2pix_step or two_ranks or world1 or norm or wgan_gp_steps or bias_grad" --durations=5 > $O/pytest.txt 2>&1
  tail -5 $O/pytest.txt
  for w in dcgan srgan cyclegan; do
    k=4; [ $w = dcgan ] && k=50
    bl $O/bench.txt $w $k
    echo "== $w --no-overlap" >> $O/bench.txt
    timeout 300 python bench.py --workload $w --steps $k --warmup 2 --no-cpu-baseline --no-extra --no-roofline --no-overlap 2>>$O/bench.txt.err | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['timing']['ms_per_step_min'])" >> $O/bench.txt
    bl $O/bench.txt $w $k
  done
  ab $O/bench.txt pix2pix 50 1
  cat $O/bench.txt
}

task_ninth() {   # the discriminator halves of srgan / cyclegan / pix2pix on the second stream; data-parallel path of the dcgan overlap
  local O=gpurun_out/r4i; mkdir -p $O
  timeout 900 python -m pytest tests/test_steps_gpu.py tests/test_dp_gpu.py -q -x -k "second_stream or cyclegan_steps or srgan_step or pix2pix_step or two_ranks or dcgan_graph or world1" --durations=5 > $O/pytest.txt 2>&1
  tail -5 $O/pytest.txt
  ab $O/bench.txt cyclegan 4 2
  ab $O/bench.txt srgan 4 2
  ab $O/bench.txt pix2pix 50 1
  ab $O/bench.txt dcgan 50 1
  cat $O/bench.txt
}

task_eighth() {   # dcgan: discriminator half on a second stream underneath the generator's backward
  local O=gpurun_out/r4h; mkdir -p $O
  timeout 600 python -m pytest tests/test_steps_gpu.py tests/test_fullsize_gpu.py -q -x -k "dcgan or acgan or clone" --durations=3 > $O/pytest.txt 2>&1
  tail -3 $O/pytest.txt
  ab $O/bench.txt dcgan 50 2
  bl $O/bench.txt dcgan 50 MIGAN_PAIR_D=0
  bl $O/bench.txt dcgan_ch3 50
  cat $O/bench.txt
}

task_seventh() {   # MLP layer 0 inside layer 1's launch; tiles-per-workgroup sweep on the SRGAN trunk; longest-class-first dgrad vs the round-start tree
  local O=gpurun_out/r4g; mkdir -p $O
  timeout 120 ./tools/abi_check.bin mlp > $O/abi_check.txt 2>&1
  grep -v "^migan" $O/abi_check.txt | cut -c1-260
  timeout 600 python -m pytest tests/test_steps_gpu.py tests/test_fullsize_gpu.py -q -x -k "wgan" --durations=3 > $O/pytest.txt 2>&1
  tail -3 $O/pytest.txt
  bl $O/bench.txt wgan_gp 200
  for tpw in 0 3 2 0 3; do
    echo "== res 64->64 @96 MIGAN_DMA_TPW=$tpw" >> $O/micro.txt
    MIGAN_DMA_TPW=$tpw timeout 200 python tools/conv_microbench.py --shapes srgan --match "res 64" --only fwd,dgrad --iters 20 --repeat 3 2>&1 | grep "res 64" >> $O/micro.txt
  done
  echo "== strided dgrad, round-start tree (classes in parity order)" >> $O/micro.txt
  (cd ab_base && timeout 200 python tools/conv_microbench.py --shapes srgan --match " s2" --only dgrad --iters 10 --repeat 3 2>&1 | grep " s2" >> $R/$O/micro.txt)
  (cd ab_base && timeout 200 python tools/conv_microbench.py --shapes cyclegan --match "d128\|d256" --only dgrad --iters 10 --repeat 3 2>&1 | grep " s2" >> $R/$O/micro.txt)
  echo "== strided dgrad, HEAD (longest class first)" >> $O/micro.txt
  timeout 200 python tools/conv_microbench.py --shapes srgan --match " s2" --only dgrad --iters 10 --repeat 3 2>&1 | grep " s2" >> $O/micro.txt
  timeout 200 python tools/conv_microbench.py --shapes cyclegan --match "d128\|d256" --only dgrad --iters 10 --repeat 3 2>&1 | grep " s2" >> $O/micro.txt
  cat $O/micro.txt
  ab $O/bench.txt dcgan 50 2
  ab $O/bench.txt srgan 4 1
  ab $O/bench.txt cyclegan 4 1
  cat $O/bench.txt
}

task_sixth() {   # MLP backward on row-group workgroups, ReLU hand-off (SRGAN), longest-class-first strided dgrad
  local O=gpurun_out/r4f; mkdir -p $O
  timeout 120 ./tools/abi_check.bin mlp > $O/abi_check.txt 2>&1
  grep -v "^migan" $O/abi_check.txt | cut -c1-260
  timeout 900 python -m pytest tests/test_steps_gpu.py tests/test_ops_gpu.py tests/test_fullsize_gpu.py -q -x -k "wgan or relu_backward or srgan or maxpool or conv2d_strided or test_conv2d" --durations=5 > $O/pytest.txt 2>&1
  tail -4 $O/pytest.txt
  bl $O/bench.txt wgan_gp 200
  bl $O/bench.txt srgan 4
  bl $O/bench.txt cyclegan 4
  bl $O/bench.txt dcgan 50
  cat $O/bench.txt
  task_prof r4f wgan_gp:graph srgan
}

t=${1:-}; shift || true
case "$t" in
  closing) task_closing "$@" ;;
  syncbn) task_syncbn "$@" ;;
  reduce_ab) task_reduce_ab "$@" ;;
  syncbn_bench) task_syncbn_bench "$@" ;;
  seventeenth) task_seventeenth "$@" ;;
  sixteenth) task_sixteenth "$@" ;;
  timeline) task_timeline "$@" ;;
  fifteenth) task_fifteenth "$@" ;;
  final) task_final "$@" ;;
  thirteenth) task_thirteenth "$@" ;;
  twelfth) task_twelfth "$@" ;;
  eleventh) task_eleventh "$@" ;;
  ninth) task_ninth "$@" ;;
  eighth) task_eighth "$@" ;;
  seventh) task_seventh "$@" ;;
  sixth) task_sixth "$@" ;;
  fifth) task_fifth "$@" ;;
  fourth) task_fourth "$@" ;;
  third) task_third "$@" ;;
  second) task_second "$@" ;;
  pmcstep) task_pmcstep "$@" ;;
  hang) task_hang "$@" ;;
  prof) task_prof "$@" ;;
  pmc) task_pmc "$@" ;;
  bench) task_bench "$@" ;;
  suite) task_suite "$@" ;;
  first)   # call 1 of the round: hang diagnosis, then kernel traces of HEAD for all five workloads, then the DCGAN PMC passes
    task_hang
    task_prof r4prof dcgan cyclegan srgan pix2pix wgan_gp wgan_gp:graph
    task_pmc dcgan dcgan G.conv ufwd,udgrad,uwgrad
    task_bench ;;
  *) echo "usage: gpu_r04.sh {hang|prof|pmc|bench|suite|first} [args]"; exit 2 ;;
esac
