#!/bin/bash
# Round-6 gpurun calls:  gpurun --timeout T -- "bash tools/gpu_r06.sh <task> [args]"
# Tasks write under gpurun_out/<dir>/ (scratch); what DESIGN.md quotes is copied into profiles/r06_*.
# ab_base/ = `git archive a3ff5e2` (the round-start tree) + its built library: git-ignored, travels with the snapshot.
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
  local O=gpurun_out/${1:-r6prof}; shift
  mkdir -p $O
  for spec in "$@"; do
    local b=; case $spec in *@*) b=${spec##*@}; spec=${spec%@*} ;; esac
    w=${spec%%:*}; mode=eager; flag="--no-graph --no-overlap"; [ "$spec" != "$w" ] && { mode=graph; flag=; }
    [ "$spec" = "$w:ov" ] && { mode=eagerov; flag="--no-graph"; }   # eager WITH the weight-gradient / second-discriminator streams
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
  local O=gpurun_out/r6pmcstep_$w
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
  python tools/pmc_step.py r06 $(for n in "$@"; do echo $O/$n; done) > $O/summary.txt 2>&1
  head -30 $O/summary.txt | cut -c1-200
  find $O -name "*.csv" -size +6M -delete
  cp profiles/r06_pmc_kernels.json gpurun_out/r06_pmc_kernels.json 2>/dev/null
}

task_bench() {
  local O=gpurun_out/r6bench; mkdir -p $O
  timeout 900 python bench.py "$@" > $O/bench_default.json 2> $O/bench_default.err
  echo "bench rc=$?"; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r6bench/bench_default.json').read().strip().splitlines()[-1])
print({k:r[k] for k in ('value','ms_per_step')})
rf=r.get('roofline',{}); print({k:rf.get(k) for k in ('kernel','frac','traffic','mfma_busy_frac','symbol','avg_launch_ms')})
print('cpu', r.get('cpu_baseline'))
for k,v in r.get('extra',{}).items(): print(k, {a:v.get(a) for a in ('images_per_s','ms_per_step','hipgraph','eager_ms_per_step','library_launches_per_step','error','hipgraph_error')}, (v.get('cpu_baseline') or {}).get('value'))
PY
  tail -5 $O/bench_default.err
}

task_suite() {
  local O=gpurun_out/r6suite; mkdir -p $O
  timeout 1500 python -m pytest tests -m gpu -q --timeout=700 --durations=25 -rxs > $O/pytest_gpu.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest_gpu.txt
  grep -v "^  \|^$" $O/pytest_gpu.txt | tail -60
}


# targeted parity:  tests <outdir> <pytest args...>
task_tests() {
  local O=gpurun_out/${1:-r6t}; shift; mkdir -p $O
  timeout 1500 python -m pytest "$@" -q --timeout=700 --durations=10 -rxs > $O/pytest.txt 2>&1
  echo "pytest rc=$?" >> $O/pytest.txt
  grep -v "^  \|^$" $O/pytest.txt | tail -40
}

# whole-step A/B against the round-start tree: abw <outdir> <workload> <steps> [reps] [flags...]
task_abw() {
  local O=gpurun_out/${1:-r6ab}; shift; mkdir -p $O
  ab $O/bench.txt "$@"
  cat $O/bench.txt
}

# stand-alone layer times: micro <outdir> <conv_microbench args...>
task_micro() {
  local O=gpurun_out/${1:-r6micro}; shift; mkdir -p $O
  timeout 600 python tools/conv_microbench.py "$@" 2>&1 | tee -a $O/micro.txt | grep " us " | cut -c1-160
}

# whole-step A/B against the round-start tree with bench flags:  abf <outdir> <workload> <steps> <reps> [--flag ...]
task_abf() {
  local O=gpurun_out/${1:-r6abf}; local w=$2 k=$3 reps=$4; shift 4; mkdir -p $O
  for r in $(seq $reps); do
    (cd ab_base && echo "== base $w $*" >> $R/$O/bench.txt && timeout 300 python bench.py --workload $w --steps $k --warmup 2 --no-cpu-baseline --no-extra --no-roofline "$@" 2>>$R/$O/bench.txt.err | line >> $R/$O/bench.txt)
    bl $O/bench.txt $w $k "$@"
  done
  cat $O/bench.txt
}

# same-box A/B of environment knobs on the tree itself:  abenv <outdir> <workload> <steps> <reps> [--flag ...] -- ENV=a ENV=b ...
task_abenv() {
  local O=gpurun_out/${1:-r6ab}; local w=$2 k=$3 reps=$4; shift 4; mkdir -p $O
  local flags=()
  while [ "$1" != "--" ]; do flags+=("$1"); shift; done; shift
  for r in $(seq $reps); do
    for e in "$@"; do bl $O/bench.txt $w $k $e "${flags[@]}"; done
  done
  cat $O/bench.txt
}

# stand-alone normalisation passes under knob settings:  normab <outdir> "<microbench args>" ENV=a ENV=b ...
task_normab() {
  local O=gpurun_out/${1:-r6norm}; local margs=$2; shift 2; mkdir -p $O
  for e in "$@"; do
    echo "== $e" | tee -a $O/norm.txt
    env $e timeout 300 python tools/norm_microbench.py $margs 2>&1 | grep " us " | tee -a $O/norm.txt | cut -c1-150
  done
}

# closing pass of the final tree: PMC passes over the steps (so that the bench line that follows reads THIS library's counter table),
# default bench line, kernel traces of all workloads
task_final() {
  rm -f profiles/r06_pmc_kernels.json   # (the box's copy: the table is rebuilt from this library's passes alone)
  task_pmcstep dcgan 3 sq l2 fetch write
  task_pmcstep cyclegan 1 sq fetch write
  task_pmcstep srgan 1 sq fetch write
  task_pmcstep pix2pix 3 sq fetch write
  task_bench
  task_prof r6final dcgan dcgan:graph cyclegan srgan wgan_gp:graph pix2pix pix2pix:graph cyclegan:graph@1
}

t=${1:-}; shift || true
case "$t" in
  prof) task_prof "$@" ;;
  pmcstep) task_pmcstep "$@" ;;
  bench) task_bench "$@" ;;
  suite) task_suite "$@" ;;
  *) if declare -F "task_$t" > /dev/null; then "task_$t" "$@"; else echo "usage: gpu_r06.sh <task> [args]"; exit 2; fi ;;
esac
