#!/bin/bash
# Round-end measurement pass on the GPU box (run through gpurun from the repo root); everything lands in gpurun_out/.
#   gpurun --timeout 900 -- 'bash tools/final_measure.sh'
set -u
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.log; tail -1 $O/pytest_gpu.log
# PMC passes first: bench.py reads profiles/r01_pmc_traffic.json (regenerated below from these passes)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o dcgan -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
# HBM traffic of the dominant kernels: separate --pmc passes (FETCH_SIZE / WRITE_SIZE do not fit one pass)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$c -o mb -- \
     python $GRAFT_REPO_ROOT/tools/conv_microbench.py --shapes dcgan --match G.conv2 --iters 5 > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1)
done
python tools/collect_profiles.py --pmc-only > $O/pmc_table.log 2>&1; cat $O/pmc_table.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; head -c 200 $O/bench.json; echo
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_loop_probe.hip -o /tmp/mfma_probe 2> $O/probe_build.log && /tmp/mfma_probe > $O/mfma_loop_probe.txt 2>&1
for s in dcgan cyclegan srgan; do python tools/conv_microbench.py --shapes $s --iters 10 2>&1 | grep -v amdgpu.ids; done > $O/conv_microbench.txt
for w in cyclegan srgan pix2pix wgan_gp; do
  k=5; [ $w = wgan_gp ] && k=200; [ $w = pix2pix ] && k=50
  timeout 300 python tools/bench_models.py --workload $w --steps $k --warmup 3 2>/dev/null | tail -1
done > $O/models.txt
cat $O/models.txt
ls $O
