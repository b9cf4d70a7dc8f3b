#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3i
mkdir -p $O
timeout 600 python -m pytest tests/test_steps_gpu.py -q -k "srgan_trajectory or wgan_gp_trajectory" > $O/pytest_traj.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_traj.txt
tail -5 $O/pytest_traj.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 600 $O/bench.json; echo; tail -3 $O/bench.err
bash tools/r3/pmc_gap.sh > $O/pmc_gap.log 2>&1; tail -60 $O/pmc_gap.log
