#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3k
mkdir -p $O
timeout 100 python tools/gpu_sanity.py || { echo "bad box, giving up"; exit 3; }
# the test that hung in the full-suite run: hard 150 s limit, stacks of every rank on abort
PYTHONFAULTHANDLER=1 timeout -s ABRT 150 python -m pytest tests/test_steps_gpu.py -q -x -k "cyclegan_strong_scaling" --timeout=140 > $O/pytest_2rank.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_2rank.txt
grep -v "^  " $O/pytest_2rank.txt | tail -25 | cut -c1-200
timeout 170 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 300 $O/bench.json; echo
