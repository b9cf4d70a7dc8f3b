#!/bin/bash
# full GPU suite on the current tree (the driver's round-end command), after a GPU sanity check
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3j
mkdir -p $O
timeout 120 python tools/gpu_sanity.py || { echo "bad box, giving up"; exit 3; }
timeout 1300 python -m pytest tests -m gpu -q --timeout=700 --durations=25 > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.txt
grep -v "^  \|^$" $O/pytest_gpu.txt | tail -60
