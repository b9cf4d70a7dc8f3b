#!/bin/bash
# round-3 GPU call 8: F2 clone tests + N1 trajectory tests + whole ops file
mkdir -p gpurun_out/r3h
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3h
timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_steps_gpu.py -q -k "clone or trajectory or pullaway" --durations=15 > $O/pytest_new.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_new.txt
tail -40 $O/pytest_new.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_data_gpu.py -q > $O/pytest_ops.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_ops.txt
tail -4 $O/pytest_ops.txt
