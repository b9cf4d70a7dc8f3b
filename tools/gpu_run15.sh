#!/bin/bash
set -u
O=gpurun_out/r2o
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_steps_gpu.py -m gpu -q --tb=short -k "acgan" > $O/pytest_sel.log 2>&1; tail -12 $O/pytest_sel.log | cut -c1-250
