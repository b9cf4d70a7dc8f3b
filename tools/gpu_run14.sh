#!/bin/bash
set -u
O=gpurun_out/r2n
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_steps_gpu.py -m gpu -q --tb=short -k "critic or wgan or dragan or gradient" > $O/pytest_sel.log 2>&1; tail -15 $O/pytest_sel.log
