#!/bin/bash
set -u
O=gpurun_out/r2p
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_dp_gpu.py -m gpu -q --tb=short -x > $O/pytest_sel.log 2>&1; tail -4 $O/pytest_sel.log
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["config"].get("hipgraph"))'
for i in 1 2; do
echo "== dcgan"
timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "$J"
done > $O/dcgan.txt 2>&1
cat $O/dcgan.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_dcgan -o dcgan -- python $R/bench.py --steps 20 --warmup 2 --min-seconds 0 --no-graph --no-roofline --no-cpu-baseline --no-extra > $R/$O/prof_dcgan.log 2>&1)
python tools/rocpd_stats.py $O/prof_dcgan/*.db 70 25 | grep -E "total|finalize|multi_permute|norm_partial" | cut -c1-170
