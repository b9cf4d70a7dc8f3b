#!/bin/bash
# round-3 GPU call 7: paired D pass (BatchNorm groups) parity + A/B, split-K rule check
mkdir -p gpurun_out/r3g
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3g
timeout 900 python -m pytest tests/test_steps_gpu.py tests/test_dp_gpu.py -q -x -k "dcgan or bench_config or two_ranks_on_one_gpu or cross_replica or rccl or acgan or pix2pix" > $O/pytest_steps.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_steps.txt
tail -15 $O/pytest_steps.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "batchnorm or norm or conv2d_fwd_bwd or splitk" > $O/pytest_ops.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_ops.txt
tail -3 $O/pytest_ops.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for cfg in "MIGAN_PAIR_D=0" "MIGAN_PAIR_D=1" "MIGAN_PAIR_D=1 MIGAN_SPLITK=0"; do
  echo "== bench dcgan $cfg" >> $O/bench.txt
  env $cfg timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | grep -v "^/opt" | cut -c1-300 >> $O/bench.txt
done
echo "== bench pix2pix" >> $O/bench.txt
timeout 600 python bench.py --workload pix2pix --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | grep -v "^/opt" | cut -c1-300 >> $O/bench.txt
cat $O/bench.txt | cut -c1-200
