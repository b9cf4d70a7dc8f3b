#!/bin/bash
# round-3 GPU call 4: wgrad LDS-DMA BK 16 vs 32, planner occupancy, split sweep on R256
mkdir -p gpurun_out/r3d
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3d
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "conv or upconv or reflect_pad1 or toeplitz or bias_grad or direct_grad" > $O/pytest_conv.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_conv.txt
tail -3 $O/pytest_conv.txt
W="--dirs wgrad,uwgrad,twgrad --repeat 3 --iters 10"
for shp in dcgan cyclegan srgan; do
  for cfg in "MIGAN_DMA_WGRAD_BK=32 MIGAN_WGRAD_OCCS=3,4,7" "MIGAN_DMA_WGRAD_BK=32 MIGAN_WGRAD_OCCS=2,3,5" "MIGAN_DMA_WGRAD_BK=16 MIGAN_WGRAD_OCCS=3,4,7" "MIGAN_DMA_WGRAD_BK=16 MIGAN_WGRAD_OCCS=4,6,8"; do
    echo "== $shp $cfg" >> $O/mb_wgrad.txt
    env $cfg timeout 300 python tools/conv_microbench.py --shapes $shp $W 2>&1 | grep -v "^/opt" | grep -v "G.conv3\|D.conv\|c7s1-64\|D.c1\| wgrad .*up2" >> $O/mb_wgrad.txt
  done
done
for s in 8 12 14 16 21 24 28 32 42 56 64; do
  echo "== splits $s" >> $O/mb_splits.txt
  MIGAN_WGRAD_SPLITS=$s timeout 300 python tools/conv_microbench.py --shapes cyclegan --dirs wgrad,uwgrad --repeat 3 --iters 10 --match "256" 2>&1 | grep -v "^/opt" | grep -v " wgrad .*up2" >> $O/mb_splits.txt
  MIGAN_WGRAD_SPLITS=$s timeout 300 python tools/conv_microbench.py --shapes srgan --dirs wgrad --repeat 3 --iters 10 --match "6" 2>&1 | grep -v "^/opt" >> $O/mb_splits.txt
done
for wl in dcgan cyclegan srgan; do
  echo "== bench $wl" >> $O/bench.txt
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | grep -v "^/opt" | cut -c1-300 >> $O/bench.txt
done
cat $O/bench.txt | cut -c1-200
