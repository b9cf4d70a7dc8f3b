#!/bin/bash
# round-3 GPU call 3: LDS-DMA wgrad parity + A/B, auto tile selection check, whole-step A/B
mkdir -p gpurun_out/r3c
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3c
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv or upconv or reflect_pad1 or toeplitz or bias_grad or direct_grad" > $O/pytest_conv.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_conv.txt
tail -5 $O/pytest_conv.txt
for shp in dcgan cyclegan srgan; do
  for dma in 0 1; do
    echo "== $shp MIGAN_DMA_WGRAD=$dma" >> $O/mb_wgrad.txt
    MIGAN_DMA_WGRAD=$dma timeout 300 python tools/conv_microbench.py --shapes $shp --dirs wgrad,uwgrad,twgrad --repeat 3 --iters 10 2>&1 | grep -v "^/opt" >> $O/mb_wgrad.txt
  done
  echo "== $shp auto" >> $O/mb_auto.txt
  timeout 300 python tools/conv_microbench.py --shapes $shp --dirs fwd,dgrad,rdgrad,ufwd,udgrad --repeat 3 --iters 10 2>&1 | grep -v "^/opt" >> $O/mb_auto.txt
done
for wl in dcgan cyclegan srgan; do
  for cfg in "MIGAN_DMA=0 MIGAN_DMA_WGRAD=0" "MIGAN_DMA=1 MIGAN_DMA_WGRAD=0" "MIGAN_DMA=1 MIGAN_DMA_WGRAD=1"; do
    echo "== bench $wl $cfg" >> $O/bench.txt
    env $cfg timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | grep -v "^/opt" | cut -c1-300 >> $O/bench.txt
  done
done
cat $O/mb_wgrad.txt; cat $O/bench.txt | cut -c1-200
