#!/bin/bash
# round-3 GPU call 1: parity of the LDS-DMA igemm kernels + A/B microbench against the register-staged kernels
mkdir -p gpurun_out/r3a
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "conv2d_fwd_bwd or conv_transpose or upconv or reflect_pad1 or conv_double or dropout2d_fused or toeplitz" > gpurun_out/r3a/pytest_conv.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r3a/pytest_conv.txt
tail -5 gpurun_out/r3a/pytest_conv.txt
for shp in dcgan cyclegan srgan; do
  for dma in 0 1; do
    echo "== $shp MIGAN_DMA=$dma" >> gpurun_out/r3a/mb.txt
    MIGAN_DMA=$dma timeout 300 python tools/conv_microbench.py --shapes $shp --only fwd,dgrad >> gpurun_out/r3a/mb.txt 2>&1
  done
done
for t in 128128 128064 256064 64064; do
  echo "== cyclegan R256/u128 MIGAN_DMA_TILE=$t" >> gpurun_out/r3a/mb_tile.txt
  MIGAN_DMA_TILE=$t timeout 300 python tools/conv_microbench.py --shapes cyclegan --only fwd,dgrad --match "256" >> gpurun_out/r3a/mb_tile.txt 2>&1
  echo "== srgan MIGAN_DMA_TILE=$t" >> gpurun_out/r3a/mb_tile.txt
  MIGAN_DMA_TILE=$t timeout 300 python tools/conv_microbench.py --shapes srgan --only fwd,dgrad --match "64" >> gpurun_out/r3a/mb_tile.txt 2>&1
done
for dma in 0 1; do
  echo "== bench dcgan MIGAN_DMA=$dma" >> gpurun_out/r3a/bench.txt
  MIGAN_DMA=$dma timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra >> gpurun_out/r3a/bench.txt 2>&1
done
cat gpurun_out/r3a/mb.txt | tail -70
