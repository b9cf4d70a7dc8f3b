#!/bin/bash
# round-3 GPU call 2: tile / BK sweep of the LDS-DMA igemm kernels (forced tiles, 3 repeats, min reported)
mkdir -p gpurun_out/r3b
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
D="--dirs fwd,dgrad,rdgrad,ufwd,udgrad --repeat 3 --iters 10"
for round in 1 2; do
for t in 0 32128128 32256064 32128064 32064064 16128128 16256064 16128064 16064064; do
  echo "== round $round dcgan MIGAN_DMA_TILE=$t" >> gpurun_out/r3b/sweep.txt
  MIGAN_DMA_TILE=$t timeout 120 python tools/conv_microbench.py --shapes dcgan $D --match "G.conv" 2>&1 | grep -v "^/opt" | grep -v "conv3" >> gpurun_out/r3b/sweep.txt
  echo "== round $round cyclegan MIGAN_DMA_TILE=$t" >> gpurun_out/r3b/sweep.txt
  MIGAN_DMA_TILE=$t timeout 120 python tools/conv_microbench.py --shapes cyclegan $D 2>&1 | grep -v "^/opt" | grep -v "c7s1\|D.c1" >> gpurun_out/r3b/sweep.txt
  echo "== round $round srgan MIGAN_DMA_TILE=$t" >> gpurun_out/r3b/sweep.txt
  MIGAN_DMA_TILE=$t timeout 120 python tools/conv_microbench.py --shapes srgan $D 2>&1 | grep -v "^/opt" | grep -v "conv3" >> gpurun_out/r3b/sweep.txt
done
done
tail -30 gpurun_out/r3b/sweep.txt
