#!/bin/bash
# round-3 GPU call 5: deep-pipeline + split-K small-GEMM path: parity, pix2pix / dcgan-D microbench A/B, whole steps
mkdir -p gpurun_out/r3e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3e
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv or upconv or reflect_pad1 or toeplitz or bias_grad or direct_grad or splitk" > $O/pytest_conv.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_conv.txt
tail -5 $O/pytest_conv.txt
D="--dirs fwd,dgrad --repeat 3 --iters 20"
for cfg in "MIGAN_DMA_DEEP=0 MIGAN_SPLITK=0" "MIGAN_DMA_DEEP=1 MIGAN_SPLITK=0" "MIGAN_DMA_DEEP=1 MIGAN_SPLITK=1"; do
  echo "== pix2pix $cfg" >> $O/mb_small.txt
  env $cfg timeout 300 python tools/conv_microbench.py --shapes pix2pix $D 2>&1 | grep -v "^/opt" >> $O/mb_small.txt
  echo "== dcgan $cfg" >> $O/mb_small.txt
  env $cfg timeout 300 python tools/conv_microbench.py --shapes dcgan $D --match "D.conv" 2>&1 | grep -v "^/opt" >> $O/mb_small.txt
  echo "== cyclegan $cfg" >> $O/mb_small.txt
  env $cfg timeout 300 python tools/conv_microbench.py --shapes cyclegan $D --match "D.c" 2>&1 | grep -v "^/opt" >> $O/mb_small.txt
done
for wl in dcgan pix2pix; do
  for cfg in "MIGAN_DMA_DEEP=0 MIGAN_SPLITK=0" "MIGAN_DMA_DEEP=1 MIGAN_SPLITK=0" "MIGAN_DMA_DEEP=1 MIGAN_SPLITK=1"; do
    echo "== bench $wl $cfg" >> $O/bench.txt
    env $cfg timeout 600 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-roofline 2>&1 | grep -v "^/opt" | cut -c1-300 >> $O/bench.txt
  done
done
cat $O/bench.txt | cut -c1-200
