#!/bin/bash
set -u
O=gpurun_out/r2k
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_data_gpu.py tests/test_ops_gpu.py -m gpu -q --tb=short -k "data or pack_plan or resize or transform or dataset or rejects" > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log
J='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["config"].get("hipgraph"))'
for v in "MIGAN_BATCH_PACKS=1" "MIGAN_BATCH_PACKS=0" "MIGAN_BATCH_PACKS_MAX=1000000000"; do
  echo "== dcgan $v"
  env $v timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-roofline --no-cpu-baseline --no-extra 2>/dev/null | python -c "$J"
  echo "== pix2pix $v"
  env $v timeout 300 python bench.py --workload pix2pix --steps 30 --warmup 5 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
  echo "== esrgan $v"
  env $v timeout 300 python bench.py --workload esrgan --steps 5 --warmup 2 --min-seconds 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "$J"
done > $O/packs_ab.txt 2>&1
cat $O/packs_ab.txt
python - <<'PY'
import time, torch, numpy as np
import pytorch_gan_amd.data as D
a = torch.randint(0, 256, (256, 218, 178, 3), dtype=torch.uint8, device="cuda:0")
pipe = D.ImagePipeline(resize=286, crop=(256, 256), hflip_p=0.5, mean=(0.5,)*3, std=(0.5,)*3)
for _ in range(3): pipe(a)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): pipe(a)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("pipeline 256 x 218x178x3 -> resize 286 bicubic -> crop 256 -> flip -> normalize: %.3f ms/batch, %.0f img/s" % (dt * 1e3, 256 / dt))
from PIL import Image
img = Image.fromarray(a[0].cpu().numpy())
t0 = time.perf_counter()
for _ in range(50): img.resize((286, 350), Image.BICUBIC)
print("Pillow bicubic resize alone, one host core: %.3f ms/img" % ((time.perf_counter() - t0) / 50 * 1e3))
PY
