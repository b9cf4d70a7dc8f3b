"""How long does gloo take to all-reduce a CycleGAN-sized gradient bucket that lives on the GPU?  (The two-ranks-on-one-GPU test mode of
tests/test_steps_gpu.py stages CUDA tensors through the host; production uses RCCL.)  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/gloo_bucket_probe.py"""
import os
import time

import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
for mb in (0.4, 11, 14, 91):
    t = torch.ones(int(mb * 1e6 / 4), device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        dist.all_reduce(t)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    h = torch.ones(int(mb * 1e6 / 4))
    t0 = time.perf_counter()
    dist.all_reduce(h)
    th = time.perf_counter() - t0
    if rank == 0:
        print("gloo all_reduce of a %5.1f MB bucket, 2 ranks on one GPU: device tensor %.3f s (min of 3; %.0f MB/s), host tensor %.3f s; "
              "OMP_NUM_THREADS=%s" % (mb, min(times), mb / min(times), th, os.environ.get("OMP_NUM_THREADS")), flush=True)
dist.barrier()
dist.destroy_process_group()
