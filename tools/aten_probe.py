"""Which ATen ops run inside one eager step of a product-model workload (debug aid for tests/test_steps_gpu.py::test_step_bodies_launch_no_aten_kernels):
    python tools/aten_probe.py srgan"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from pytorch_gan_amd import models, steps  # noqa: E402

DEV = "cuda:0"
torch.manual_seed(0)
G, D, V = models.SrganGenerator(3, 3, 2).to(DEV), models.SrganDiscriminator((3, 64, 64)).to(DEV), models.SrganFeatureExtractor().to(DEV)
s = steps.make_srgan_state(G, D, V)
lr, hr = torch.randn(2, 3, 16, 16).to(DEV), torch.randn(2, 3, 64, 64).to(DEV)
for _ in range(2):
    steps.srgan_step(s, lr, hr)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    steps.srgan_step(s, lr, hr)
    torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::copy_", "aten::cat", "aten::mul"):
        print(e.name, e.input_shapes, [str(f) for f in (e.stack or [])[:6]])
