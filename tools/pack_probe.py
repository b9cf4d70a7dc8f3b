"""How many weight-pack launches a CycleGAN step issues (debug aid): python tools/pack_probe.py [batch]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytorch_gan_amd import functional as F  # noqa: E402
from pytorch_gan_amd import models, steps  # noqa: E402
from pytorch_gan_amd._lib import lib  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
random.seed(0)
shape = (3, 256, 256)
nets = [models.CycleGenerator(shape, 9).to(dev), models.CycleGenerator(shape, 9).to(dev), models.CycleDiscriminator(shape).to(dev),
        models.CycleDiscriminator(shape).to(dev)]
s = steps.make_cyclegan_state(*nets)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
a, b = (torch.rand(bs, *shape) * 2 - 1).to(dev), (torch.rand(bs, *shape) * 2 - 1).to(dev)
for it in range(4):
    lib.migan_debug_launch_reset()
    steps.cyclegan_step(s, a, b)
    torch.cuda.synchronize()
    print(it, {k: lib.migan_debug_launch_count(k.encode()) for k in ("multi_permute4", "permute4_kernel", "pack_transpose", "upconv_pack", "toep_pack", "")})
print("two streams ok:", steps._two_streams_ok(s, a), "chains:", steps._CHAINS)
