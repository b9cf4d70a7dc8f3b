"""Diagnosis aid (round 5): record the CycleGAN step into a hipGraph under one combination of the step body's stream forks, in THIS
process (a crash inside hipStreamEndCapture takes only this probe down).

    python tools/capture_probe.py <chains 0|1> <d_fork 0|1> <wgrad_stream 0|1> [side=64] [n_res=2] [batch=2]
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytorch_gan_amd import functional as F  # noqa: E402
from pytorch_gan_amd import models, steps  # noqa: E402

chains, dfork, wstream = (int(a) for a in sys.argv[1:4])
side = int(sys.argv[4]) if len(sys.argv) > 4 else 64
n_res = int(sys.argv[5]) if len(sys.argv) > 5 else 2
batch = int(sys.argv[6]) if len(sys.argv) > 6 else 2
steps._CHAINS, steps._OVERLAP_D, F._WGRAD_STREAM = bool(chains), bool(dfork), bool(wstream)
if os.environ.get("PROBE_WGRAD_MIN"):
    F._WGRAD_STREAM_MIN = int(os.environ["PROBE_WGRAD_MIN"])
torch.manual_seed(0)
random.seed(0)
shape = (3, side, side)
dev = torch.device("cuda", 0)
nets = [models.CycleGenerator(shape, n_res), models.CycleGenerator(shape, n_res), models.CycleDiscriminator(shape), models.CycleDiscriminator(shape)]
for n in nets:
    n.apply(models.init_normal_cyclegan)
nets = [n.to(dev) for n in nets]
st = steps.make_cyclegan_state(*nets)
st.buf_A.max_size = st.buf_B.max_size = 3
a = (torch.rand(batch, *shape, device=dev) * 2 - 1)
b = (torch.rand(batch, *shape, device=dev) * 2 - 1)
print("probe chains=%d d_fork=%d wgrad_stream=%d %s n_res=%d batch=%d" % (chains, dfork, wstream, shape, n_res, batch), flush=True)
r = steps.CycleGanRunner(st, a, b, use_graph=True, warmup=2).prepare()
print("  graphed:", r.graphed, "error:", r.capture_error, flush=True)
for _ in range(3):
    o = r.run()
torch.cuda.synchronize()
print("  replays ok, loss_G %.5f" % float(o["loss_G"]), flush=True)
