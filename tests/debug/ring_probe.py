"""Diagnostic: ordering of the reflect-1 ring correction on a side stream."""
import os
import sys

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pytorch_gan_amd as pg  # noqa: E402
from pytorch_gan_amd._lib import check, lib  # noqa: E402

F = pg.functional
DEV = "cuda:0"
torch.manual_seed(0)
N, Ci, H, W, Co = 1, 256, 12, 12, 256
x = torch.randn(N, Ci, H, W, requires_grad=True)
w = (torch.randn(Co, Ci, 3, 3) * 0.2).requires_grad_(True)
y = TF.conv2d(TF.pad(x, (1, 1, 1, 1), mode="reflect"), w, None, 1)
gy = torch.randn_like(y)
y.backward(gy)
dy = gy.to(DEV).contiguous(memory_format=torch.channels_last)
wt = w.detach().to(DEV).permute(1, 2, 3, 0).contiguous()   # ihwo
want = x.grad


def rel(a):
    a = a.permute(0, 3, 1, 2).cpu() if a.dim() == 4 and a.shape[-1] == Ci and a.shape[1] != Ci else a.cpu()
    return float((a.double() - want.double()).norm() / want.double().norm())


main = torch.cuda.current_stream()
side = torch.cuda.Stream()
for mode in ("same-stream", "side+wait_stream", "side+sync", "side+event"):
    dx = torch.full((N, H, W, Ci), 1e6, device=DEV)   # NHWC storage, poisoned
    check(lib.migan_conv2d_dgrad(dy.data_ptr(), wt.data_ptr(), None, dx.data_ptr(), N, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 0, 0.0,
                                 main.cuda_stream))
    if mode == "same-stream":
        st = main
    elif mode == "side+wait_stream":
        side.wait_stream(main)
        st = side
    elif mode == "side+sync":
        torch.cuda.synchronize()
        st = side
    else:
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        st = side
    check(lib.migan_conv2d_dgrad_reflect1_ring(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), N, H, W, Ci, Co, st.cuda_stream))
    main.wait_stream(side)
    torch.cuda.synchronize()
    print(mode, "rel err %.3e" % rel(dx), "main", hex(main.cuda_stream), "side", hex(side.cuda_stream))
# through the Function
for ov in (True, False):
    F._RING_OVERLAP = ov
    xg = x.detach().to(DEV).requires_grad_(True)
    wg = w.detach().to(DEV).requires_grad_(True)
    out = F.conv2d(xg, wg, None, 1, (1, 1, 1, 1), F.GATHER_REFLECT)
    out.backward(gy.to(DEV))
    torch.cuda.synchronize()
    print("Function ring_overlap=%s rel err %.3e wgrad rel %.3e" % (ov, float((xg.grad.cpu().double() - want.double()).norm() / want.double().norm()),
                                                                     float((wg.grad.cpu().double() - w.grad.double()).norm() / w.grad.double().norm())))
