"""Diagnostic 2: the pieces of steps.compute_gradient_penalty around the critic (interpolation, row norm, loss)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import gpu_copy  # noqa: E402

from oracle import reference_models as M  # noqa: E402
from pytorch_gan_amd import functional as F  # noqa: E402

DEV = "cuda:0"
g = np.load(os.path.join(ROOT, "tests", "golden", "critic_gp_32.npz"))
real, fake, alpha = (torch.from_numpy(g["dualgan_" + k]).to(DEV) for k in ("real", "fake", "alpha"))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(b.norm(), 1e-300))


B = 4
a = alpha.reshape(B)
r_, f_ = F.canon(real), F.canon(fake)
mix = F.axpby(F.rowscale(r_, a), F.rowscale(f_, 1 - a), 1.0, 1.0)
want = alpha * real + (1 - alpha) * fake
print("mix layout", mix.shape, mix.stride(), "rel vs torch", "%.2e" % rel(mix, want))
x = torch.randn(4, 3, 32, 32, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
rows = x.permute(0, 2, 3, 1).reshape(B, -1)
n = F.rownorm(rows)
print("rownorm fwd rel", "%.2e" % rel(n, x.detach().reshape(B, -1).norm(2, dim=1)))
loss = F.loss(F.LOSS_MSE, n, None, 1.0)
loss.backward()
xr = x.detach().clone().requires_grad_(True)
lr = ((xr.reshape(B, -1).norm(2, dim=1) - 1) ** 2).mean()
lr.backward()
print("loss rel %.2e" % (abs(float(loss) - float(lr)) / float(lr)), "grad rel %.2e" % rel(x.grad, xr.grad))
# gradient tensor as the critic returns it
torch.manual_seed(0)
Dg = gpu_copy(M.DualganDiscriminator(3))
m2 = mix.view(real.shape).requires_grad_(True)
d = Dg(m2)
with F.input_grad_only():
    gr = torch.autograd.grad(d, m2, torch.ones(d.shape, device=DEV), create_graph=True)[0]
print("grads type", type(gr).__name__, gr.shape, gr.stride(), "contig", gr.is_contiguous(), "cl", gr.is_contiguous(memory_format=torch.channels_last))
gr2 = torch.autograd.grad(d, m2, torch.ones(d.shape, device=DEV), create_graph=True)[0]
print("input_grad_only vs not: rel %.2e" % rel(gr, gr2))
rows = gr.permute(0, 2, 3, 1).reshape(B, -1) if not gr.is_contiguous() else gr.view(B, -1)
print("rownorm of critic grads rel %.2e" % rel(F.rownorm(rows), gr.detach().reshape(B, -1).norm(2, dim=1)))
for tag, ctx in (("input_grad_only", F.input_grad_only), ("plain", None)):
    for p in Dg.parameters():
        p.grad = None
    mm = mix.detach().view(real.shape).requires_grad_(True)
    dd = Dg(mm)
    if ctx:
        with ctx():
            gg = torch.autograd.grad(dd, mm, torch.ones(dd.shape, device=DEV), create_graph=True)[0]
    else:
        gg = torch.autograd.grad(dd, mm, torch.ones(dd.shape, device=DEV), create_graph=True)[0]
    pen_t = ((gg.reshape(B, -1).norm(2, dim=1) - 1) ** 2).mean()
    pen_t.backward()
    gt = [p.grad.clone() if p.grad is not None else None for p in Dg.parameters()]
    for p in Dg.parameters():
        p.grad = None
    mm = mix.detach().view(real.shape).requires_grad_(True)
    dd = Dg(mm)
    if ctx:
        with ctx():
            gg = torch.autograd.grad(dd, mm, torch.ones(dd.shape, device=DEV), create_graph=True)[0]
    else:
        gg = torch.autograd.grad(dd, mm, torch.ones(dd.shape, device=DEV), create_graph=True)[0]
    rows = gg.permute(0, 2, 3, 1).reshape(B, -1) if not gg.is_contiguous() else gg.view(B, -1)
    pen_k = F.loss(F.LOSS_MSE, F.rownorm(rows), None, 1.0)
    pen_k.backward()
    print(tag, "penalty torch-ops %.8f kernels %.8f" % (float(pen_t), float(pen_k)))
    for (k, p), t in zip(Dg.named_parameters(), gt):
        if t is not None and p.grad is not None:
            print("   ", k, "kernel-path vs torch-ops-path rel %.2e" % rel(p.grad, t))
