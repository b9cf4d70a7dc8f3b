"""Diagnostic: where does the HIP dualgan-critic gradient penalty leave the fp64 evaluation? (run on the GPU box)"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import gpu_copy  # noqa: E402

from oracle import reference_models as M  # noqa: E402
from oracle import reference_steps as S  # noqa: E402
from pytorch_gan_amd import steps  # noqa: E402

DEV = "cuda:0"
g = np.load(os.path.join(ROOT, "tests", "golden", "critic_gp_32.npz"))
real, fake, alpha = (torch.from_numpy(g["dualgan_" + k]) for k in ("real", "fake", "alpha"))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(b.norm(), 1e-300))


def variant(tag, specs_fn):
    torch.manual_seed(0)
    D = specs_fn()
    D64 = copy.deepcopy(D).double()
    Dg = gpu_copy(D)
    x64 = (alpha.double() * real.double() + (1 - alpha.double()) * fake.double()).requires_grad_(True)
    xg = x64.detach().float().to(DEV).requires_grad_(True)
    o64, og = D64(x64), Dg(xg)
    print(tag, "fwd rel", "%.2e" % rel(og, o64))
    g64 = torch.autograd.grad(o64, x64, torch.ones_like(o64), create_graph=True)[0]
    gg = torch.autograd.grad(og, xg, torch.ones_like(og), create_graph=True)[0]
    print(tag, "dD/dx rel", "%.2e" % rel(gg, g64))
    p64 = ((g64.view(4, -1).norm(2, dim=1) - 1) ** 2).mean()
    pg = ((gg.reshape(4, -1).norm(2, dim=1) - 1) ** 2).mean()   # torch ops on the HIP gradient (differentiable through the Functions)
    print(tag, "gp", float(pg), float(p64), "rel %.2e" % (abs(float(pg) - float(p64)) / abs(float(p64))))
    p64.backward()
    pg.backward()
    for (k, a), (_, b) in zip(Dg.named_parameters(), D64.named_parameters()):
        if b.grad is not None:
            print("   ", tag, k, "rel %.2e" % rel(a.grad, b.grad), "|f64| %.2e" % float(b.grad.norm()))
    # the product function (rowscale/axpby interpolation, rownorm, loss kernels)
    for p in Dg.parameters():
        p.grad = None
    gp2 = steps.compute_gradient_penalty(Dg, real.to(DEV), fake.to(DEV), alpha.to(DEV))
    print(tag, "product gp rel %.2e" % (abs(float(gp2.detach()) - float(p64)) / abs(float(p64))))


nn = torch.nn
variant("dualgan", lambda: M.DualganDiscriminator(3))
variant("no-bn", lambda: torch.nn.Sequential(nn.Conv2d(3, 64, 4, 2, 1), nn.LeakyReLU(0.2), nn.Conv2d(64, 128, 4, 2, 1), nn.LeakyReLU(0.2),
                                            nn.Conv2d(128, 256, 4, 2, 1), nn.LeakyReLU(0.2), nn.ZeroPad2d((1, 0, 1, 0)), nn.Conv2d(256, 1, 4)))
variant("bn-no-act", lambda: torch.nn.Sequential(nn.Conv2d(3, 64, 4, 2, 1), nn.LeakyReLU(0.2), nn.Conv2d(64, 128, 4, 2, 1), nn.BatchNorm2d(128, 0.8),
                                                nn.Conv2d(128, 256, 4, 2, 1), nn.BatchNorm2d(256, 0.8), nn.ZeroPad2d((1, 0, 1, 0)), nn.Conv2d(256, 1, 4)))
variant("bn-act-sympad", lambda: torch.nn.Sequential(nn.Conv2d(3, 64, 4, 2, 1), nn.LeakyReLU(0.2), nn.Conv2d(64, 128, 4, 2, 1), nn.BatchNorm2d(128, 0.8),
                                                    nn.LeakyReLU(0.2), nn.Conv2d(128, 256, 4, 2, 1), nn.BatchNorm2d(256, 0.8), nn.LeakyReLU(0.2),
                                                    nn.Conv2d(256, 1, 3, 1, 1)))
