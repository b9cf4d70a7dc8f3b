"""Fused multi-tensor Adam over a flat gradient bucket (one HIP launch per optimiser step).

Mirrors `torch.optim.Adam(params, lr=..., betas=(b1, b2))` as the reference constructs it (dcgan.py:134-135,
wgan_gp.py:112-113, cyclegan.py:87-92, pix2pix.py:74-75, srgan.py:81-82): same update arithmetic
(`_single_tensor_adam`), `zero_grad()` / `step()` call pattern, `param_groups[0]["lr"]`
for the LambdaLR schedule.

Design for MI355X: all gradients of one optimiser live in ONE flat fp32 buffer (`flat_grad`); every
`p.grad` is a view into it, so autograd accumulates in place, the data-parallel all-reduce is a single
RCCL call on `flat_grad`, and the update is a single kernel that reads a device-side pointer table.  The
step counter lives on the device so the whole training step can be captured in a hipGraph.
"""
import itertools

import numpy as np
import torch

from ._lib import check, lib

# Weight epochs (functional.set_weight_cache): process-wide unique, never re-used stamps, so a cached pack made under one
# optimiser can never match a parameter that a later optimiser object has updated in between.
_EPOCH = itertools.count(1)

_ADAM_T = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8")])
_BLK_T = np.dtype([("tensor", "<i4"), ("chunk", "<i4")])


def _to_device_bytes(arr, device):
    return torch.from_numpy(np.frombuffer(arr.tobytes(), dtype=np.uint8).copy()).to(device)


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("pytorch_gan_amd.optim.Adam needs parameters on the GPU (call .cuda() first, as the "
                               "reference does before building its optimisers)")
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("all parameters must be contiguous fp32 tensors on one device")
        self.param_groups = [{"params": self.params, "lr": float(lr), "betas": tuple(betas), "eps": float(eps)}]
        self.device = dev
        sizes = [p.numel() for p in self.params]
        # 64-element (256 B) aligned slots so every tensor starts on a fresh cache line
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 63) // 64 * 64
        self.total = off
        self.flat_grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(off, device=dev, dtype=torch.float32)
        self.step_t = torch.zeros(1, device=dev, dtype=torch.float32)
        self._attach()
        chunk = lib.migan_adam_chunk()
        tab = np.zeros(len(self.params), dtype=_ADAM_T)
        blks = []
        for i, (p, o, n) in enumerate(zip(self.params, self.offsets, sizes)):
            tab[i] = (p.data_ptr(), self.flat_grad.data_ptr() + 4 * o, self.exp_avg.data_ptr() + 4 * o,
                      self.exp_avg_sq.data_ptr() + 4 * o, n)
            blks += [(i, c) for c in range((n + chunk - 1) // chunk)]
        blk = np.array(blks, dtype=_BLK_T)
        self._tab = _to_device_bytes(tab, dev)
        self._blk = _to_device_bytes(blk, dev)
        self._nblocks = len(blks)
        self._ptrs = [p.data_ptr() for p in self.params]
        for p in self.params:
            p._migan_epoch = next(_EPOCH)  # renewed by step(): lets functional.set_weight_cache re-use packed weights

    def _attach(self):
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def zero_grad(self, set_to_none=False):
        """Zero the flat bucket (grads stay views of it; `set_to_none` is accepted for API parity)."""
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def step(self, grad_scale=1.0):
        for p, o, ptr in zip(self.params, self.offsets, self._ptrs):
            if p.data_ptr() != ptr:
                raise RuntimeError("parameter storage moved after the optimiser was built")
            g = p.grad
            if g is None:
                raise RuntimeError("a parameter has no gradient; the reference always back-props every parameter")
            if g.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                # someone replaced .grad (e.g. zero_grad(set_to_none) from foreign code): fold it back
                slot = self.flat_grad[o:o + p.numel()].view_as(p)
                slot.copy_(g)
                p.grad = slot
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        check(lib.migan_adam_step(self._tab.data_ptr(), self._blk.data_ptr(), self._nblocks, self.step_t.data_ptr(),
                                  float(g0["lr"]), float(b1), float(b2), float(g0["eps"]), float(grad_scale),
                                  torch.cuda.current_stream().cuda_stream), "adam_step")
        ep = next(_EPOCH)
        for p in self.params:
            p._migan_epoch = ep

    def state_dict(self):
        return {"step": self.step_t.clone(), "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.step_t.copy_(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0].update(sd["param_groups"][0])
