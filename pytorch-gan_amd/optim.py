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
import weakref

import numpy as np
import torch

from . import functional as F
from ._lib import check, lib

# Weight epochs (functional.set_weight_cache): process-wide unique, never re-used stamps, so a cached pack made under one
# optimiser can never match a parameter that a later optimiser object has updated in between.
_EPOCH = itertools.count(1)

_ADAM_T = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8")])
_BLK_T = np.dtype([("tensor", "<i4"), ("chunk", "<i4")])


def bucket_layout(sizes):
    """(offsets, total) of the flat bucket: one 64-element (256 B) aligned slot per tensor, so every tensor starts on a fresh
    cache line and the padding between slots is part of what the data-parallel all-reduce moves (tests/test_dp_cpu.py drives
    dp.DataParallel over exactly this layout on the CPU)."""
    offsets, off = [], 0
    for n in sizes:
        offsets.append(off)
        off += (int(n) + 63) // 64 * 64
    return offsets, off


def _to_device_bytes(arr, device):
    return torch.from_numpy(np.frombuffer(arr.tobytes(), dtype=np.uint8).copy()).to(device)


class Adam:
    _live = weakref.WeakSet()  # sync_all_lr(): learning rates of every live optimiser -> their device scalars

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        if not F.on_device(self.params[0]):
            raise RuntimeError("pytorch_gan_amd.optim.Adam needs parameters on the GPU (call .cuda() first, as the "
                               "reference does before building its optimisers)")
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("all parameters must be contiguous fp32 tensors on one device")
        self.param_groups = [{"params": self.params, "lr": float(lr), "betas": tuple(betas), "eps": float(eps)}]
        self.device = dev
        sizes = [p.numel() for p in self.params]
        self.offsets, off = bucket_layout(sizes)
        self.total = off
        self.flat_grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(off, device=dev, dtype=torch.float32)
        self.step_t = torch.zeros(1, device=dev, dtype=torch.float32)   # fp32 like torch.optim.Adam's state["step"]
        self._ticket = torch.zeros(1, device=dev, dtype=torch.int32)
        # the kernel reads the learning rate from the device, so a captured hipGraph follows LambdaLR (cyclegan.py:275-277)
        self.lr_t = torch.full((1,), float(lr), device=dev, dtype=torch.float32)
        self._lr_on_device = float(lr)
        self.pending = None  # event of an update still running on a side stream (dp.DataParallel)
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
        Adam._live.add(self)

    def _attach(self):
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def wait_pending(self):
        """Make the current stream wait for an update of this optimiser that is still running on a side stream."""
        ev, self.pending = self.pending, None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def sync_lr(self):
        """param_groups[0]['lr'] -> the device scalar the kernel reads (only when it changed; never while capturing)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device and not torch.cuda.is_current_stream_capturing():
            self.lr_t.fill_(lr)
            self._lr_on_device = lr

    def zero_grad(self, set_to_none=False):
        """Zero the flat bucket.  Deviation from torch: grads stay views of the bucket (`set_to_none` is accepted for
        API parity and ignored), because the wgrad kernels and the all-reduce work on the bucket in place."""
        self.wait_pending()  # a side-stream update may still be reading the bucket
        if self.flat_grad.is_cuda:
            from . import functional as F

            F.zero_(self.flat_grad)   # hipMemsetAsync through the C ABI (torch's fill is an ATen kernel)
        else:
            self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def attach_grads(self):
        """zero_grad() without the fill: every `p.grad` is (again) its view of the bucket.  For callers whose backward WRITES every
        gradient of this optimiser (the fused WGAN-GP kernels, steps.wgan_gp_step) - zeroing first would be a dead store."""
        self.wait_pending()
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def step(self, grad_scale=1.0):
        F.join_wgrad_streams()   # weight gradients still in flight on their own stream (functional._Fork) land before they are read
        for p, o, ptr in zip(self.params, self.offsets, self._ptrs):
            if p.data_ptr() != ptr:
                raise RuntimeError("parameter storage moved after the optimiser was built")
            g = p.grad
            if g is None:
                # torch skips such parameters; on this path every parameter of the reference models is back-propagated
                # every step, so a missing gradient is a bug in the caller: fail loudly
                raise RuntimeError("a parameter has no gradient; the reference always back-props every parameter")
            if g.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                # someone replaced .grad (e.g. zero_grad(set_to_none) from foreign code): fold it back
                slot = self.flat_grad[o:o + p.numel()].view_as(p)
                slot.copy_(g)
                p.grad = slot
        self.sync_lr()
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        check(lib.migan_adam_step(self._tab.data_ptr(), self._blk.data_ptr(), self._nblocks, self.step_t.data_ptr(),
                                  self._ticket.data_ptr(), self.lr_t.data_ptr(), float(g0["lr"]), float(b1), float(b2),
                                  float(g0["eps"]), float(grad_scale), torch.cuda.current_stream().cuda_stream),
              "adam_step")
        self.bump_epoch()

    def bump_epoch(self):
        """Renew the weight-epoch stamp of every parameter (invalidates packed-weight cache entries)."""
        ep = next(_EPOCH)
        for p in self.params:
            p._migan_epoch = ep

    # ---- torch.optim.Adam-compatible checkpoints (cyclegan.py:73-78 resumes from state_dicts) -----------------
    def state_dict(self):
        """Same structure as torch.optim.Adam.state_dict(): {'state': {i: {step, exp_avg, exp_avg_sq}}, 'param_groups'}."""
        state = {}
        step = self.step_t.detach().clone().reshape(())
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {"step": step.clone(), "exp_avg": self.exp_avg[o:o + n].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).clone()}
        g0 = self.param_groups[0]
        group = {"lr": g0["lr"], "betas": g0["betas"], "eps": g0["eps"], "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("optimizer state has %d parameters in %d groups, this optimiser has %d in 1"
                             % (sum(len(g["params"]) for g in groups), len(groups), len(self.params)))
        g = groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("weight_decay / amsgrad / maximize are not on the reference path")
        steps = set()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(i)
            n = p.numel()
            if st is None:  # torch: parameter never stepped
                self.exp_avg[o:o + n].zero_()
                self.exp_avg_sq[o:o + n].zero_()
                continue
            for k in ("exp_avg", "exp_avg_sq"):
                if tuple(st[k].shape) != tuple(p.shape):
                    raise ValueError("optimizer state %s of parameter %d has shape %s, expected %s"
                                     % (k, i, tuple(st[k].shape), tuple(p.shape)))
            self.exp_avg[o:o + n].view_as(p).copy_(st["exp_avg"])
            self.exp_avg_sq[o:o + n].view_as(p).copy_(st["exp_avg_sq"])
            steps.add(float(st["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ (%s); this optimiser keeps one counter" % sorted(steps))
        self.step_t.fill_(steps.pop() if steps else 0.0)
        self.param_groups[0].update(lr=float(g["lr"]), betas=tuple(g["betas"]), eps=float(g["eps"]))
        self.sync_lr()


def sync_all_lr():
    """Called by graph.StepRunner before a replay: captured Adam launches read the learning rate from the device."""
    for opt in list(Adam._live):
        opt.sync_lr()
