"""Drop-in `torch.nn` layer set for the reference GAN model definitions, backed by libmigan.so.

Usage mirrors the reference scripts:

    import pytorch_gan_amd.nn as nn          # instead of `import torch.nn as nn`
    nn.Conv2d(128, 128, 3, stride=1, padding=1); nn.BatchNorm2d(128, 0.8); nn.LeakyReLU(0.2, inplace=True) ...

or, for a model that was already built from stock torch.nn layers (the reference's `Generator()`,
`GeneratorResNet(...)`, ...):  `swap(model)` re-classes every supported leaf in place.  Parameters, buffers,
`state_dict` keys/shapes, class-name substrings used by `weights_init_normal` (dcgan.py:36-42) and the
module tree are unchanged.

`Sequential` applies run-time peephole fusion over its unchanged child list:
  [Upsample(2)] [ZeroPad2d|ReflectionPad2d] Conv2d [LeakyReLU|ReLU|Tanh|Sigmoid]  -> one conv launch
  BatchNorm/InstanceNorm [LeakyReLU|ReLU]                                           -> stats + one apply pass
"""
import contextlib

import torch
import torch.nn as tnn

from . import functional as F

_FUSE = True
_RELU_HANDOFF = True   # conv, ReLU, conv | MaxPool2d: the consumer applies the ReLU backward (tests flip it for the A/B comparison)


_PRELU_FUSE = True     # False = PReLU as its own launches
_SHUFFLE_FUSE = True   # False = PixelShuffle as its own launches


def set_fusion(enabled):
    """Enable/disable Sequential peephole fusion (both modes give the same results up to fp32 rounding)."""
    global _FUSE
    _FUSE = bool(enabled)


# ----------------------------------------------------------------------------------------------- GanTensor
class GanTensor(torch.Tensor):
    """Marks a channels_last activation produced by this package.

    The reference model code mixes raw tensor ops with layers: `out.view(B, -1)` (dcgan.py:96),
    `x + self.block(x)` (cyclegan/models.py:37), `torch.add(out1, out2)` (srgan/models.py:68),
    `torch.cat((x, skip), 1)` (pix2pix/models.py:50,132).  This subclass routes exactly those calls to
    the HIP kernels (and makes `.view` legal on NHWC storage); every other op runs unchanged and
    returns a plain tensor.
    """

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        with torch._C.DisableTorchFunctionSubclass():
            handler = _OVERRIDES.get(func)
            if handler is not None:
                out = handler(args, kwargs)
                if out is not NotImplemented:
                    return out
            return func(*args, **kwargs)


def _is_act4(t):
    return isinstance(t, torch.Tensor) and t.dim() == 4 and F.on_device(t) and t.dtype == torch.float32


def _h_view(func):
    def handler(args, kwargs):
        x = args[0]
        if _is_act4(x) and not x.is_contiguous():
            return func(F.relayout(x, False), *args[1:], **kwargs)
        return NotImplemented

    return handler


def _h_add(args, kwargs):
    if len(args) != 2 or kwargs:
        return NotImplemented
    a, b = args
    if _is_act4(a) and _is_act4(b) and a.shape == b.shape:
        return _wrap(F.add(a, b))
    return NotImplemented


def _h_mul(args, kwargs):
    if len(args) != 2 or kwargs:
        return NotImplemented
    a, b = args
    ok = lambda t: isinstance(t, torch.Tensor) and F.on_device(t) and t.dtype == torch.float32  # noqa: E731
    if ok(a) and ok(b) and a.shape == b.shape:
        return _wrap(F.mul(a, b))
    if ok(b) and isinstance(a, (int, float)):
        a, b = b, a
    if ok(a) and isinstance(b, (int, float)) and not isinstance(b, bool):  # out.mul(self.res_scale), esrgan/models.py:45,57
        return _wrap(F.axpby(a, None, float(b), 0.0))
    return NotImplemented


def _h_cat(args, kwargs):
    tensors = args[0]
    dim = args[1] if len(args) > 1 else kwargs.get("dim", 0)
    if dim != 1 or len(tensors) < 2 or not all(_is_act4(t) for t in tensors):
        return NotImplemented
    out = tensors[0]
    for t in tensors[1:]:
        out = F.cat_channels(out, t)
    return _wrap(out)


_OVERRIDES = {
    torch.Tensor.view: _h_view(torch.Tensor.view),
    torch.Tensor.reshape: _h_view(torch.Tensor.reshape),
    torch.Tensor.flatten: _h_view(torch.Tensor.flatten),
    torch.flatten: _h_view(torch.flatten),
    torch.reshape: _h_view(torch.reshape),
    torch.Tensor.add: _h_add,
    torch.Tensor.__add__: _h_add,
    torch.Tensor.__radd__: _h_add,
    torch.add: _h_add,
    torch.mul: _h_mul,
    torch.Tensor.mul: _h_mul,
    torch.Tensor.__mul__: _h_mul,
    torch.Tensor.__rmul__: _h_mul,
    torch.cat: _h_cat,
}


def _wrap(y):
    if type(y) is torch.Tensor and y.dim() == 4 and not y.is_contiguous():
        return y.as_subclass(GanTensor)
    return y


# ----------------------------------------------------------------------------------------------- helpers
def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def _act_of(m):
    """(act code, slope) for activation modules that can ride in a conv / norm epilogue."""
    if isinstance(m, tnn.LeakyReLU):
        return F.ACT_LRELU, float(m.negative_slope)
    if isinstance(m, tnn.ReLU):
        return F.ACT_RELU, 0.0
    if isinstance(m, tnn.Tanh):
        return F.ACT_TANH, 0.0
    if isinstance(m, tnn.Sigmoid):
        return F.ACT_SIGMOID, 0.0
    return None


# ----------------------------------------------------------------------------------------------- layers
class Conv2d(tnn.Conv2d):
    def _check(self):
        if self.groups != 1 or _pair(self.dilation) != (1, 1) or self.padding_mode != "zeros":
            raise ValueError("Conv2d: groups/dilation/padding_mode outside the reference path")
        s = _pair(self.stride)
        if s[0] != s[1] or s[0] not in (1, 2):
            raise ValueError("Conv2d: stride must be 1 or 2")
        if isinstance(self.padding, str):
            raise ValueError("Conv2d: string padding is not supported")
        return s[0], _pair(self.padding)

    def fused_forward(self, x, pre_pads=(0, 0, 0, 0), gather=F.GATHER_ZERO, act=F.ACT_NONE, slope=0.0, dropout=None,
                      stats=None, relu_in=False, handed=False):
        """`dropout`: a following training-mode Dropout2d module whose mask multiply rides in the conv epilogue.
        `stats` ("batch" / "instance"): a normalisation layer consumes this output next - its statistics are taken in the
        conv epilogue (only where everything in between is fused into this launch)."""
        stride, (ph, pw) = self._check()
        if gather == F.GATHER_REFLECT and (ph or pw):
            raise ValueError("reflection gather cannot be combined with conv zero padding")
        pads = (pre_pads[0] + ph, pre_pads[1] + pw, pre_pads[2] + ph, pre_pads[3] + pw)
        if (gather == F.GATHER_UP2 and stride == 1 and pads == (1, 1, 1, 1)
                and tuple(self.weight.shape[2:]) == (3, 3)):
            # phase-collapsed Upsample+Conv3x3
            if relu_in or handed:
                raise ValueError("the ReLU hand-off (Sequential) never reaches an Upsample+Conv pair")
            y = _wrap(F.upconv3x3(x, self.weight, self.bias, act, slope, stats if dropout is None else None))
            return dropout(y) if dropout is not None else y
        if dropout is not None and self.out_channels % 4 == 0:
            mask = _next_mask((x.shape[0], self.out_channels), dropout.p, x.device)
            return _wrap(F.conv2d(x, self.weight, self.bias, stride, pads, gather, act, slope, mask, stats, relu_in, handed))
        y = _wrap(F.conv2d(x, self.weight, self.bias, stride, pads, gather, act, slope, None,
                           stats if dropout is None else None, relu_in, handed))
        return dropout(y) if dropout is not None else y

    def forward(self, x):
        return self.fused_forward(x)


class ConvTranspose2d(tnn.ConvTranspose2d):
    def forward(self, x, output_size=None, act=F.ACT_NONE, slope=0.0):
        s, p = _pair(self.stride), _pair(self.padding)
        if (self.groups != 1 or _pair(self.dilation) != (1, 1) or _pair(self.output_padding) != (0, 0)
                or output_size is not None or s[0] != s[1] or p[0] != p[1] or s[0] not in (1, 2)):
            raise ValueError("ConvTranspose2d: configuration outside the reference path")
        return _wrap(F.conv_transpose2d(x, self.weight, self.bias, s[0], p[0], act, slope))


class Linear(tnn.Linear):
    def forward(self, x, act=F.ACT_NONE, slope=0.0):
        return F.linear(x, self.weight, self.bias, act, slope)


class _BatchNormMixin:
    def fused_forward(self, x, act=F.ACT_NONE, slope=0.0, res=None, prelu=None, shuffle=0):
        if self.momentum is None:
            raise ValueError("BatchNorm: cumulative moving average (momentum=None) is not on the reference path")
        use_batch = self.training or not self.track_running_stats
        rm = self.running_mean if self.track_running_stats else None
        rv = self.running_var if self.track_running_stats else None
        nbt = None
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            nbt = self.num_batches_tracked  # incremented by the statistics kernel (no separate aten::add launch)
        y = F.norm(x, self.weight if self.affine else None, self.bias if self.affine else None, res,
                   rm if (self.training or not use_batch) else None, rv if (self.training or not use_batch) else None,
                   use_batch, self.momentum, self.eps, False, act, slope, nbt, prelu, shuffle)
        return _wrap(y)

    def forward(self, x):
        self._check_input_dim(x)
        return self.fused_forward(x)


class BatchNorm2d(_BatchNormMixin, tnn.BatchNorm2d):
    pass


class BatchNorm1d(_BatchNormMixin, tnn.BatchNorm1d):
    def _check_input_dim(self, x):
        if x.dim() != 2:
            raise ValueError("BatchNorm1d: expected (B, C) input on this path, got %dD" % x.dim())


class InstanceNorm2d(tnn.InstanceNorm2d):
    def fused_forward(self, x, act=F.ACT_NONE, slope=0.0, res=None, mask=None):
        if self.affine or self.track_running_stats:
            raise ValueError("InstanceNorm2d: affine/track_running_stats are not on the reference path")
        if x.dim() != 4:
            raise ValueError("InstanceNorm2d: expected 4D input")
        if x.shape[2] * x.shape[3] == 1 and self.training:
            raise ValueError("Expected more than 1 spatial element when training, got input size %s" % (x.shape,))
        return _wrap(F.norm(x, None, None, res, None, None, True, 0.1, self.eps, True, act, slope, mask=mask))

    def forward(self, x):
        return self.fused_forward(x)


class LeakyReLU(tnn.LeakyReLU):
    def forward(self, x):
        return _wrap(F.activation(x, F.ACT_LRELU, self.negative_slope))


class ReLU(tnn.ReLU):
    def forward(self, x):
        return _wrap(F.activation(x, F.ACT_RELU))


class Tanh(tnn.Tanh):
    def forward(self, x):
        return _wrap(F.activation(x, F.ACT_TANH))


class Sigmoid(tnn.Sigmoid):
    def forward(self, x):
        return _wrap(F.activation(x, F.ACT_SIGMOID))


class PReLU(tnn.PReLU):
    def forward(self, x):
        return _wrap(F.prelu(x, self.weight))


class Upsample(tnn.Upsample):
    def _check(self):
        sf = self.scale_factor
        sf = sf if not isinstance(sf, (tuple, list)) else (sf[0] if sf[0] == sf[1] else None)
        if self.size is not None or sf is None or float(sf) != 2.0 or self.mode != "nearest":
            raise ValueError("Upsample: only scale_factor=2, mode='nearest' is on the reference path")

    def forward(self, x):
        self._check()
        return _wrap(F.gather2d(x, (0, 0, 0, 0), F.GATHER_UP2))


def _pads_tlbr(padding):
    """torch pad order (left, right, top, bottom) -> (top, left, bottom, right)."""
    if isinstance(padding, int):
        return (padding,) * 4
    l, r, t, b = padding
    return (int(t), int(l), int(b), int(r))


class ReflectionPad2d(tnn.ReflectionPad2d):
    def forward(self, x):
        return _wrap(F.gather2d(x, _pads_tlbr(self.padding), F.GATHER_REFLECT))


class ZeroPad2d(tnn.ZeroPad2d):
    def forward(self, x):
        return _wrap(F.gather2d(x, _pads_tlbr(self.padding), F.GATHER_ZERO))


class PixelShuffle(tnn.PixelShuffle):
    def forward(self, x):
        return _wrap(F.pixel_shuffle(x, self.upscale_factor))


class MaxPool2d(tnn.MaxPool2d):
    def forward(self, x, relu_in=False):
        if (_pair(self.kernel_size) != (2, 2) or _pair(self.stride) != (2, 2) or _pair(self.padding) != (0, 0)
                or _pair(self.dilation) != (1, 1) or self.ceil_mode or self.return_indices):
            raise ValueError("MaxPool2d: only kernel 2 / stride 2 (VGG19) is on the reference path")
        return _wrap(F.maxpool2(x, relu_in))


# ---- dropout: device Philox stream for training runs, injected host masks for parity tests ---------
class _DropoutRNG:
    seed = 0x5EED
    counters = {}
    injected = None  # list of host/device masks consumed in call order (parity tests)

    @classmethod
    def counter(cls, device):
        key = str(device)
        if key not in cls.counters:
            cls.counters[key] = torch.zeros(2, dtype=torch.int64, device=device)  # [stream position, ticket]
        return cls.counters[key]


def manual_seed(seed):
    """Seed the device dropout stream (independent of torch's generators)."""
    _DropoutRNG.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    for c in _DropoutRNG.counters.values():
        c.zero_()


@contextlib.contextmanager
def dropout_masks(masks):
    """Feed pre-drawn masks (already scaled by 1/(1-p)) to the next Dropout/Dropout2d calls, in order."""
    prev = _DropoutRNG.injected
    _DropoutRNG.injected = list(masks)
    try:
        yield
    finally:
        _DropoutRNG.injected = prev


@contextlib.contextmanager
def paired_batches(module):
    """`module(cat(a, b))` inside this context stands for `module(a)` followed by `module(b)` (steps.dcgan_step runs the
    discriminator on real and generated images as one batch): BatchNorm layers keep per-half statistics
    (functional.batch_groups(2)), and masks injected with dropout_masks() - which arrive in the reference's call order,
    all layers of the first forward, then all layers of the second - are re-paired layer by layer."""
    inj = _DropoutRNG.injected
    if inj is not None:
        k = sum(1 for m in module.modules() if isinstance(m, (tnn.Dropout, tnn.Dropout2d)) and m.training and m.p > 0.0)
        if len(inj) < 2 * k:
            raise RuntimeError("dropout_masks(): a paired forward needs %d masks, %d left" % (2 * k, len(inj)))
        first, second = inj[:k], inj[k:2 * k]
        inj[:2 * k] = [torch.cat([torch.as_tensor(a, dtype=torch.float32), torch.as_tensor(b, dtype=torch.float32)])
                       for a, b in zip(first, second)]
    with F.batch_groups(2):
        yield


class _MaskPlan:
    """All dropout masks of one training step from ONE launch.  A step (functional.weight_cache_scope: one `steps.*_step`
    body) asks for the same masks in the same order every iteration - dcgan.py:77-80: four Dropout2d(0.25) layers x three
    discriminator forwards - so the sequence recorded in one step becomes the plan of the next: its first request draws
    every mask of the plan into one flat buffer (one Philox launch instead of twelve), later requests are views.  A
    request that departs from the plan (other shape / p, different p values in one step) falls back to its own launch."""

    plans = {}   # device -> instance

    def __init__(self):
        self.scope, self.seq, self.plan, self.buf, self.offsets, self.pos = None, [], None, None, None, 0

    @classmethod
    def get(cls, device):
        plans, key = F._owner_plans(cls.plans), ("mask", str(device))   # one plan per step state (weight_cache_scope(owner))
        if key not in plans:
            plans[key] = cls()
        return plans[key]

    def next(self, shape, p, device):
        scope = F._CACHE_SCOPE
        if scope is None or not _BATCH_MASKS:
            return None
        shape = tuple(int(v) for v in shape)
        if scope != self.scope:  # a new step: what the last one asked for is the plan
            if self.scope is not None and self.seq and len({q for _, q in self.seq}) == 1 \
                    and all(_numel(sh) % 4 == 0 for sh, _ in self.seq):
                self.plan = tuple(self.seq)
            self.scope, self.seq, self.buf, self.pos = scope, [], None, 0
            if self.plan is not None and self.plan[0] == (shape, p):
                self.offsets, total = [], 0
                for sh, _ in self.plan:
                    self.offsets.append(total)
                    total += _numel(sh)
                self.buf = F.rand_mask((total,), p, _DropoutRNG.seed, _DropoutRNG.counter(device), device)
        self.seq.append((shape, p))
        if self.buf is None:
            return None
        if self.pos < len(self.plan) and self.plan[self.pos] == (shape, p):
            o = self.offsets[self.pos]
            self.pos += 1
            return self.buf[o:o + _numel(shape)].view(shape)
        self.buf = None  # departed from the plan: individual launches for the rest of this step
        return None


def _numel(shape):
    n = 1
    for v in shape:
        n *= int(v)
    return n


_BATCH_MASKS = True
_DROPOUT_FUSE = True   # nn.Dropout inside the small InstanceNorm launch


def _next_mask(shape, p, device):
    if _DropoutRNG.injected is not None:
        if not _DropoutRNG.injected:
            raise RuntimeError("dropout_masks(): more dropout calls than injected masks")
        m = _DropoutRNG.injected.pop(0)
        m = torch.as_tensor(m, dtype=torch.float32).to(device)
        if tuple(m.shape) != tuple(shape):
            raise ValueError("injected dropout mask has shape %s, expected %s" % (tuple(m.shape), tuple(shape)))
        return m
    m = _MaskPlan.get(device).next(shape, float(p), device)
    if m is None:
        m = F.rand_mask(shape, p, _DropoutRNG.seed, _DropoutRNG.counter(device), device)
    if len(shape) == 4:
        # a DRAWN mask is i.i.d.: read the same numbers as NHWC, the layout of the activations it multiplies - as a contiguous
        # NCHW tensor every nn.Dropout call paid a transpose launch for it (8 per pix2pix step; injected masks keep their layout)
        N, C, H, W = shape
        m = m.view(N, H, W, C).permute(0, 3, 1, 2)
    return m


class Dropout2d(tnn.Dropout2d):
    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        if x.dim() != 4:
            raise ValueError("Dropout2d: expected 4D input")
        mask = _next_mask((x.shape[0], x.shape[1]), self.p, x.device)
        return _wrap(F.mul_mask(x, mask))


class Dropout(tnn.Dropout):
    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        mask = _next_mask(tuple(x.shape), self.p, x.device)
        return _wrap(F.mul_mask(x, mask))


# ---- losses (torch.nn.BCELoss / MSELoss / L1Loss with the default 'mean' reduction) ------------------
class _MeanLoss(tnn.Module):
    kind = None

    def __init__(self, reduction="mean"):
        super().__init__()
        if reduction != "mean":
            raise ValueError("only reduction='mean' is on the reference path")

    def forward(self, x, target):
        if target.requires_grad:
            raise ValueError("loss target must not require grad (detach it, as the reference does)")
        return F.loss(self.kind, x, target)


class BCELoss(_MeanLoss):
    kind = F.LOSS_BCE


class MSELoss(_MeanLoss):
    kind = F.LOSS_MSE


class L1Loss(_MeanLoss):
    kind = F.LOSS_L1


class BCEWithLogitsLoss(_MeanLoss):
    """relativistic_gan.py:95 (no weight / pos_weight, mean reduction)."""
    kind = F.LOSS_BCE_LOGITS

    def __init__(self, weight=None, reduction="mean", pos_weight=None):
        super().__init__(reduction)
        if weight is not None or pos_weight is not None:
            raise ValueError("BCEWithLogitsLoss: weight / pos_weight are not on the reference path")


class CrossEntropyLoss(tnn.Module):
    """acgan.py:113 `torch.nn.CrossEntropyLoss()`: class-index targets, mean reduction, no weights / smoothing."""

    def __init__(self, weight=None, ignore_index=-100, reduction="mean", label_smoothing=0.0):
        super().__init__()
        if weight is not None or reduction != "mean" or label_smoothing != 0.0 or ignore_index != -100:
            raise ValueError("CrossEntropyLoss: only the default configuration is on the reference path")

    def forward(self, x, target):
        return F.cross_entropy(x, target)


# ---- layers of the DCGAN-block clones ---------------------------------------------------------------
class Embedding(tnn.Embedding):
    def forward(self, idx):
        if self.padding_idx is not None or self.max_norm is not None or self.sparse or self.scale_grad_by_freq:
            raise ValueError("Embedding: padding_idx / max_norm / sparse / scale_grad_by_freq are not on the reference path")
        y = F.embedding(idx, self.weight)
        return y.as_subclass(GanTensor) if type(y) is torch.Tensor else y  # routes `torch.mul(emb, noise)` (acgan.py:61)


class Softmax(tnn.Softmax):
    def forward(self, x):
        if x.dim() != 2 or self.dim not in (None, 1, -1):  # nn.Softmax() on (B, classes): implicit dim = 1
            raise ValueError("Softmax: only the class dimension of a (B, classes) tensor is on the reference path")
        return F.softmax(x)


# ----------------------------------------------------------------------------------------------- Sequential
class Sequential(tnn.Sequential):
    """nn.Sequential with run-time peephole fusion over the unchanged child list."""

    def forward(self, x, res=None):
        """`res` (extension, used by pytorch_gan_amd.models): a residual added to the output - `x + self.block(x)` of
        cyclegan/models.py:37 and srgan/models.py:31 - inside the final normalisation launch when the chain ends in one."""
        mods = list(self._modules.values())
        if not _FUSE:
            for m in mods:
                x = m(x)
            return x if res is None else x + res
        i, n = 0, len(mods)
        # conv, ReLU, conv | MaxPool2d (vgg19.features[:18], srgan/models.py:8-15): the ReLU backward of the first conv is applied by
        # its consumer - in the epilogue of the second conv's input-gradient launch, or inside the pool's backward - instead of a
        # pass of its own.  relu_hand: x is the output of a conv+ReLU that skips its own ReLU backward; the next module MUST take it.
        relu_hand = False
        while i < n:
            m = mods[i]
            if relu_hand and type(m) is MaxPool2d:
                x, relu_hand, i = m(x, relu_in=True), False, i + 1
                continue
            # -- [Upsample] [ZeroPad2d | ReflectionPad2d] Conv2d [act] ------------------------------------
            j, gather, pre = i, F.GATHER_ZERO, (0, 0, 0, 0)
            if isinstance(mods[j], Upsample) and j + 1 < n:
                mods[j]._check()
                gather, j = F.GATHER_UP2, j + 1
            if j < n and isinstance(mods[j], ZeroPad2d) and j + 1 < n:
                pre, j = _pads_tlbr(mods[j].padding), j + 1
            elif j < n and isinstance(mods[j], ReflectionPad2d) and gather == F.GATHER_ZERO and j + 1 < n:
                pre, gather, j = _pads_tlbr(mods[j].padding), F.GATHER_REFLECT, j + 1
            if j < n and isinstance(mods[j], Conv2d) and not (gather == F.GATHER_REFLECT and any(_pair(mods[j].padding))):
                act, slope, k = F.ACT_NONE, 0.0, j + 1
                if k < n and _act_of(mods[k]) is not None and type(mods[k]) in _OURS:
                    (act, slope), k = _act_of(mods[k]), k + 1
                drop = None
                if k < n and type(mods[k]) is Dropout2d and mods[k].training and 0.0 < mods[k].p < 1.0 and x.dim() == 4:
                    drop, k = mods[k], k + 1
                stats = None  # the next module normalises this output: statistics from the conv epilogue
                if k < n and type(mods[k]) is InstanceNorm2d:
                    stats = "instance"
                elif k < n and type(mods[k]) is BatchNorm2d and (mods[k].training or not mods[k].track_running_stats):
                    stats = "batch"
                rin, relu_hand = relu_hand, False
                if (_RELU_HANDOFF and act == F.ACT_RELU and drop is None and stats is None and k < n and x.dim() == 4
                        and type(mods[k]) in (Conv2d, MaxPool2d)):
                    relu_hand = True    # mods[k] is next in this loop: a Conv2d reached with no pad / upsample module in front, or the pool
                x = mods[j].fused_forward(x, pre, gather, act, slope, drop, stats, rin, relu_hand)
                i = k
                continue
            if relu_hand:
                raise RuntimeError("Sequential: a handed-off ReLU backward was not taken")
            # -- Linear [LeakyReLU | ReLU | Tanh | Sigmoid] ------------------------------------------------
            if type(m) is Linear and i + 1 < n and _act_of(mods[i + 1]) is not None and type(mods[i + 1]) in _OURS \
                    and x.dim() == 2:
                act, slope = _act_of(mods[i + 1])
                x = m(x, act, slope)
                i += 2
                continue
            # -- Norm [LeakyReLU | ReLU] -------------------------------------------------------------------
            if isinstance(m, (BatchNorm2d, BatchNorm1d, InstanceNorm2d)):
                act, slope, k = F.ACT_NONE, 0.0, i + 1
                if k < n and isinstance(mods[k], (LeakyReLU, ReLU)):
                    (act, slope), k = _act_of(mods[k]), k + 1
                if isinstance(m, (BatchNorm2d, BatchNorm1d)):
                    m._check_input_dim(x)
                # BatchNorm2d(64, 0.8), LeakyReLU(0.2), Conv2d(64, 1, 3, 1, 1), Tanh (dcgan.py:60-62): the image-output conv reads the
                # BatchNorm input through the normalisation; its backward recomputes the conv's input gradient from the one-channel dz
                cv = mods[k] if (k < n and type(mods[k]) is Conv2d) else None
                if (cv is not None and type(m) is BatchNorm2d and m.training and m.momentum is not None and res is None
                        and act in (F.ACT_NONE, F.ACT_LRELU, F.ACT_RELU) and cv.padding_mode == "zeros"
                        and not isinstance(cv.padding, str) and _pair(cv.stride) == (1, 1)
                        and F.bn_act_conv1_takes(x, cv.weight, 1, _pair(cv.padding) * 2, _pair(cv.dilation), cv.groups)):
                    oact, oslope, kk = F.ACT_NONE, 0.0, k + 1
                    if kk < n and _act_of(mods[kk]) is not None and type(mods[kk]) in _OURS:
                        (oact, oslope), kk = _act_of(mods[kk]), kk + 1
                    trk = m.track_running_stats
                    x = _wrap(F.bn_act_conv1(x, m.weight if m.affine else None, m.bias if m.affine else None,
                                             m.running_mean if trk else None, m.running_var if trk else None,
                                             m.num_batches_tracked if trk else None, m.momentum, m.eps, act, slope, cv.weight, cv.bias,
                                             oact, oslope))
                    i = kk
                    continue
                # BatchNorm2d [PixelShuffle] PReLU (srgan/models.py:23-24, 55-57): the single-slope PReLU commutes with the
                # shuffle, so it is applied (and differentiated) inside the norm launches and the shuffle moves behind it
                if _PRELU_FUSE and act == F.ACT_NONE and type(m) is BatchNorm2d and x.dim() == 4 \
                        and (m.training or not m.track_running_stats) \
                        and F._BN_GROUPS == 1:   # paired batches (functional.batch_groups): the plain BatchNorm path, PReLU / shuffle as their own launches
                    q = k + 1 if (k < n and type(mods[k]) is PixelShuffle) else k
                    if q < n and type(mods[q]) is PReLU and mods[q].num_parameters == 1:
                        cv = mods[q + 1] if (q == k and q + 1 < n and type(mods[q + 1]) is Conv2d) else None
                        if (cv is not None and m.training and m.momentum is not None and cv.padding_mode == "zeros"
                                and not isinstance(cv.padding, str) and _pair(cv.stride) == (1, 1)
                                and F.bn_prelu_conv64_takes(x, cv.weight, 1, _pair(cv.padding) * 2, _pair(cv.dilation), cv.groups)):
                            # BatchNorm2d(64, 0.8), PReLU(), Conv2d(64, 64, 3, 1, 1) (srgan/models.py:23-25): the conv reads the block's
                            # first conv output through the normalisation and the PReLU - the tensor between them is never stored
                            trk = m.track_running_stats
                            x = _wrap(F.bn_prelu_conv64(x, m.weight if m.affine else None, m.bias if m.affine else None,
                                                        m.running_mean if trk else None, m.running_var if trk else None,
                                                        m.num_batches_tracked if trk else None, m.momentum, m.eps, mods[q].weight,
                                                        cv.weight, cv.bias))
                            i = q + 2
                            continue
                        if q > k and mods[k].upscale_factor == 2 and x.shape[1] % 4 == 0 and _SHUFFLE_FUSE:
                            x = m.fused_forward(x, F.ACT_NONE, 0.0, None, mods[q].weight, 2)   # shuffle = store index map
                        else:
                            x = m.fused_forward(x, F.ACT_NONE, 0.0, None, mods[q].weight)
                            if q > k:
                                x = mods[k](x)
                        i = q + 1
                        continue
                if res is not None and k == n and act == F.ACT_NONE and x.dim() == 4 and res.shape == x.shape:
                    x, res = m.fused_forward(x, act, slope, res), None   # y = norm(x) + res in the apply kernel
                elif _DROPOUT_FUSE and type(m) is InstanceNorm2d and k < n and type(mods[k]) is Dropout and mods[k].training \
                        and 0.0 < mods[k].p < 1.0 and x.dim() == 4 and F.norm_small_takes(x, True):
                    # InstanceNorm2d [LeakyReLU | ReLU] Dropout (pix2pix/models.py:25-28,41-45): the mask multiplies inside the
                    # one-launch normalisation of the small inner U-Net levels, forward and backward
                    x = m.fused_forward(x, act, slope, None, _next_mask(tuple(x.shape), mods[k].p, x.device))
                    k += 1
                else:
                    x = m.fused_forward(x, act, slope)
                i = k
                continue
            x = m(x)
            i += 1
        return x if res is None else x + res


_SWAP = {
    tnn.Conv2d: Conv2d, tnn.ConvTranspose2d: ConvTranspose2d, tnn.Linear: Linear, tnn.BatchNorm2d: BatchNorm2d,
    tnn.BatchNorm1d: BatchNorm1d, tnn.InstanceNorm2d: InstanceNorm2d, tnn.LeakyReLU: LeakyReLU, tnn.ReLU: ReLU,
    tnn.Tanh: Tanh, tnn.Sigmoid: Sigmoid, tnn.PReLU: PReLU, tnn.Upsample: Upsample,
    tnn.ReflectionPad2d: ReflectionPad2d, tnn.ZeroPad2d: ZeroPad2d, tnn.PixelShuffle: PixelShuffle,
    tnn.MaxPool2d: MaxPool2d, tnn.Dropout: Dropout, tnn.Dropout2d: Dropout2d, tnn.Sequential: Sequential,
    tnn.BCELoss: BCELoss, tnn.MSELoss: MSELoss, tnn.L1Loss: L1Loss, tnn.BCEWithLogitsLoss: BCEWithLogitsLoss,
    tnn.CrossEntropyLoss: CrossEntropyLoss, tnn.Embedding: Embedding, tnn.Softmax: Softmax,
}
_OURS = set(_SWAP.values())


def swap(module):
    """Re-class every supported stock torch.nn leaf (and Sequential container) of `module` in place.

    Unsupported layer types are left untouched only if they hold no parameters and are containers of the
    reference (Generator, ResidualBlock, ...); an unsupported *leaf* raises, so a swapped model can never
    silently run a layer on ATen.
    """
    for m in module.modules():
        cls = type(m)
        if cls in _SWAP:
            if cls in (tnn.BCELoss, tnn.MSELoss, tnn.L1Loss, tnn.BCEWithLogitsLoss, tnn.CrossEntropyLoss):
                if m.reduction != "mean":
                    raise ValueError("swap: loss reduction %r is not supported" % m.reduction)
                if getattr(m, "weight", None) is not None or getattr(m, "pos_weight", None) is not None \
                        or getattr(m, "label_smoothing", 0.0) != 0.0 or getattr(m, "ignore_index", -100) != -100:
                    raise ValueError("swap: loss weights / label smoothing / ignore_index are not on the reference path")
            m.__class__ = _SWAP[cls]
        elif cls in _OURS:
            continue
        elif isinstance(m, tnn.Dropout2d):   # e.g. the oracle's mask-injectable subclasses
            m.__class__ = Dropout2d
        elif isinstance(m, tnn.Dropout) and type(m).forward is not tnn.Dropout.forward:
            m.__class__ = Dropout
        elif cls.__module__.startswith("torch.nn") and len(list(m.children())) == 0:
            raise NotImplementedError("swap: no HIP implementation for leaf layer %s" % cls.__name__)
    return module


# re-exports so that `import pytorch_gan_amd.nn as nn` covers what the reference model files touch
Module = tnn.Module
ModuleList = tnn.ModuleList
Parameter = tnn.Parameter
init = tnn.init
