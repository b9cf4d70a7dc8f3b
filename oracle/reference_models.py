"""ORACLE (test infrastructure, not product code): CPU restatement of the reference networks.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product path (pytorch-gan_amd/) never does.

Arithmetic: everything here runs on stock `torch.nn` layers on the CPU (the reference's own third-party
arithmetic: torch, unpinned in requirements.txt:1; this image: torch 2.10.0 CPU path, oneDNN/MKL).  The
restatement is pinned against the real reference classes by oracle/pin_against_reference.py, which
imports /root/reference, checks state_dict keys/shapes, seeded-init equality and bit-equal
forward/backward, and writes the golden fixtures under tests/golden/ (see that script's header).

Each network is described as a flat layer spec and materialised by `_build`; attribute names follow the
reference so that `state_dict()` keys are identical:
  dcgan      implementations/dcgan/dcgan.py:45-99
  wgan_gp    implementations/wgan_gp/wgan_gp.py:42-83        gan  implementations/gan/gan.py:38-81
  cyclegan   implementations/cyclegan/models.py:22-122
  pix2pix    implementations/pix2pix/models.py:20-133
  srgan      implementations/srgan/models.py:8-105 (VGG19 cfg-E features[:18], random init: torchvision and
             its pretrained weights are unavailable offline — SURVEY.md §8c)
"""
import math
import random

import torch
import torch.nn as nn


# --------------------------------------------------------------------------------------------- dropout masks
class _MaskFeed:
    """Host-drawn dropout masks shared by the oracle and the HIP path (CPU and GPU generators differ)."""

    queue = None
    record = None


class feed_masks:
    def __init__(self, masks=None, record=None):
        self.masks, self.rec = masks, record

    def __enter__(self):
        self.prev = (_MaskFeed.queue, _MaskFeed.record)
        _MaskFeed.queue = list(self.masks) if self.masks is not None else None
        _MaskFeed.record = self.rec

    def __exit__(self, *a):
        _MaskFeed.queue, _MaskFeed.record = self.prev


def _draw(shape, p, like):
    if _MaskFeed.queue is not None:
        m = torch.as_tensor(_MaskFeed.queue.pop(0), dtype=like.dtype)
        assert tuple(m.shape) == tuple(shape), (m.shape, shape)
    else:
        m = torch.bernoulli(torch.full(shape, 1.0 - p)).div_(1.0 - p)
    if _MaskFeed.record is not None:
        _MaskFeed.record.append(m.clone())
    return m


class PlaneDropout(nn.Dropout2d):
    """nn.Dropout2d with an injectable mask: y = x * m[n,c] (m already scaled by 1/(1-p))."""

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        m = _draw((x.shape[0], x.shape[1]), self.p, x)
        return x * m[:, :, None, None]


class ElementDropout(nn.Dropout):
    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        return x * _draw(tuple(x.shape), self.p, x)


# --------------------------------------------------------------------------------------------- spec builder
def _layer(spec):
    kind, a = spec[0], spec[1:]
    if kind == "conv":      # cin, cout, k, stride, pad, [bias]
        return nn.Conv2d(a[0], a[1], a[2], a[3], a[4], bias=(a[5] if len(a) > 5 else True))
    if kind == "convT":     # cin, cout, k, stride, pad, [bias]
        return nn.ConvTranspose2d(a[0], a[1], a[2], a[3], a[4], bias=(a[5] if len(a) > 5 else True))
    if kind == "lin":
        return nn.Linear(a[0], a[1])
    if kind == "bn2":       # channels, [eps]  (second positional arg of BatchNorm IS eps)
        return nn.BatchNorm2d(*a)
    if kind == "bn1":
        return nn.BatchNorm1d(*a)
    if kind == "in2":
        return nn.InstanceNorm2d(a[0])
    if kind == "lrelu":
        return nn.LeakyReLU(a[0], inplace=(a[1] if len(a) > 1 else False))
    if kind == "relu":
        return nn.ReLU(inplace=True)
    if kind == "prelu":
        return nn.PReLU()
    if kind == "tanh":
        return nn.Tanh()
    if kind == "sigmoid":
        return nn.Sigmoid()
    if kind == "up2":
        return nn.Upsample(scale_factor=2)
    if kind == "rpad":
        return nn.ReflectionPad2d(a[0])
    if kind == "zpad":
        return nn.ZeroPad2d(a[0])
    if kind == "shuffle":
        return nn.PixelShuffle(upscale_factor=a[0])
    if kind == "pool":
        return nn.MaxPool2d(kernel_size=2, stride=2)
    if kind == "drop2":
        return PlaneDropout(a[0])
    if kind == "drop":
        return ElementDropout(a[0])
    if kind == "cres":      # constructed lazily so parameter-init RNG draws happen in reference order
        return CycleResidualBlock(a[0])
    raise KeyError(kind)


def _build(specs):
    return nn.Sequential(*[_layer(s) for s in specs])


# --------------------------------------------------------------------------------------------- init
def init_normal_dcgan(m):
    """dcgan.py:36-42 / pix2pix/models.py:6-12: *Conv* weight ~ N(0,.02); BatchNorm2d weight ~ N(1,.02), bias 0."""
    name = type(m).__name__
    if "Conv" in name:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif "BatchNorm2d" in name:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0.0)


def init_normal_cyclegan(m):
    """cyclegan/models.py:6-14: as above, plus conv bias = 0."""
    name = type(m).__name__
    if "Conv" in name:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
        if getattr(m, "bias", None) is not None:
            nn.init.constant_(m.bias.data, 0.0)
    elif "BatchNorm2d" in name:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0.0)


# --------------------------------------------------------------------------------------------- dcgan
class DcganGenerator(nn.Module):
    def __init__(self, img_size=32, latent_dim=100, channels=1):
        super().__init__()
        self.init_size = img_size // 4
        self.l1 = _build([("lin", latent_dim, 128 * self.init_size ** 2)])
        self.conv_blocks = _build([
            ("bn2", 128), ("up2",), ("conv", 128, 128, 3, 1, 1), ("bn2", 128, 0.8), ("lrelu", 0.2, True),
            ("up2",), ("conv", 128, 64, 3, 1, 1), ("bn2", 64, 0.8), ("lrelu", 0.2, True),
            ("conv", 64, channels, 3, 1, 1), ("tanh",)])

    def forward(self, z):
        h = self.l1(z)
        return self.conv_blocks(h.view(h.shape[0], 128, self.init_size, self.init_size))


class DcganDiscriminator(nn.Module):
    def __init__(self, img_size=32, channels=1):
        super().__init__()
        specs, cin = [], channels
        for cout, bn in ((16, False), (32, True), (64, True), (128, True)):
            specs += [("conv", cin, cout, 3, 2, 1), ("lrelu", 0.2, True), ("drop2", 0.25)]
            if bn:
                specs.append(("bn2", cout, 0.8))
            cin = cout
        self.model = _build(specs)
        self.adv_layer = _build([("lin", 128 * (img_size // 16) ** 2, 1), ("sigmoid",)])

    def forward(self, img):
        h = self.model(img)
        return self.adv_layer(h.view(h.shape[0], -1))


# --------------------------------------------------------------------------------------------- wgan_gp / gan (MLP)
def _mlp_generator_specs(latent_dim, out_features):
    specs = [("lin", latent_dim, 128), ("lrelu", 0.2, True)]
    for cin, cout in ((128, 256), (256, 512), (512, 1024)):
        specs += [("lin", cin, cout), ("bn1", cout, 0.8), ("lrelu", 0.2, True)]
    return specs + [("lin", 1024, out_features), ("tanh",)]


class MlpGenerator(nn.Module):
    """wgan_gp.py:42-65 and gan.py:38-61 (identical apart from img_shape)."""

    def __init__(self, img_shape=(1, 32, 32), latent_dim=100):
        super().__init__()
        self.img_shape = tuple(img_shape)
        self.model = _build(_mlp_generator_specs(latent_dim, int(math.prod(img_shape))))

    def forward(self, z):
        img = self.model(z)
        return img.view(img.shape[0], *self.img_shape)


class MlpCritic(nn.Module):
    """wgan_gp.py:68-83 (no sigmoid); gan.py:64-81 adds a Sigmoid (sigmoid=True)."""

    def __init__(self, img_shape=(1, 32, 32), sigmoid=False):
        super().__init__()
        d = int(math.prod(img_shape))
        specs = [("lin", d, 512), ("lrelu", 0.2, True), ("lin", 512, 256), ("lrelu", 0.2, True), ("lin", 256, 1)]
        if sigmoid:
            specs.append(("sigmoid",))
        self.model = _build(specs)

    def forward(self, img):
        return self.model(img.view(img.shape[0], -1))


# --------------------------------------------------------------------------------------------- cyclegan
class CycleResidualBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.block = _build([("rpad", 1), ("conv", c, c, 3, 1, 0), ("in2", c), ("relu",),
                             ("rpad", 1), ("conv", c, c, 3, 1, 0), ("in2", c)])

    def forward(self, x):
        return x + self.block(x)


class CycleGenerator(nn.Module):
    def __init__(self, input_shape=(3, 256, 256), num_residual_blocks=9):
        super().__init__()
        ch = input_shape[0]
        specs = [("rpad", ch), ("conv", ch, 64, 7, 1, 0), ("in2", 64), ("relu",)]
        c = 64
        for _ in range(2):
            specs += [("conv", c, 2 * c, 3, 2, 1), ("in2", 2 * c), ("relu",)]
            c *= 2
        specs += [("cres", c)] * num_residual_blocks
        for _ in range(2):
            specs += [("up2",), ("conv", c, c // 2, 3, 1, 1), ("in2", c // 2), ("relu",)]
            c //= 2
        specs += [("rpad", ch), ("conv", c, ch, 7, 1, 0), ("tanh",)]
        self.model = _build(specs)

    def forward(self, x):
        return self.model(x)


def _patch_blocks(cin, first_norm):
    specs, c = [], cin
    for i, cout in enumerate((64, 128, 256, 512)):
        specs.append(("conv", c, cout, 4, 2, 1))
        if i > 0 or first_norm:
            specs.append(("in2", cout))
        specs.append(("lrelu", 0.2, True))
        c = cout
    return specs


class CycleDiscriminator(nn.Module):
    def __init__(self, input_shape=(3, 256, 256)):
        super().__init__()
        ch, h, w = input_shape
        self.output_shape = (1, h // 16, w // 16)
        self.model = _build(_patch_blocks(ch, False) + [("zpad", (1, 0, 1, 0)), ("conv", 512, 1, 4, 1, 1)])

    def forward(self, img):
        return self.model(img)


class ReplayBuffer:
    """cyclegan/utils.py:13-33 — per-sample history of generated images (python `random`)."""

    def __init__(self, max_size=50):
        assert max_size > 0
        self.max_size, self.data = max_size, []

    def push_and_pop(self, batch):
        out = []
        for sample in batch.data:
            sample = sample.unsqueeze(0)
            if len(self.data) < self.max_size:
                self.data.append(sample)
                out.append(sample)
            elif random.uniform(0, 1) > 0.5:
                k = random.randint(0, self.max_size - 1)
                out.append(self.data[k].clone())
                self.data[k] = sample
            else:
                out.append(sample)
        return torch.cat(out)


def lambda_lr(n_epochs, offset, decay_start_epoch):
    """cyclegan/utils.py:36-44."""
    assert n_epochs - decay_start_epoch > 0
    return lambda epoch: 1.0 - max(0, epoch + offset - decay_start_epoch) / (n_epochs - decay_start_epoch)


# --------------------------------------------------------------------------------------------- pix2pix
class UNetDown(nn.Module):
    def __init__(self, cin, cout, normalize=True, dropout=0.0):
        super().__init__()
        specs = [("conv", cin, cout, 4, 2, 1, False)]
        if normalize:
            specs.append(("in2", cout))
        specs.append(("lrelu", 0.2))
        if dropout:
            specs.append(("drop", dropout))
        self.model = _build(specs)

    def forward(self, x):
        return self.model(x)


class UNetUp(nn.Module):
    def __init__(self, cin, cout, dropout=0.0):
        super().__init__()
        specs = [("convT", cin, cout, 4, 2, 1, False), ("in2", cout), ("relu",)]
        if dropout:
            specs.append(("drop", dropout))
        self.model = _build(specs)

    def forward(self, x, skip):
        return torch.cat((self.model(x), skip), 1)


class Pix2pixGenerator(nn.Module):
    def __init__(self, in_channels=3, out_channels=3):
        super().__init__()
        downs = [(in_channels, 64, False, 0.0), (64, 128, True, 0.0), (128, 256, True, 0.0), (256, 512, True, 0.5),
                 (512, 512, True, 0.5), (512, 512, True, 0.5), (512, 512, True, 0.5), (512, 512, False, 0.5)]
        for i, (ci, co, nrm, dr) in enumerate(downs, 1):
            setattr(self, "down%d" % i, UNetDown(ci, co, normalize=nrm, dropout=dr))
        ups = [(512, 512, 0.5), (1024, 512, 0.5), (1024, 512, 0.5), (1024, 512, 0.5), (1024, 256, 0.0),
               (512, 128, 0.0), (256, 64, 0.0)]
        for i, (ci, co, dr) in enumerate(ups, 1):
            setattr(self, "up%d" % i, UNetUp(ci, co, dropout=dr))
        self.final = _build([("up2",), ("zpad", (1, 0, 1, 0)), ("conv", 128, out_channels, 4, 1, 1), ("tanh",)])

    def forward(self, x):
        d = [x]
        for i in range(1, 9):
            d.append(getattr(self, "down%d" % i)(d[-1]))
        u = d[8]
        for i in range(1, 8):
            u = getattr(self, "up%d" % i)(u, d[8 - i])
        return self.final(u)


class Pix2pixDiscriminator(nn.Module):
    def __init__(self, in_channels=3):
        super().__init__()
        self.model = _build(_patch_blocks(in_channels * 2, False)
                            + [("zpad", (1, 0, 1, 0)), ("conv", 512, 1, 4, 1, 1, False)])

    def forward(self, img_A, img_B):
        return self.model(torch.cat((img_A, img_B), 1))


# --------------------------------------------------------------------------------------------- srgan
def vgg19_features_18():
    """torchvision vgg19 cfg 'E' features, children [:18] = conv1_1 ... relu3_4 (srgan/models.py:11-12)."""
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return layers[:18]


class SrganFeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        self.feature_extractor = nn.Sequential(*vgg19_features_18())

    def forward(self, img):
        return self.feature_extractor(img)


class SrganResidualBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv_block = _build([("conv", c, c, 3, 1, 1), ("bn2", c, 0.8), ("prelu",),
                                  ("conv", c, c, 3, 1, 1), ("bn2", c, 0.8)])

    def forward(self, x):
        return x + self.conv_block(x)


class SrganGenerator(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, n_residual_blocks=16):
        super().__init__()
        self.conv1 = _build([("conv", in_channels, 64, 9, 1, 4), ("prelu",)])
        self.res_blocks = nn.Sequential(*[SrganResidualBlock(64) for _ in range(n_residual_blocks)])
        self.conv2 = _build([("conv", 64, 64, 3, 1, 1), ("bn2", 64, 0.8)])
        self.upsampling = _build([("conv", 64, 256, 3, 1, 1), ("bn2", 256), ("shuffle", 2), ("prelu",)] * 2)
        self.conv3 = _build([("conv", 64, out_channels, 9, 1, 4), ("tanh",)])

    def forward(self, x):
        o1 = self.conv1(x)
        o2 = self.conv2(self.res_blocks(o1))
        return self.conv3(self.upsampling(torch.add(o1, o2)))


class SrganDiscriminator(nn.Module):
    def __init__(self, input_shape=(3, 384, 384)):
        super().__init__()
        self.input_shape = input_shape
        c, h, w = input_shape
        self.output_shape = (1, int(h / 16), int(w / 16))
        specs, cin = [], c
        for i, cout in enumerate((64, 128, 256, 512)):
            specs.append(("conv", cin, cout, 3, 1, 1))
            if i != 0:
                specs.append(("bn2", cout))
            specs += [("lrelu", 0.2, True), ("conv", cout, cout, 3, 2, 1), ("bn2", cout), ("lrelu", 0.2, True)]
            cin = cout
        specs.append(("conv", cin, 1, 3, 1, 1))
        self.model = _build(specs)

    def forward(self, img):
        return self.model(img)


# --------------------------------------------------------------------------------------------- esrgan
def vgg19_features(n):
    """torchvision vgg19 cfg 'E' features, children [:n]; n = 35 ends at conv5_4 BEFORE its ReLU (esrgan/models.py:12)."""
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return layers[:n]


class EsrganFeatureExtractor(nn.Module):
    """esrgan/models.py:8-15."""

    def __init__(self):
        super().__init__()
        self.vgg19_54 = nn.Sequential(*vgg19_features(35))

    def forward(self, img):
        return self.vgg19_54(img)


class EsrganDenseBlock(nn.Module):
    """esrgan/models.py:18-45: five 3x3 convs over the growing channel concatenation; LeakyReLU() (slope 0.01) on the
    first four; out * res_scale + x."""

    def __init__(self, filters, res_scale=0.2):
        super().__init__()
        self.res_scale = res_scale
        for i in range(1, 6):
            specs = [("conv", i * filters, filters, 3, 1, 1)] + ([("lrelu", 0.01)] if i < 5 else [])
            setattr(self, "b%d" % i, _build(specs))

    def forward(self, x):
        inputs = x
        for i in range(1, 6):
            out = getattr(self, "b%d" % i)(inputs)
            inputs = torch.cat([inputs, out], 1)
        return out.mul(self.res_scale) + x


class EsrganRRDB(nn.Module):
    """esrgan/models.py:48-57."""

    def __init__(self, filters, res_scale=0.2):
        super().__init__()
        self.res_scale = res_scale
        self.dense_blocks = nn.Sequential(EsrganDenseBlock(filters), EsrganDenseBlock(filters), EsrganDenseBlock(filters))

    def forward(self, x):
        return self.dense_blocks(x).mul(self.res_scale) + x


class EsrganGenerator(nn.Module):
    """esrgan/models.py:60-94 GeneratorRRDB."""

    def __init__(self, channels=3, filters=64, num_res_blocks=16, num_upsample=2):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, filters, kernel_size=3, stride=1, padding=1)
        self.res_blocks = nn.Sequential(*[EsrganRRDB(filters) for _ in range(num_res_blocks)])
        self.conv2 = nn.Conv2d(filters, filters, kernel_size=3, stride=1, padding=1)
        self.upsampling = _build([("conv", filters, filters * 4, 3, 1, 1), ("lrelu", 0.01), ("shuffle", 2)] * num_upsample)
        self.conv3 = _build([("conv", filters, filters, 3, 1, 1), ("lrelu", 0.01), ("conv", filters, channels, 3, 1, 1)])

    def forward(self, x):
        out1 = self.conv1(x)
        out2 = self.conv2(self.res_blocks(out1))
        return self.conv3(self.upsampling(torch.add(out1, out2)))


class EsrganDiscriminator(SrganDiscriminator):
    """esrgan/models.py:97-130: layer for layer the SRGAN discriminator (3x3 s1 / s2 pairs, BatchNorm2d, LeakyReLU(0.2),
    3x3 -> 1 patch logits); kept as its own class so class-name matches and state_dict keys follow the reference file."""


# --------------------------------------------------------------------------------------------- stargan / dualgan critics (8f F1)
class StarganDiscriminator(nn.Module):
    """stargan/models.py:87-115: n_strided x [Conv4x4 s2 p1, LeakyReLU(0.01)], then a 3x3 patch head and a class head."""

    def __init__(self, img_shape=(3, 128, 128), c_dim=5, n_strided=6):
        super().__init__()
        channels, img_size, _ = img_shape
        specs, cin, cout = [], channels, 64
        for _ in range(n_strided):
            specs += [("conv", cin, cout, 4, 2, 1), ("lrelu", 0.01)]
            cin, cout = cout, cout * 2
        self.model = _build(specs)
        self.out1 = nn.Conv2d(cin, 1, 3, padding=1, bias=False)
        self.out2 = nn.Conv2d(cin, c_dim, img_size // 2 ** n_strided, bias=False)

    def forward(self, img):
        f = self.model(img)
        out_cls = self.out2(f)
        return self.out1(f), out_cls.view(out_cls.size(0), -1)


class DualganDiscriminator(nn.Module):
    """dualgan/models.py:102-123: 3 x [Conv4x4 s2 p1, (BatchNorm2d(C, 0.8)), LeakyReLU(0.2)], ZeroPad2d((1,0,1,0)), Conv4x4."""

    def __init__(self, in_channels=3):
        super().__init__()
        specs, cin = [], in_channels
        for i, cout in enumerate((64, 128, 256)):
            specs.append(("conv", cin, cout, 4, 2, 1))
            if i != 0:
                specs.append(("bn2", cout, 0.8))
            specs.append(("lrelu", 0.2, True))
            cin = cout
        specs += [("zpad", (1, 0, 1, 0)), ("conv", 256, 1, 4, 1, 0)]
        self.model = _build(specs)

    def forward(self, img):
        return self.model(img)


# --------------------------------------------------------------------------------------------- acgan
class AcganGenerator(nn.Module):
    """acgan/acgan.py:46-73: the DCGAN generator behind label_emb(labels) * noise."""

    def __init__(self, img_size=32, latent_dim=100, channels=1, n_classes=10):
        super().__init__()
        self.label_emb = nn.Embedding(n_classes, latent_dim)
        self.init_size = img_size // 4
        self.l1 = _build([("lin", latent_dim, 128 * self.init_size ** 2)])
        self.conv_blocks = _build([
            ("bn2", 128), ("up2",), ("conv", 128, 128, 3, 1, 1), ("bn2", 128, 0.8), ("lrelu", 0.2, True),
            ("up2",), ("conv", 128, 64, 3, 1, 1), ("bn2", 64, 0.8), ("lrelu", 0.2, True),
            ("conv", 64, channels, 3, 1, 1), ("tanh",)])

    def forward(self, noise, labels):
        out = self.l1(torch.mul(self.label_emb(labels), noise))
        return self.conv_blocks(out.view(out.shape[0], 128, self.init_size, self.init_size))


class AcganDiscriminator(nn.Module):
    """acgan/acgan.py:76-107: DCGAN discriminator blocks, a Sigmoid validity head and a Softmax class head
    (nn.Softmax() without dim: the legacy implicit choice for a 2-D input is dim 1)."""

    def __init__(self, img_size=32, channels=1, n_classes=10):
        super().__init__()
        specs, cin = [], channels
        for cout, bn in ((16, False), (32, True), (64, True), (128, True)):
            specs += [("conv", cin, cout, 3, 2, 1), ("lrelu", 0.2, True), ("drop2", 0.25)]
            if bn:
                specs.append(("bn2", cout, 0.8))
            cin = cout
        self.conv_blocks = _build(specs)
        feat = 128 * (img_size // 16) ** 2
        self.adv_layer = _build([("lin", feat, 1), ("sigmoid",)])
        self.aux_layer = nn.Sequential(nn.Linear(feat, n_classes), nn.Softmax(dim=1))

    def forward(self, img):
        out = self.conv_blocks(img)
        out = out.view(out.shape[0], -1)
        return self.adv_layer(out), self.aux_layer(out)



# --------------------------------------------------------------------------------------------- DCGAN-block clones
# lsgan.py:45-90, sgan.py:46-107, infogan.py:58-121, relativistic_gan.py:37-91, cogan.py:51-122, began.py:47-99,
# ebgan.py:47-101: the dcgan.py generator / discriminator blocks with other heads, inputs and losses.  Sub-modules are built
# in the reference's order (seeded parameter init consumes the RNG in construction order) and carry its attribute names
# (state_dict keys).  Every class is checked bit-for-bit against the reference's own class by oracle/pin_against_reference.py.
def _g_conv_specs(channels, first_bn):
    return ([("bn2", 128)] if first_bn else []) + [
        ("up2",), ("conv", 128, 128, 3, 1, 1), ("bn2", 128, 0.8), ("lrelu", 0.2, True),
        ("up2",), ("conv", 128, 64, 3, 1, 1), ("bn2", 64, 0.8), ("lrelu", 0.2, True),
        ("conv", 64, channels, 3, 1, 1), ("tanh",)]


def _d_block_specs(channels, bn_first=False):
    """dcgan-style discriminator blocks; bn_first: cogan.py:94-99 puts BatchNorm in front of LeakyReLU / Dropout2d."""
    specs, cin = [], channels
    for cout, bn in ((16, False), (32, True), (64, True), (128, True)):
        specs.append(("conv", cin, cout, 3, 2, 1))
        if bn and bn_first:
            specs.append(("bn2", cout, 0.8))
        specs += [("lrelu", 0.2, True), ("drop2", 0.25)]
        if bn and not bn_first:
            specs.append(("bn2", cout, 0.8))
        cin = cout
    return specs


class CloneGenerator(nn.Module):
    """lsgan.py:45-67 / ebgan.py:47-71 (first_bn=False), relativistic_gan.py:37-62 / began.py:47-72 (first_bn=True, = dcgan's),
    sgan.py:46-74 (label_emb: an nn.Embedding the forward never uses, but a parameter of the optimiser and the first consumer
    of the init RNG), infogan.py:58-85 (extra = n_classes + code_dim inputs concatenated behind the noise)."""

    def __init__(self, img_size=32, latent_dim=100, channels=1, first_bn=True, label_emb=0, extra=0):
        super().__init__()
        if label_emb:
            self.label_emb = nn.Embedding(label_emb, latent_dim)
        self.init_size = img_size // 4
        self.l1 = _build([("lin", latent_dim + extra, 128 * self.init_size ** 2)])
        self.conv_blocks = _build(_g_conv_specs(channels, first_bn))

    def forward(self, noise, labels=None, code=None):
        if labels is not None:
            noise = torch.cat((noise, labels, code), -1)
        out = self.l1(noise)
        return self.conv_blocks(out.view(out.shape[0], 128, self.init_size, self.init_size))


class CloneDiscriminator(nn.Module):
    """The dcgan.py:77-92 blocks under the reference's attribute names with its heads:
       lsgan.py:72-90         blocks='model',       heads (('adv_layer', 1, None, False),)            plain nn.Linear
       relativistic_gan.py:65-91  blocks='model',   heads (('adv_layer', 1, None, True),)             Sequential(Linear)
       sgan.py:76-107         blocks='conv_blocks', heads adv_layer -> Sigmoid, aux_layer (classes + 1) -> Softmax
       infogan.py:88-121      blocks='conv_blocks', heads adv_layer, aux_layer -> Softmax, latent_layer
    A head is (name, out_features, 'sigmoid' | 'softmax' | None, wrapped_in_Sequential)."""

    def __init__(self, img_size=32, channels=1, blocks="model", heads=(("adv_layer", 1, None, False),)):
        super().__init__()
        setattr(self, blocks, _build(_d_block_specs(channels)))
        self.blocks_name, self.head_names = blocks, [h[0] for h in heads]
        feat = 128 * (img_size // 16) ** 2
        for name, nout, act, seq in heads:
            lin = nn.Linear(feat, nout)
            if act == "sigmoid":
                mod = nn.Sequential(lin, nn.Sigmoid())
            elif act == "softmax":
                mod = nn.Sequential(lin, nn.Softmax(dim=1))  # nn.Softmax() on a 2-D input = dim 1
            else:
                mod = nn.Sequential(lin) if seq else lin
            setattr(self, name, mod)

    def forward(self, img):
        out = getattr(self, self.blocks_name)(img)
        out = out.view(out.shape[0], -1)
        outs = tuple(getattr(self, n)(out) for n in self.head_names)
        return outs[0] if len(outs) == 1 else outs


class CoganGenerators(nn.Module):
    """cogan.py:51-87: shared trunk, two image heads."""

    def __init__(self, img_size=32, latent_dim=100, channels=3):
        super().__init__()
        self.init_size = img_size // 4
        self.fc = _build([("lin", latent_dim, 128 * self.init_size ** 2)])
        self.shared_conv = _build([("bn2", 128), ("up2",), ("conv", 128, 128, 3, 1, 1), ("bn2", 128, 0.8),
                                   ("lrelu", 0.2, True), ("up2",)])
        tail = [("conv", 128, 64, 3, 1, 1), ("bn2", 64, 0.8), ("lrelu", 0.2, True), ("conv", 64, channels, 3, 1, 1), ("tanh",)]
        self.G1 = _build(tail)
        self.G2 = _build(tail)

    def forward(self, noise):
        out = self.fc(noise)
        emb = self.shared_conv(out.view(out.shape[0], 128, self.init_size, self.init_size))
        return self.G1(emb), self.G2(emb)


class CoganDiscriminators(nn.Module):
    """cogan.py:90-122: shared conv blocks (BatchNorm in front of the activation), one linear head per domain."""

    def __init__(self, img_size=32, channels=3):
        super().__init__()
        self.shared_conv = _build(_d_block_specs(channels, bn_first=True))
        feat = 128 * (img_size // 16) ** 2
        self.D1 = nn.Linear(feat, 1)
        self.D2 = nn.Linear(feat, 1)

    def forward(self, img1, img2):
        o1 = self.shared_conv(img1)
        v1 = self.D1(o1.view(o1.shape[0], -1))
        o2 = self.shared_conv(img2)
        v2 = self.D2(o2.view(o2.shape[0], -1))
        return v1, v2


def init_normal_cogan(m):
    """cogan.py:42-48: *Linear* weight ~ N(0,.02); *BatchNorm* weight ~ N(1,.02), bias 0 (convs keep the default init)."""
    name = type(m).__name__
    if "Linear" in name:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif "BatchNorm" in name:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0.0)


class AutoencoderDiscriminator(nn.Module):
    """began.py:75-99 (embedding_attr=False: the bottleneck Linear is fc[0]) and ebgan.py:74-101 (embedding_attr=True: it is
    `self.embedding`, and forward also returns the 32-d code): Conv s2 + ReLU, Linear -> BatchNorm1d(32, 0.8) -> ReLU ->
    Linear -> BatchNorm1d -> ReLU, Upsample + Conv back to the image."""

    def __init__(self, img_size=32, channels=1, embedding_attr=False):
        super().__init__()
        self.down = nn.Sequential(nn.Conv2d(channels, 64, 3, 2, 1), nn.ReLU())
        self.down_size = img_size // 2
        down_dim = 64 * self.down_size ** 2
        self.embedding_attr = embedding_attr
        if embedding_attr:
            self.embedding = nn.Linear(down_dim, 32)
            self.fc = nn.Sequential(nn.BatchNorm1d(32, 0.8), nn.ReLU(inplace=True), nn.Linear(32, down_dim),
                                    nn.BatchNorm1d(down_dim), nn.ReLU(inplace=True))
        else:
            self.fc = nn.Sequential(nn.Linear(down_dim, 32), nn.BatchNorm1d(32, 0.8), nn.ReLU(inplace=True),
                                    nn.Linear(32, down_dim), nn.BatchNorm1d(down_dim), nn.ReLU(inplace=True))
        self.up = nn.Sequential(nn.Upsample(scale_factor=2), nn.Conv2d(64, channels, 3, 1, 1))

    def forward(self, img):
        out = self.down(img)
        flat = out.view(out.size(0), -1)
        if self.embedding_attr:
            emb = self.embedding(flat)
            out = self.fc(emb)
        else:
            out = self.fc(flat)
        out = self.up(out.view(out.size(0), 64, self.down_size, self.down_size))
        return (out, emb) if self.embedding_attr else out


def clone_models(name, img_size=32):
    """(G, D, init function or None) of one F2 script with the reference's defaults."""
    if name == "lsgan":
        return CloneGenerator(img_size, 100, 1, first_bn=False), CloneDiscriminator(img_size, 1), init_normal_dcgan
    if name == "sgan":
        return (CloneGenerator(img_size, 100, 1, label_emb=10),
                CloneDiscriminator(img_size, 1, "conv_blocks", (("adv_layer", 1, "sigmoid", True), ("aux_layer", 11, "softmax", True))),
                init_normal_dcgan)
    if name == "infogan":
        return (CloneGenerator(img_size, 62, 1, extra=12),
                CloneDiscriminator(img_size, 1, "conv_blocks", (("adv_layer", 1, None, True), ("aux_layer", 10, "softmax", True),
                                                                ("latent_layer", 2, None, True))),
                init_normal_dcgan)
    if name == "relativistic_gan":
        return CloneGenerator(img_size, 100, 1), CloneDiscriminator(img_size, 1, "model", (("adv_layer", 1, None, True),)), None
    if name == "cogan":
        return CoganGenerators(img_size, 100, 3), CoganDiscriminators(img_size, 3), init_normal_cogan
    if name == "began":
        return CloneGenerator(img_size, 62, 1), AutoencoderDiscriminator(img_size, 1, False), init_normal_dcgan
    if name == "ebgan":
        return CloneGenerator(img_size, 62, 1, first_bn=False), AutoencoderDiscriminator(img_size, 1, True), init_normal_dcgan
    raise KeyError(name)
