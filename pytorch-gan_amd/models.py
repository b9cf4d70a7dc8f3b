"""The reference networks defined directly on the HIP layer set.

These are the `nn.Module` trees of implementations/{dcgan,wgan_gp,gan,cyclegan,pix2pix,srgan,esrgan} written against
`pytorch_gan_amd.nn` (the drop-in for `torch.nn`), so constructing them needs no `swap()`.  Attribute names,
layer order and constructor arguments follow the reference files cited per class, hence `state_dict()` keys and
shapes are interchangeable with the reference's checkpoints (cyclegan.py:73-78,279-284).  A model built from
the reference's own source + `pytorch_gan_amd.swap()` is equivalent; this module exists so that the package is
usable (bench, examples) on machines without the reference checkout.
"""
import math

import torch

from . import functional as F
from . import nn


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
        return nn.Dropout2d(a[0])
    if kind == "drop":
        return nn.Dropout(a[0])
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
        a, b = F.fork2(x)   # the block's input has two consumers: their gradients meet in the library's add, not in autograd's
        return self.block(a, res=b)   # x + self.block(x), the add inside the last InstanceNorm launch


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
        d, cur = [x], x
        for i in range(1, 9):
            cur = getattr(self, "down%d" % i)(cur)
            if i < 8:   # d1 .. d7 feed the next level AND a skip connection (pix2pix/models.py:84-98)
                cur, skip = F.fork2(cur)
                d.append(skip)
            else:
                d.append(cur)
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
        a, b = F.fork2(x)
        return self.conv_block(a, res=b)   # x + self.conv_block(x), the add inside the last BatchNorm launch


class SrganGenerator(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, n_residual_blocks=16):
        super().__init__()
        self.conv1 = _build([("conv", in_channels, 64, 9, 1, 4), ("prelu",)])
        self.res_blocks = nn.Sequential(*[SrganResidualBlock(64) for _ in range(n_residual_blocks)])
        self.conv2 = _build([("conv", 64, 64, 3, 1, 1), ("bn2", 64, 0.8)])
        self.upsampling = _build([("conv", 64, 256, 3, 1, 1), ("bn2", 256), ("shuffle", 2), ("prelu",)] * 2)
        self.conv3 = _build([("conv", 64, out_channels, 9, 1, 4), ("tanh",)])

    def forward(self, x):
        o1, o1s = F.fork2(self.conv1(x))
        o2 = self.conv2(self.res_blocks(o1))
        return self.conv3(self.upsampling(torch.add(o1s, o2)))


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
    """torchvision vgg19 cfg 'E' features, children [:n]; n = 35 ends at conv5_4 before its ReLU (esrgan/models.py:12)."""
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
    """esrgan/models.py:18-45 DenseResidualBlock; `out.mul(res_scale) + x` is one axpby launch here."""

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
            if i < 5:
                inputs = torch.cat([inputs, out], 1)
        return nn._wrap(F.axpby(out, x, self.res_scale, 1.0))


class EsrganRRDB(nn.Module):
    """esrgan/models.py:48-57 ResidualInResidualDenseBlock."""

    def __init__(self, filters, res_scale=0.2):
        super().__init__()
        self.res_scale = res_scale
        self.dense_blocks = nn.Sequential(EsrganDenseBlock(filters), EsrganDenseBlock(filters), EsrganDenseBlock(filters))

    def forward(self, x):
        return nn._wrap(F.axpby(self.dense_blocks(x), x, self.res_scale, 1.0))


class EsrganGenerator(nn.Module):
    """esrgan/models.py:60-94 GeneratorRRDB (also the network of the inference entry point test_on_image.py:24-37)."""

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
    """esrgan/models.py:97-130: layer for layer the SRGAN discriminator; its output is used as logits
    (BCEWithLogitsLoss on relativistic differences, esrgan.py:137,165-166)."""


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
    """acgan/acgan.py:76-107: DCGAN discriminator blocks, a Sigmoid validity head and a Softmax class head."""

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

