"""Input pipeline on the device: the per-image work of the reference's DataLoader workers between the decoded
uint8 bitmap and the fp32 batch, as three streaming HIP kernels over a batch of equally sized images.

Reference call sites: cyclegan.py:111-117 (Resize(int(h*1.12), Image.BICUBIC), RandomCrop, RandomHorizontalFlip, ToTensor,
Normalize((.5,.5,.5),(.5,.5,.5))), srgan/datasets.py:16-33 (Resize((h/4, h/4), BICUBIC) and Resize((h, h), BICUBIC) of every image,
Normalize(ImageNet mean/std)), dcgan.py:120-131 (Resize(img_size) of MNIST, bilinear), pix2pix/datasets.py (BICUBIC resize, flip).

Results are bit-exact with Pillow's resize + torchvision's transforms (tests/test_data_gpu.py): the resample kernels do
Pillow's 8-bit fixed-point arithmetic (csrc/image_pipeline.hip), and the coefficient rows - a function of the sizes only - are
computed here on the host exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do, once per (in, out, filter).
Decoding (JPEG/PNG -> uint8) stays on the host; there is no CPU fallback for the rest."""
import math

import numpy as np
import torch

from ._lib import check, lib

PRECISION_BITS = 32 - 8 - 2  # Pillow Resample.c: 8-bit pixels, 2 bits of head room for negative lobes


def _bicubic(x, a=-0.5):  # Pillow bicubic_filter (Keys, a = -0.5), support 2
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):  # Pillow bilinear_filter, support 1
    if x < 0.0:
        x = -x
    return 1.0 - x if x < 1.0 else 0.0


FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}
_COEFFS = {}


def pil_resample_coeffs(in_size, out_size, filt):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for a whole-image box:
    (kk int32 [out][ksize], bounds int32 [out][2] = (first source index, taps), ksize).  Plain Python floats are C doubles and
    the operation order is Pillow's, so the fixed-point coefficients are identical."""
    key = (int(in_size), int(out_size), filt)
    if key in _COEFFS:
        return _COEFFS[key]
    fn, support = FILTERS[filt]
    in0, in1 = 0.0, float(in_size)
    scale = (in1 - in0) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        for x in range(xmax):
            w = fn((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                k[x] /= ww
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
        for x in range(ksize):
            v = k[x]
            kk[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
    _COEFFS[key] = (kk, bounds, ksize)
    return _COEFFS[key]


_DEV_COEFFS = {}


def _device_coeffs(in_size, out_size, filt, device):
    key = (int(in_size), int(out_size), filt, str(device))
    if key not in _DEV_COEFFS:
        kk, bounds, ksize = pil_resample_coeffs(in_size, out_size, filt)
        _DEV_COEFFS[key] = (torch.from_numpy(kk).to(device), torch.from_numpy(bounds).to(device), ksize)
    return _DEV_COEFFS[key]


def resize_output_size(size, height, width):
    """torchvision.transforms.Resize semantics: an int matches the SMALLER edge and keeps the aspect ratio
    (cyclegan.py:112, dcgan.py:126), a pair is (h, w) (srgan/datasets.py:21,28).  Returns (out_h, out_w)."""
    if isinstance(size, int):
        short, long_ = (width, height) if width <= height else (height, width)
        new_short, new_long = size, int(size * long_ / short)
        return (new_long, new_short) if width <= height else (new_short, new_long)
    h, w = size
    return int(h), int(w)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check_u8(images):
    from .functional import on_device

    if not (isinstance(images, torch.Tensor) and on_device(images) and images.dtype == torch.uint8 and images.dim() == 4
            and images.is_contiguous()):
        raise TypeError("expected a contiguous uint8 CUDA tensor [N, H, W, C] (decoded bitmaps, as np.asarray(PIL image) lays them out)")
    if not 1 <= images.shape[3] <= 4:
        raise ValueError("1..4 channels per pixel")


def resize_u8(images, size, filt="bicubic"):
    """PIL.Image.resize((w, h), BICUBIC | BILINEAR) of every image of a uint8 batch [N, H, W, C] on the device (bit-exact).
    Like Pillow: the horizontal pass first, its output rounded to uint8, then the vertical pass; a pass whose size does not
    change is skipped."""
    _check_u8(images)
    if filt not in FILTERS:
        raise ValueError("filter must be one of %s" % sorted(FILTERS))
    N, H, W, C = images.shape
    oh, ow = resize_output_size(size, H, W)
    if oh < 1 or ow < 1:
        raise ValueError("empty output size")
    cur = images
    if ow != W:
        kk, bounds, ksize = _device_coeffs(W, ow, filt, images.device)
        nxt = torch.empty((N, H, ow, C), dtype=torch.uint8, device=images.device)
        check(lib.migan_resample_u8(cur.data_ptr(), nxt.data_ptr(), kk.data_ptr(), bounds.data_ptr(), ksize, N, H, W, C, ow, 1,
                                    _stream()), "resample_u8(h)")
        cur = nxt
    if oh != H:
        kk, bounds, ksize = _device_coeffs(H, oh, filt, images.device)
        nxt = torch.empty((N, oh, ow, C), dtype=torch.uint8, device=images.device)
        check(lib.migan_resample_u8(cur.data_ptr(), nxt.data_ptr(), kk.data_ptr(), bounds.data_ptr(), ksize, N, H, ow, C, oh, 0,
                                    _stream()), "resample_u8(v)")
        cur = nxt
    return cur


def to_float(images, crop=None, corners=None, flip=None, mean=None, std=None, channels_last=True):
    """crop window + horizontal flip + ToTensor + Normalize in one launch: uint8 [N, H, W, C] -> fp32 logical [N, C, h, w].
    corners: int32 [N, 2] (top, left) per image or None; flip: uint8/bool [N] or None; mean/std: sequences of C floats or None.
    channels_last=True returns NHWC storage behind NCHW strides (what the first conv consumes without a re-layout)."""
    _check_u8(images)
    N, H, W, C = images.shape
    h, w = (H, W) if crop is None else (int(crop[0]), int(crop[1]))
    if h > H or w > W:
        raise ValueError("Required crop size (%d, %d) is larger than input image size (%d, %d)" % (h, w, H, W))
    dev = images.device
    cptr = fptr = mptr = sptr = None
    if corners is not None:
        host = torch.as_tensor(corners, dtype=torch.int32)
        if tuple(host.shape) != (N, 2):
            raise ValueError("corners must be [N, 2]")
        if not host.is_cuda:  # host data (what ImagePipeline.draw and callers pass): the window must lie inside the image,
            # the kernel does not clamp (a device tensor is the caller's responsibility: checking it would cost a sync)
            if int(host[:, 0].min()) < 0 or int(host[:, 0].max()) > H - h or int(host[:, 1].min()) < 0 \
                    or int(host[:, 1].max()) > W - w:
                raise ValueError("corners: crop window (%d, %d) leaves the %d x %d image" % (h, w, H, W))
        corners = host.to(dev).contiguous()
        cptr = corners.data_ptr()
    if flip is not None:
        flip = torch.as_tensor(flip).to(torch.uint8).to(dev).contiguous()
        fptr = flip.data_ptr()
    if (mean is None) != (std is None):
        raise ValueError("mean and std go together")
    if mean is not None:
        mean = torch.as_tensor(np.asarray(mean), dtype=torch.float32).to(dev)   # Normalize: as_tensor(mean, dtype=float32)
        std = torch.as_tensor(np.asarray(std), dtype=torch.float32).to(dev)
        if mean.numel() != C or std.numel() != C:
            raise ValueError("mean / std need one entry per channel")
        mptr, sptr = mean.data_ptr(), std.data_ptr()
    if channels_last:
        out = torch.empty((N, h, w, C), dtype=torch.float32, device=dev)
    else:
        out = torch.empty((N, C, h, w), dtype=torch.float32, device=dev)
    check(lib.migan_u8_to_f32(images.data_ptr(), out.data_ptr(), cptr, fptr, mptr, sptr, N, H, W, C, h, w, 0 if channels_last else 1,
                              _stream()), "u8_to_f32")
    return out.permute(0, 3, 1, 2) if channels_last else out


class ImagePipeline:
    """transforms.Compose([Resize(resize, filter), RandomCrop(crop), RandomHorizontalFlip(hflip_p), ToTensor(), Normalize(mean, std)])
    (cyclegan.py:111-117) for a batch of decoded images; every stage optional.  Random draws follow torchvision, per image in
    Compose order: RandomCrop.get_params draws torch.randint(0, H - h + 1, (1,)) then torch.randint(0, W - w + 1, (1,)) (none when
    the sizes already match), RandomHorizontalFlip draws torch.rand(1) < p - so a seeded run consumes the global torch RNG
    exactly like the reference's transform applied to the same images in the same order."""

    def __init__(self, resize=None, filt="bicubic", crop=None, hflip_p=0.0, mean=None, std=None, channels_last=True):
        self.resize, self.filt, self.crop, self.hflip_p = resize, filt, crop, float(hflip_p)
        self.mean, self.std, self.channels_last = mean, std, channels_last

    def draw(self, n, height, width):
        """(corners [n,2] or None, flips [n] or None) for n images of the resized size, in torchvision's draw order."""
        corners = np.zeros((n, 2), dtype=np.int32) if self.crop is not None else None
        flips = np.zeros(n, dtype=np.uint8) if self.hflip_p > 0.0 else None
        for i in range(n):
            if self.crop is not None:
                th, tw = self.crop
                if height < th or width < tw:
                    raise ValueError("Required crop size (%d, %d) is larger than input image size (%d, %d)" % (th, tw, height, width))
                if not (width == tw and height == th):
                    corners[i, 0] = torch.randint(0, height - th + 1, size=(1,)).item()
                    corners[i, 1] = torch.randint(0, width - tw + 1, size=(1,)).item()
            if flips is not None:
                flips[i] = 1 if torch.rand(1) < self.hflip_p else 0
        return corners, flips

    def __call__(self, images, corners=None, flips=None):
        x = images if self.resize is None else resize_u8(images, self.resize, self.filt)
        if corners is None and flips is None:
            corners, flips = self.draw(x.shape[0], x.shape[1], x.shape[2])
        return to_float(x, self.crop, corners, flips, self.mean, self.std, self.channels_last)
