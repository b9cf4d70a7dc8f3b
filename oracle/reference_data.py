"""ORACLE (test infrastructure): the reference's per-image input transforms on the CPU.

The reference builds them from torchvision.transforms (requirements.txt:2, unpinned; NOT installed in this image) applied to
PIL images (Pillow IS installed: 12.2.0 here).  The torchvision classes are restated below from their published algorithm
(torchvision/transforms/transforms.py and functional.py, 0.15-0.20: Resize on a PIL image = img.resize((w, h), interpolation) with
the smaller-edge rule for an int size; RandomCrop.get_params; RandomHorizontalFlip; ToTensor = uint8 HWC -> float CHW / 255;
Normalize = (x - mean) / std with fp32 mean/std tensors), while the arithmetic that matters - the bicubic / bilinear resample - is
executed by Pillow itself, i.e. by the reference's own dependency.  Call sites followed:
    cyclegan.py:111-117               Resize(int(h*1.12), BICUBIC), RandomCrop((h,w)), RandomHorizontalFlip(), ToTensor(), Normalize(.5,.5)
    srgan/datasets.py:16-33           Resize((h//4, h//4), BICUBIC) / Resize((h, h), BICUBIC), ToTensor(), Normalize(mean, std)
    dcgan.py:120-131                  Resize(img_size) (bilinear default), ToTensor(), Normalize([0.5],[0.5])
    pix2pix/datasets.py:24-41         Resize((h,w), BICUBIC), ToTensor(), Normalize; np.random flip of both halves
Only tests/ may import this module."""
import numpy as np
import torch
from PIL import Image

INTERP = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR}


def resize(img, size, interpolation="bilinear"):
    """torchvision.transforms.functional.resize for a PIL image."""
    w, h = img.size
    if isinstance(size, int):
        short, long_ = (w, h) if w <= h else (h, w)
        if short == size:
            return img
        new_short, new_long = size, int(size * long_ / short)
        ow, oh = (new_short, new_long) if w <= h else (new_long, new_short)
    else:
        oh, ow = size
    return img.resize((ow, oh), INTERP[interpolation])


def random_crop(img, size):
    """torchvision.transforms.RandomCrop (no padding): get_params draws i then j with torch.randint."""
    w, h = img.size
    th, tw = size
    if h < th or w < tw:
        raise ValueError("Required crop size %s is larger than input image size %s" % ((th, tw), (h, w)))
    if w == tw and h == th:
        return img
    i = torch.randint(0, h - th + 1, size=(1,)).item()
    j = torch.randint(0, w - tw + 1, size=(1,)).item()
    return img.crop((j, i, j + tw, i + th))


def random_hflip(img, p=0.5):
    if torch.rand(1) < p:
        return img.transpose(Image.FLIP_LEFT_RIGHT)
    return img


def to_tensor(img):
    """ToTensor for 8-bit PIL images: HWC uint8 -> CHW float32 in [0, 1]."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).contiguous()
    return t.to(dtype=torch.float32).div(255)


def normalize(t, mean, std):
    mean = torch.as_tensor(np.asarray(mean), dtype=t.dtype)
    std = torch.as_tensor(np.asarray(std), dtype=t.dtype)
    return t.clone().sub_(mean.view(-1, 1, 1)).div_(std.view(-1, 1, 1))


def cyclegan_transform(img, img_height, img_width):
    """cyclegan.py:111-117 applied to one PIL image."""
    x = resize(img, int(img_height * 1.12), "bicubic")
    x = random_crop(x, (img_height, img_width))
    x = random_hflip(x)
    return normalize(to_tensor(x), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))


SRGAN_MEAN = np.array([0.485, 0.456, 0.406])   # srgan/datasets.py:12-13
SRGAN_STD = np.array([0.229, 0.224, 0.225])


def srgan_transform(img, hr_height):
    """srgan/datasets.py:16-33,38-42 -> (lr, hr)."""
    lr = normalize(to_tensor(resize(img, (hr_height // 4, hr_height // 4), "bicubic")), SRGAN_MEAN, SRGAN_STD)
    hr = normalize(to_tensor(resize(img, (hr_height, hr_height), "bicubic")), SRGAN_MEAN, SRGAN_STD)
    return lr, hr


def dcgan_transform(img, img_size):
    """dcgan.py:125-127 on one MNIST digit (mode L)."""
    return normalize(to_tensor(resize(img, img_size, "bilinear")), [0.5], [0.5])
