"""Input pipeline on the device against Pillow + the oracle's restatement of torchvision's transforms:
bit-exact uint8 resamples, bit-exact fp32 batches for the reference's transform chains (cyclegan.py:111-117,
srgan/datasets.py:16-33, dcgan.py:120-131), ragged sizes, and size-independent properties at dataset scale."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PIL_FILTER = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR}


def _imgs(n, h, w, c, seed):
    rng = np.random.RandomState(seed)
    a = rng.randint(0, 256, (n, h, w, c)).astype(np.uint8)
    a[:, 0, 0], a[:, -1, -1] = 255, 0
    return a


def _pil(a):
    return Image.fromarray(a[:, :, 0], "L") if a.shape[2] == 1 else Image.fromarray(a, "RGB")


@pytest.mark.parametrize("case", [(2, 256, 256, 3, (286, 286), "bicubic"), (3, 218, 178, 3, 286, "bicubic"), (2, 178, 218, 3, (64, 64), "bicubic"),
                                  (4, 28, 28, 1, 64, "bilinear"), (1, 500, 375, 3, (96, 96), "bicubic"), (2, 37, 53, 3, (128, 91), "bilinear"),
                                  (1, 64, 64, 3, (64, 17), "bicubic"), (1, 5, 7, 3, (1, 1), "bicubic"), (2, 1, 1, 1, (9, 4), "bilinear"),
                                  (1, 96, 96, 3, (96, 96), "bicubic")])
def test_resize_bit_exact_vs_pillow(case):
    import pytorch_gan_amd.data as D

    n, h, w, c, size, filt = case
    a = _imgs(n, h, w, c, seed=h + w)
    got = D.resize_u8(torch.from_numpy(a).to(DEV), size, filt).cpu().numpy()
    oh, ow = D.resize_output_size(size, h, w)
    assert got.shape == (n, oh, ow, c)
    for i in range(n):
        ref = np.asarray(_pil(a[i]).resize((ow, oh), PIL_FILTER[filt]))
        ref = ref[:, :, None] if ref.ndim == 2 else ref
        assert np.array_equal(got[i], ref), "image %d differs from Pillow" % i


def test_cyclegan_transform_chain_bit_exact():
    """cyclegan.py:111-117 on a batch: same torch seed -> same crops and flips -> identical fp32 tensors."""
    import pytorch_gan_amd.data as D
    from oracle import reference_data as R

    a = _imgs(5, 218, 178, 3, seed=3)
    pipe = D.ImagePipeline(resize=int(64 * 1.12), filt="bicubic", crop=(64, 64), hflip_p=0.5, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5))
    torch.manual_seed(11)
    got = pipe(torch.from_numpy(a).to(DEV))
    torch.manual_seed(11)
    want = torch.stack([R.cyclegan_transform(_pil(a[i]), 64, 64) for i in range(5)])
    assert got.shape == want.shape == (5, 3, 64, 64)
    assert got.is_contiguous(memory_format=torch.channels_last)     # NHWC storage: feeds the first conv without a re-layout
    assert torch.equal(got.cpu(), want)
    nchw = D.ImagePipeline(resize=int(64 * 1.12), crop=(64, 64), hflip_p=0.5, mean=(0.5,) * 3, std=(0.5,) * 3, channels_last=False)
    torch.manual_seed(11)
    got2 = nchw(torch.from_numpy(a).to(DEV))
    assert got2.is_contiguous() and torch.equal(got2.cpu(), want)


def test_srgan_and_dcgan_transform_chains_bit_exact():
    import pytorch_gan_amd.data as D
    from oracle import reference_data as R

    a = _imgs(3, 218, 178, 3, seed=4)
    dev = torch.from_numpy(a).to(DEV)
    lr = D.to_float(D.resize_u8(dev, (24, 24), "bicubic"), mean=R.SRGAN_MEAN, std=R.SRGAN_STD)
    hr = D.to_float(D.resize_u8(dev, (96, 96), "bicubic"), mean=R.SRGAN_MEAN, std=R.SRGAN_STD)
    for i in range(3):
        wl, wh = R.srgan_transform(_pil(a[i]), 96)
        assert torch.equal(lr[i].cpu(), wl) and torch.equal(hr[i].cpu(), wh)
    m = _imgs(6, 28, 28, 1, seed=5)
    got = D.ImagePipeline(resize=64, filt="bilinear", mean=[0.5], std=[0.5])(torch.from_numpy(m).to(DEV))
    want = torch.stack([R.dcgan_transform(_pil(m[i]), 64) for i in range(6)])
    assert torch.equal(got.cpu(), want)
    # ToTensor alone (no Normalize) and explicit corner / flip arguments (pix2pix flips both halves with one numpy draw)
    t = D.to_float(dev, crop=(100, 90), corners=[[3, 5], [0, 0], [118, 88]], flip=[1, 0, 1], channels_last=False).cpu()
    for i, (cy, cx, f) in enumerate([(3, 5, 1), (0, 0, 0), (118, 88, 1)]):
        w_ = a[i, cy:cy + 100, cx:cx + 90]
        w_ = w_[:, ::-1] if f else w_
        assert torch.equal(t[i], torch.from_numpy(np.ascontiguousarray(w_)).permute(2, 0, 1).float().div(255))


def test_rejects_what_the_reference_rejects():
    import pytorch_gan_amd.data as D

    dev = torch.zeros(1, 8, 8, 3, dtype=torch.uint8, device=DEV)
    with pytest.raises(ValueError):
        D.to_float(dev, crop=(9, 8))                      # RandomCrop: crop larger than the image
    with pytest.raises(TypeError):
        D.resize_u8(torch.zeros(1, 8, 8, 3, dtype=torch.uint8), (4, 4))   # host tensor: no CPU fallback
    with pytest.raises(TypeError):
        D.resize_u8(dev.float(), (4, 4))
    with pytest.raises(ValueError):
        D.resize_u8(dev, (4, 4), "lanczos")


def test_dataset_scale_properties():
    """CelebA-sized batch (64 x 218 x 178 x 3 -> 286-short-edge bicubic): properties that hold at any size - a constant image
    stays constant (coefficient rows sum to one), a same-size resize is the identity, every image of the batch is processed
    like that image alone, and image 0 equals Pillow."""
    import pytorch_gan_amd.data as D

    a = _imgs(64, 218, 178, 3, seed=9)
    a[1] = 77
    dev = torch.from_numpy(a).to(DEV)
    out = D.resize_u8(dev, 286, "bicubic")
    assert out.shape == (64, 350, 286, 3)
    assert bool((out[1] == 77).all())
    assert D.resize_u8(dev, (218, 178), "bicubic") is dev
    assert torch.equal(D.resize_u8(dev[5:6].contiguous(), 286, "bicubic")[0], out[5])
    assert np.array_equal(out[0].cpu().numpy(), np.asarray(_pil(a[0]).resize((286, 350), Image.BICUBIC)))
