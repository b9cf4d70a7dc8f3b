"""Input pipeline, host side (no GPU): the resample coefficient tables pytorch_gan_amd.data computes for the device kernels
reproduce Pillow's resize bit for bit when evaluated with Pillow's integer arithmetic in numpy; torchvision's Resize size rule;
the random-draw order of the crop / flip stage against the oracle's restatement of torchvision (oracle/reference_data.py)."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _np_resample(D, img, oh, ow, filt):
    """Pillow's two passes (horizontal first, uint8 intermediate) in numpy int64, driven by the product's coefficient tables."""
    a = img.astype(np.int64)
    h, w, c = img.shape

    def clip8(v):
        return np.clip(v >> D.PRECISION_BITS, 0, 255).astype(np.uint8)

    half = 1 << (D.PRECISION_BITS - 1)
    if ow != w:
        kk, b, _ = D.pil_resample_coeffs(w, ow, filt)
        t = np.zeros((h, ow, c), np.uint8)
        for xx in range(ow):
            x0, n = b[xx]
            t[:, xx] = clip8(half + (a[:, x0:x0 + n] * kk[xx, :n].astype(np.int64)[None, :, None]).sum(1))
        a, w = t.astype(np.int64), ow
    if oh != h:
        kk, b, _ = D.pil_resample_coeffs(h, oh, filt)
        t = np.zeros((oh, w, c), np.uint8)
        for yy in range(oh):
            y0, n = b[yy]
            t[yy] = clip8(half + (a[y0:y0 + n] * kk[yy, :n].astype(np.int64)[:, None, None]).sum(0))
        return t
    return a.astype(np.uint8)


@pytest.mark.parametrize("case", [(256, 256, 286, 286, "bicubic"), (178, 218, 64, 64, "bicubic"), (218, 178, 350, 286, "bicubic"),
                                  (28, 28, 64, 64, "bilinear"), (500, 375, 96, 96, "bicubic"), (37, 53, 128, 91, "bilinear"),
                                  (64, 64, 64, 17, "bicubic"), (5, 7, 1, 1, "bicubic"), (1, 1, 9, 4, "bilinear")])
def test_coefficient_tables_reproduce_pillow(case):
    import pytorch_gan_amd.data as D

    h, w, oh, ow, filt = case
    rng = np.random.RandomState(h * 1000 + w)
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    img[0, 0], img[-1, -1] = 255, 0   # saturating corners (negative bicubic lobes -> clip8 on both sides)
    ref = np.array(Image.fromarray(img, "RGB").resize((ow, oh), {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR}[filt]))
    assert np.array_equal(_np_resample(D, img, oh, ow, filt), ref)
    kk, b, ks = D.pil_resample_coeffs(w, ow, filt)
    assert kk.shape == (ow, ks) and b.shape == (ow, 2) and kk.dtype == np.int32
    assert (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= w).all() and (b[:, 1] <= ks).all()
    assert (np.abs(kk.astype(np.int64).sum(1) - (1 << D.PRECISION_BITS)) <= ks).all()   # rows sum to 1.0 up to rounding


def test_resize_size_rule_and_draw_order():
    import pytorch_gan_amd.data as D
    from oracle import reference_data as R

    for (h, w) in [(218, 178), (178, 218), (256, 256), (300, 301)]:
        img = Image.fromarray(np.zeros((h, w, 3), np.uint8))
        for size in (286, 64, (96, 96), (24, 50)):
            out = R.resize(img, size, "bicubic")
            assert D.resize_output_size(size, h, w) == (out.size[1], out.size[0])
    # crop / flip draws: the product draws per image in Compose order like the oracle (= torchvision) does
    pipe = D.ImagePipeline(resize=286, crop=(256, 256), hflip_p=0.5)
    torch.manual_seed(42)
    corners, flips = pipe.draw(6, 286, 350)
    torch.manual_seed(42)
    rng = np.random.RandomState(0)
    for i in range(6):
        a = rng.randint(0, 256, (286, 350, 3)).astype(np.uint8)
        a[:, :, 0] = np.arange(350, dtype=np.uint8)[None, :]      # column ramp: reveals offset and flip
        a[:, :, 1] = (np.arange(286) % 256).astype(np.uint8)[:, None]
        x = R.random_hflip(R.random_crop(Image.fromarray(a), (256, 256)))
        px = np.asarray(x)
        top, left = int(corners[i, 0]), int(corners[i, 1])
        assert int(px[0, 0, 1]) == top % 256
        if flips[i]:
            assert int(px[0, 0, 0]) == (left + 255) % 256 and int(px[0, 255, 0]) == left % 256
        else:
            assert int(px[0, 0, 0]) == left % 256 and int(px[0, 255, 0]) == (left + 255) % 256
    assert 0 < int(flips.sum()) < 6   # the seed exercises both branches
    # sizes already equal: no draw is consumed (RandomCrop.get_params)
    torch.manual_seed(7)
    s0 = torch.get_rng_state()
    c, f = D.ImagePipeline(crop=(10, 12)).draw(3, 10, 12)
    assert (c == 0).all() and f is None and torch.equal(torch.get_rng_state(), s0)
    with pytest.raises(ValueError):
        D.ImagePipeline(crop=(300, 300)).draw(1, 286, 286)
