"""CPU suite: the C-ABI library loads, exports every symbol include/migan.h declares (no compute calls without a
GPU), the host-side mirror of the reference interface behaves (module tree / state_dict / error paths), and the
product path refuses to run without the HIP device."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "migan.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(migan_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    import pytorch_gan_amd  # noqa: F401
    from pytorch_gan_amd import _lib

    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(_lib.lib, s), "libmigan.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == syms, "ctypes table and header drifted: %s" % (set(_lib.EXPORTS) ^ set(syms))
    assert _lib.version().startswith("migan")
    # pure host-side queries are callable without a GPU
    assert _lib.lib.migan_adam_chunk() > 0
    assert _lib.lib.migan_conv2d_wgrad_workspace(128, 32, 32, 128, 3, 3, 128) > 0
    assert _lib.lib.migan_norm_workspace(1, 131072, 128) > 0
    assert _lib.lib.migan_igemm_tile_code(131072, 128, 128, 1) == 1128128


def test_library_is_gfx950_code_object():
    import subprocess

    so = os.path.join(ROOT, "pytorch-gan_amd", "csrc", "libmigan.so")
    out = subprocess.run(["strings", "-n", "6", so], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_no_cpu_fallback():
    import pytorch_gan_amd as pg

    with pytest.raises(RuntimeError):
        pg.functional.conv2d(torch.zeros(1, 4, 4, 4), torch.zeros(4, 4, 3, 3))
    with pytest.raises(RuntimeError):
        pg.optim.Adam([torch.nn.Parameter(torch.zeros(3))])
    m = pg.nn.Conv2d(4, 4, 3)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pytorch-gan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_swap_and_dropin_namespace():
    import pytorch_gan_amd as pg
    import pytorch_gan_amd.nn as nn
    from oracle import reference_models as M

    # constructor call sites of the reference keep working verbatim
    layers = [nn.Conv2d(128, 128, 3, stride=1, padding=1), nn.Conv2d(16, 32, 3, 2, 1), nn.Conv2d(3, 64, 4, 2, 1, bias=False),
              nn.Conv2d(512, 1, 4, padding=1), nn.ConvTranspose2d(64, 32, 4, 2, 1, bias=False), nn.BatchNorm2d(128),
              nn.BatchNorm2d(128, 0.8), nn.BatchNorm1d(256, 0.8), nn.InstanceNorm2d(64), nn.LeakyReLU(0.2, inplace=True),
              nn.ReLU(inplace=True), nn.PReLU(), nn.Upsample(scale_factor=2), nn.ReflectionPad2d(3),
              nn.ZeroPad2d((1, 0, 1, 0)), nn.PixelShuffle(upscale_factor=2), nn.Dropout2d(0.25), nn.Dropout(0.5),
              nn.Linear(100, 128), nn.Tanh(), nn.Sigmoid(), nn.BCELoss(), nn.MSELoss(), nn.L1Loss()]
    assert layers[6].eps == 0.8 and layers[5].eps == 1e-5
    for fam in (M.DcganGenerator(32), M.DcganDiscriminator(32), M.CycleGenerator((3, 32, 32), 2), M.Pix2pixGenerator(),
                M.SrganGenerator(n_residual_blocks=2), M.SrganDiscriminator((3, 32, 32)), M.SrganFeatureExtractor(),
                M.MlpGenerator(), M.MlpCritic()):
        keys = list(fam.state_dict().keys())
        shapes = [tuple(v.shape) for v in fam.state_dict().values()]
        pg.swap(fam)
        assert list(fam.state_dict().keys()) == keys
        assert [tuple(v.shape) for v in fam.state_dict().values()] == shapes
        for m in fam.modules():
            if len(list(m.children())) == 0:
                assert type(m) in nn._OURS, type(m)
    # weights_init_normal dispatches on class-name substrings (dcgan.py:36-42)
    G = pg.swap(M.DcganGenerator(32))
    assert any("Conv" in type(m).__name__ for m in G.modules())
    assert any("BatchNorm2d" in type(m).__name__ for m in G.modules())
    with pytest.raises(NotImplementedError):
        pg.swap(torch.nn.Sequential(torch.nn.GRU(4, 4)))
    with pytest.raises(ValueError):
        nn.Conv2d(4, 4, 3, groups=2)._check()
    with pytest.raises(ValueError):
        nn.BCELoss(reduction="sum")


def test_product_models_match_reference_trees():
    from oracle import reference_models as OM
    from pytorch_gan_amd import models as PM

    pairs = [(PM.DcganGenerator(32), OM.DcganGenerator(32)), (PM.DcganDiscriminator(32), OM.DcganDiscriminator(32)),
             (PM.MlpGenerator(), OM.MlpGenerator()), (PM.MlpCritic(), OM.MlpCritic()),
             (PM.CycleGenerator((3, 32, 32), 2), OM.CycleGenerator((3, 32, 32), 2)),
             (PM.CycleDiscriminator((3, 32, 32)), OM.CycleDiscriminator((3, 32, 32))),
             (PM.Pix2pixDiscriminator(), OM.Pix2pixDiscriminator()), (PM.SrganGenerator(), OM.SrganGenerator()),
             (PM.SrganDiscriminator((3, 32, 32)), OM.SrganDiscriminator((3, 32, 32))),
             (PM.SrganFeatureExtractor(), OM.SrganFeatureExtractor()),
             (PM.EsrganGenerator(3, 64, 2), OM.EsrganGenerator(3, 64, 2)),
             (PM.EsrganDiscriminator((3, 32, 32)), OM.EsrganDiscriminator((3, 32, 32))),
             (PM.EsrganFeatureExtractor(), OM.EsrganFeatureExtractor()),
             (PM.AcganGenerator(32), OM.AcganGenerator(32)), (PM.AcganDiscriminator(32), OM.AcganDiscriminator(32))]
    for a, b in pairs:
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        assert all(sa[k].shape == sb[k].shape for k in sa)
        b.load_state_dict(sa)  # checkpoints are interchangeable


def test_replay_buffer_and_lambda_lr_match_oracle():
    import random

    from oracle import reference_models as OM
    from pytorch_gan_amd import steps

    random.seed(3)
    a, b = steps.ReplayBuffer(max_size=3), OM.ReplayBuffer(max_size=3)
    for i in range(8):
        batch = torch.full((2, 1, 2, 2), float(i)) + torch.tensor([0.0, 0.5]).view(2, 1, 1, 1)
        st = random.getstate()
        x = a.push_and_pop(batch)
        random.setstate(st)
        y = b.push_and_pop(batch)
        assert torch.equal(x, y)
    lam = OM.lambda_lr(200, 0, 100)
    mine = steps.LambdaLR(200, 0, 100)
    assert all(lam(e) == mine.step(e) for e in range(0, 200, 7))
    with pytest.raises(AssertionError):
        steps.LambdaLR(100, 0, 100)


def test_thin_toeplitz_applicability():
    """Pure host functions of the C ABI (no launch): where the width-Toeplitz expansion applies, and its buffer sizes."""
    from pytorch_gan_amd._lib import lib

    assert lib.migan_thin_toeplitz_ok(3, 3, 3, 64, 1, 0) == 0    # 3x3: 9 of 32 columns - stays on the direct kernel
    assert lib.migan_thin_toeplitz_ok(3, 7, 7, 64, 2, 0) == 0    # stride 2
    assert lib.migan_thin_toeplitz_ok(3, 7, 7, 3, 1, 0) == 0     # 3 source channels
    assert lib.migan_thin_toeplitz_ok(3, 7, 7, 64, 1, 2) == 0    # upsampling gather
    assert lib.migan_thin_toeplitz_ok(8, 7, 7, 64, 1, 0) == 0
    # Co' = 32 since round 5 (the input-gradient GEMM's source channels: whole 32-channel K-tiles for the LDS-DMA kernels)
    assert lib.migan_thin_toeplitz_cols(3, 9) == 32 and lib.migan_thin_toeplitz_cols(3, 7) == 32
    assert lib.migan_thin_toeplitz_workspace(16, 384, 384, 3, 9) == 16 * 384 * 384 * 32 * 4
    # image-input layers (csrc/rgb_conv.hip): 3 source channels, stride 1, square 3 / 7 kernels, 32 / 64 output channels
    assert lib.migan_rgb_conv_ok(3, 64, 3, 3, 1, 0, 16 * 384 * 384) == 1 and lib.migan_rgb_conv_ok(3, 64, 7, 7, 1, 1, 8 * 256 * 256) == 1
    assert lib.migan_rgb_conv_ok(3, 64, 3, 3, 2, 0, 1 << 20) == 0 and lib.migan_rgb_conv_ok(1, 16, 3, 3, 1, 0, 1 << 20) == 0
    assert lib.migan_rgb_conv_ok(3, 64, 3, 3, 1, 0, 1000) == 0 and lib.migan_rgb_conv_ok(3, 64, 4, 4, 1, 0, 1 << 20) == 0
    assert lib.migan_rgb_conv_wgrad_ok(3, 64, 9, 9, 1, 0, 1 << 20) == 0 and lib.migan_rgb_conv_wgrad_ok(3, 64, 7, 7, 1, 1, 1 << 20) == 1
    assert lib.migan_rgb_conv_wgrad_workspace(64, 3, 3) == 1024 * 64 * 32 * 4


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype of include/migan.h against the argtypes / restype table the host mirror calls through (pytorch_gan_amd/_lib.py):
    same parameter count, and per parameter the same class - pointer, int, float, size_t, (unsigned) long long.  A drift here is silent
    on the GPU: ctypes passes what the table says, the kernel launcher reads what the prototype says."""
    import ctypes

    from pytorch_gan_amd import _lib

    src = open(os.path.join(ROOT, "include", "migan.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = re.findall(r"([A-Za-z_][\w \*]*?)\b(migan_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src)
    assert len(protos) >= 100, len(protos)

    def cls(decl):
        d = decl.strip()
        if "*" in d:
            return "ptr"
        d = re.sub(r"\b\w+$", "", d).strip() if len(d.split()) > 1 else d   # drop the parameter name
        d = d.replace("const", "").strip()
        return {"int": "int", "float": "float", "size_t": "size_t", "long long": "longlong", "unsigned long long": "ulonglong",
                "long": "long", "unsigned": "uint", "unsigned int": "uint", "void": "void"}[d]

    def ccls(t):
        if t is None:
            return "void"
        if t in (ctypes.c_void_p, ctypes.c_char_p) or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            return "ptr"
        return {ctypes.c_int: "int", ctypes.c_float: "float", ctypes.c_size_t: "size_t", ctypes.c_longlong: "longlong",
                ctypes.c_ulonglong: "ulonglong", ctypes.c_long: "long", ctypes.c_uint: "uint"}[t]

    # on this ABI size_t and unsigned long long are the same register class, as are long and long long: equal for the call
    same = {"size_t": "u64", "ulonglong": "u64", "long": "i64", "longlong": "i64"}
    checked = 0
    for ret, name, params in protos:
        fn = getattr(_lib.lib, name)
        want = [] if params.strip() in ("", "void") else [cls(p) for p in params.split(",")]
        got = [ccls(t) for t in (fn.argtypes or [])]
        assert len(want) == len(got), (name, len(want), len(got))
        for i, (a, b) in enumerate(zip(want, got)):
            assert same.get(a, a) == same.get(b, b), (name, i, a, b)
        r_want, r_got = cls(ret + " x") if "*" not in ret else "ptr", ccls(fn.restype)
        assert same.get(r_want, r_want) == same.get(r_got, r_got), (name, "return", r_want, r_got)
        checked += 1
    assert checked == len(_lib.EXPORTS), (checked, len(_lib.EXPORTS))


def test_integration_md_ctypes_stubs_match_the_signature_table():
    """The reference-side binding stubs INTEGRATION.md shows (`lib.<entry>.argtypes = ...`, `.restype = ...`) are the signatures of
    the product's own ctypes table - a stub that drifts from the ABI (a removed sync word, an added phase argument) fails here."""
    import ctypes

    from pytorch_gan_amd import _lib

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    ns = {"ctypes": ctypes, "P": ctypes.c_void_p, "I": ctypes.c_int, "F": ctypes.c_float}
    found = 0
    for name, expr in re.findall(r"^lib\.(migan_\w+)\.argtypes = ([^#\n]+)", text, flags=re.M):
        want = _lib._SIGS[name][1]
        got = eval(expr, ns)   # noqa: S307 - our own document
        assert [t for t in got] == [t for t in want], (name, got, want)
        found += 1
    for name, expr in re.findall(r"^lib\.(migan_\w+)\.restype = ([^#\n]+)", text, flags=re.M):
        assert eval(expr, ns) is _lib._SIGS[name][0], name   # noqa: S307
        found += 1
    assert found >= 4, found
