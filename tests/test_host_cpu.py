"""Host-side logic that needs no GPU: the packed-weight cache stamps, the direct-gradient slot rules and the
bookkeeping helpers of bench.py (kernel symbol names, committed PMC traffic table)."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_weight_cache_stamps():
    """Packed weights are re-used only inside ONE weight_cache_scope() (= one training-step body) and only while the
    optimiser epoch, the tensor version and the storage are unchanged; nothing cached survives the scope, so writes no
    stamp can see (`m.weight.data.normal_()` of weights_init_normal, dcgan.py:36-42; a graph-replayed Adam) are safe."""
    import pytorch_gan_amd.functional as F
    from pytorch_gan_amd import optim

    calls = []

    def make():
        calls.append(1)
        return torch.full((1,), float(len(calls)))

    w = torch.nn.Parameter(torch.zeros(4))
    w._migan_epoch = next(optim._EPOCH)
    assert F._packed(w, w, "k", make) is not F._packed(w, w, "k", make)       # outside a scope: never cached
    assert len(calls) == 2
    with F.weight_cache_scope():
        v = torch.zeros(4)
        assert F._packed(v, v, "k", make) is not F._packed(v, v, "k", make)   # not a Parameter (an activation) -> never cached
        frozen = torch.nn.Parameter(torch.zeros(4))                           # no optimiser (the VGG of srgan.py:60-62):
        assert F._packed(frozen, frozen, "k", make) is F._packed(frozen, frozen, "k", make)   # cached for this scope only
        n0 = len(calls)
        a = F._packed(w, w, "k", make)
        assert F._packed(w, w, "k", make) is a and len(calls) == n0 + 1       # same stamp -> hit
        with F.weight_cache_scope():                                          # nested scope == the outer one
            assert F._packed(w, w, "k", make) is a
        assert F._packed(w, w, "other", make) is not a                        # cache is per pack kind
        w._migan_epoch = next(optim._EPOCH)                                   # an optimiser step
        b = F._packed(w, w, "k", make)
        assert b is not a
        with torch.no_grad():
            w.add_(1.0)                                                       # version bump (copy_/load_state_dict)
        c = F._packed(w, w, "k", make)
        assert c is not b
        e1, e2 = next(optim._EPOCH), next(optim._EPOCH)
        assert e2 > e1 > w._migan_epoch                                       # stamps are never re-used
    with F.weight_cache_scope():
        w.data.fill_(3.0)                                                     # invisible to epoch AND version ...
        d = F._packed(w, w, "k", make)
        assert d is not c                                                     # ... but a new step never sees old packs
        assert F._packed(w, w, "k", make) is d
        F.set_weight_cache(False)
        try:
            assert F._packed(w, w, "k", make) is not d                        # globally off: always repack
        finally:
            F.set_weight_cache(True)


def test_grad_slot_rules():
    import pytorch_gan_amd.functional as F

    p = torch.nn.Parameter(torch.zeros(3, 4))
    with torch.no_grad():  # first-order backward runs with grad mode off
        assert F._grad_slot(p) is None                      # no .grad yet -> autograd accumulates
        p.grad = torch.zeros(3, 4)
        assert F._grad_slot(p) is p.grad                    # contiguous fp32 same-shape buffer -> direct accumulation
        p.grad = torch.zeros(4, 3).t()
        assert F._grad_slot(p) is None                      # non-contiguous
        p.grad = torch.zeros(3, 4)
        q = (p * 2.0)
        assert F._grad_slot(q) is None                      # non-leaf
        F.set_direct_grad(False)
        try:
            assert F._grad_slot(p) is None
        finally:
            F.set_direct_grad(True)
    assert F._grad_slot(p) is None                          # grad mode on (create_graph backward): stay differentiable


def test_bench_flop_accounting_and_pmc_table():
    import bench

    # 3 G forwards-equivalents + 8 D forwards-equivalents: 2.811 GFLOP per image, of which 2.718 are in
    # the two Upsample+Conv3x3 layers; the phase-collapsed kernels execute 16/36 of those -> 1.301 GFLOP/img executed
    assert abs(bench.GFLOP_PER_IMG["dcgan"] - 2.8107) < 1e-3
    assert abs(bench.UPCONV_GFLOP_PER_IMG["dcgan"] - 2.7181) < 1e-3
    assert abs(bench.executed_gflop_per_image("dcgan") - 1.3007) < 1e-3
    # cyclegan: u128 + u64 = 2 x 77.31 GFLOP per G forward at bs 8, 18 G-forward equivalents per step
    assert abs(bench.UPCONV_GFLOP_PER_IMG["cyclegan"] - 347.9) < 0.1
    assert abs(bench.executed_gflop_per_image("cyclegan") - (2097.99 - 347.9 * 20 / 36)) < 0.1
    assert bench.executed_gflop_per_image("srgan") == bench.GFLOP_PER_IMG["srgan"]
    # esrgan.py defaults (hr 256, 23 RRDB): 3 G + 9 D + 3 VGG19[:35] forward-equivalents; G forward = 324.0, D 12.31, VGG 50.96 GFLOP
    assert abs(bench.GFLOP_PER_IMG["esrgan"] - 1235.58) < 0.05
    assert abs(bench.esrgan_flops_per_image(256, 23) / 1e9 - (3 * 324.00 + 9 * 12.308 + 3 * 50.96)) < 1.0
    s = bench.summarise("dcgan", 128, 1, 20, [0.08, 0.07, 0.09])
    assert s["ms_per_step"] == 4.0 and s["ms_per_step_min"] == 3.5 and s["blocks"] == 3
    assert abs(s["images_per_s"] - 32000.0) < 1e-6
    assert abs(s["step_executed_frac"] - 32000 * 1.3007e9 / 157.3e12) < 2e-4
    # the current round's table (tools/pmc_step.py over the rocprofv3 --pmc passes of `bench.py --pmc-log`: counters of the kernels IN the
    # training step); the committed table of round 2 (tools/pmc_kernels.py, stand-alone microbench passes, register-staged kernels)
    # is kept as history - the arithmetic of both is checked
    assert bench.PROFILE_ROUND == "r06"
    tab4, src4 = bench.pmc_table()   # this round's table, or the previous round's until this round's passes have been taken
    if src4 is not None:
        assert src4 in ("profiles/r06_pmc_kernels.json", "profiles/r05_pmc_kernels.json") and any(k.startswith("upconv_wgrad[") for k in tab4)
    bench.PROFILE_ROUND = "r02"
    try:
        tab, src = bench.pmc_table()
    finally:
        bench.PROFILE_ROUND = "r06"
    assert src == "profiles/r02_pmc_kernels.json" and "upconv_fwd[128x 128->64 @64]" in tab and "_calibration" in tab
    for name, ent in list(tab.items()) + [kv for kv in tab4.items() if "hbm_bytes_per_launch" in kv[1]]:
        if name.startswith("_"):
            continue
        assert ent["hbm_bytes_per_launch"] > 0 and ent["symbol"]
        rebuilt = (2 * ent["fetch_size_kb_reported"] + ent["write_size_kb_reported"]) * 1024  # FETCH x2 (gfx950), WRITE x1
        assert abs(ent["hbm_bytes_per_launch"] - rebuilt) <= 1e-3 * rebuilt + 2048
        if "mfma_busy_frac" in ent:   # SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs / GRBM_GUI_ACTIVE over the 8 XCDs
            assert abs(ent["mfma_busy_frac"] - ent["mfma_busy_cycles"] / (128.0 * ent["grbm_gui_active"])) < 2e-3
            assert 0.0 <= ent["mfma_busy_frac"] <= 1.0


def test_sequential_fusion_plan(monkeypatch):
    """Which launcher families nn.Sequential routes the reference's layer patterns to (no kernels run: the functional
    entry points are replaced by recorders that only produce tensors of the right shape)."""
    import pytorch_gan_amd.functional as F
    import pytorch_gan_amd.nn as nn

    calls = []

    def out_hw(h, w, k, stride, pads, gather):
        hl, wl = (2 * h, 2 * w) if gather == F.GATHER_UP2 else (h, w)
        return (hl + pads[0] + pads[2] - k) // stride + 1, (wl + pads[1] + pads[3] - k) // stride + 1

    def conv2d(x, w, b=None, stride=1, pads=(0, 0, 0, 0), gather=0, act=0, slope=0.0, dropout_mask=None, stats=None, relu_in=False,
               handed=False):
        calls.append(("conv2d", tuple(pads), gather, act, dropout_mask is not None) + ((stats,) if stats else ())
                     + (("relu_in",) if relu_in else ()) + (("handed",) if handed else ()))
        ho, wo = out_hw(x.shape[2], x.shape[3], w.shape[2], stride, pads, gather)
        return torch.zeros(x.shape[0], w.shape[0], ho, wo)

    def upconv3x3(x, w, b=None, act=0, slope=0.0, stats=None):
        calls.append(("upconv3x3", act) + ((stats,) if stats else ()))
        return torch.zeros(x.shape[0], w.shape[0], 2 * x.shape[2], 2 * x.shape[3])

    def norm(x, gamma=None, beta=None, res=None, rm=None, rv=None, use_batch_stats=True, momentum=0.1, eps=1e-5,
             instance=False, act=0, slope=0.0, num_batches_tracked=None, prelu=None, shuffle=0, mask=None):
        calls.append(("norm", bool(instance), act, float(eps), num_batches_tracked is not None) + (("prelu",) if prelu is not None else ())
                     + (("shuffle", shuffle) if shuffle else ()))
        if shuffle:
            return torch.zeros(x.shape[0], x.shape[1] // (shuffle * shuffle), x.shape[2] * shuffle, x.shape[3] * shuffle)
        return torch.zeros_like(x)

    def activation(x, act, slope=0.0):
        calls.append(("activation", act))
        return torch.zeros_like(x)

    def gather2d(x, pads, mode):
        calls.append(("gather2d", tuple(pads), mode))
        h, w = out_hw(x.shape[2], x.shape[3], 1, 1, pads, mode)
        return torch.zeros(x.shape[0], x.shape[1], h, w)

    for name, fn in dict(conv2d=conv2d, upconv3x3=upconv3x3, norm=norm, activation=activation, gather2d=gather2d).items():
        monkeypatch.setattr(F, name, fn)
    monkeypatch.setattr(nn, "_next_mask", lambda shape, p, device: torch.ones(shape))

    # DCGAN discriminator block (dcgan.py:77-80): Conv -> LeakyReLU -> Dropout2d -> BatchNorm2d(out, 0.8)
    blk = nn.Sequential(nn.Conv2d(16, 32, 3, 2, 1), nn.LeakyReLU(0.2, inplace=True), nn.Dropout2d(0.25),
                        nn.BatchNorm2d(32, 0.8))
    y = blk(torch.zeros(2, 16, 8, 8))
    assert tuple(y.shape) == (2, 32, 4, 4)
    # ... and the BatchNorm statistics are requested from the conv epilogue
    assert calls == [("conv2d", (1, 1, 1, 1), F.GATHER_ZERO, F.ACT_LRELU, True, "batch"), ("norm", False, F.ACT_NONE, 0.8, True)]

    # DCGAN generator block (dcgan.py:54-57): Upsample -> Conv3x3 -> BatchNorm -> LeakyReLU  = collapsed up-conv + fused norm
    calls.clear()
    blk = nn.Sequential(nn.Upsample(scale_factor=2), nn.Conv2d(8, 8, 3, stride=1, padding=1), nn.BatchNorm2d(8, 0.8),
                        nn.LeakyReLU(0.2, inplace=True))
    y = blk(torch.zeros(2, 8, 4, 4))
    assert tuple(y.shape) == (2, 8, 8, 8)
    assert calls == [("upconv3x3", F.ACT_NONE, "batch"), ("norm", False, F.ACT_LRELU, 0.8, True)]

    # CycleGAN residual block body (cyclegan/models.py:27-35): ReflectionPad -> Conv -> InstanceNorm -> ReLU -> ...
    calls.clear()
    blk = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(8, 8, 3), nn.InstanceNorm2d(8), nn.ReLU(inplace=True),
                        nn.ReflectionPad2d(1), nn.Conv2d(8, 8, 3), nn.InstanceNorm2d(8))
    y = blk(torch.zeros(1, 8, 6, 6))
    assert tuple(y.shape) == (1, 8, 6, 6)
    assert calls == [("conv2d", (1, 1, 1, 1), F.GATHER_REFLECT, F.ACT_NONE, False, "instance"), ("norm", True, F.ACT_RELU, 1e-5, False),
                     ("conv2d", (1, 1, 1, 1), F.GATHER_REFLECT, F.ACT_NONE, False, "instance"), ("norm", True, F.ACT_NONE, 1e-5, False)]

    # vgg19.features[:10] (srgan/models.py:8-15): conv ReLU conv ReLU MaxPool conv ReLU conv ReLU MaxPool - the ReLU backward of a conv is
    # handed to its consumer (the next conv's input-gradient epilogue, or the pool's backward); the chain's last ReLU has none
    calls.clear()
    monkeypatch.setattr(F, "maxpool2", lambda x, relu_in=False: (calls.append(("maxpool2",) + (("relu_in",) if relu_in else ())),
                                                                 torch.zeros(x.shape[0], x.shape[1], x.shape[2] // 2, x.shape[3] // 2))[1])
    blk = nn.Sequential(nn.Conv2d(3, 8, 3, 1, 1), nn.ReLU(inplace=True), nn.Conv2d(8, 8, 3, 1, 1), nn.ReLU(inplace=True), nn.MaxPool2d(2, 2),
                        nn.Conv2d(8, 16, 3, 1, 1), nn.ReLU(inplace=True), nn.Conv2d(16, 16, 3, 1, 1), nn.ReLU(inplace=True))
    y = blk(torch.zeros(1, 3, 8, 8))
    assert tuple(y.shape) == (1, 16, 4, 4)
    c = ("conv2d", (1, 1, 1, 1), F.GATHER_ZERO, F.ACT_RELU, False)
    assert calls == [c + ("handed",), c + ("relu_in", "handed"), ("maxpool2", "relu_in"), c + ("handed",), c + ("relu_in",)]

    # PatchGAN tail (cyclegan/models.py:117-118): ZeroPad2d((1,0,1,0)) -> Conv2d(C, 1, 4, padding=1): pads fold into the conv
    calls.clear()
    blk = nn.Sequential(nn.ZeroPad2d((1, 0, 1, 0)), nn.Conv2d(8, 1, 4, padding=1))
    y = blk(torch.zeros(1, 8, 6, 6))
    assert tuple(y.shape) == (1, 1, 6, 6)
    assert calls == [("conv2d", (2, 2, 1, 1), F.GATHER_ZERO, F.ACT_NONE, False)]

    # SRGAN residual block head and up-sampling stage (srgan/models.py:22-24, 53-57): the single-slope PReLU rides in the
    # BatchNorm launches; a PixelShuffle(2) between them becomes the store index map of the same launches (PReLU commutes
    # with the permutation)
    calls.clear()
    monkeypatch.setattr(F, "pixel_shuffle", lambda x, r: (calls.append(("shuffle", r)), torch.zeros(x.shape[0], x.shape[1] // (r * r), x.shape[2] * r, x.shape[3] * r))[1])
    monkeypatch.setattr(F, "prelu", lambda x, w: (calls.append(("prelu",)), torch.zeros_like(x))[1])
    blk = nn.Sequential(nn.Conv2d(8, 8, 3, 1, 1), nn.BatchNorm2d(8, 0.8), nn.PReLU(), nn.Conv2d(8, 16, 3, 1, 1), nn.BatchNorm2d(16),
                        nn.PixelShuffle(upscale_factor=2), nn.PReLU())
    y = blk(torch.zeros(2, 8, 4, 4))
    assert tuple(y.shape) == (2, 4, 8, 8)
    assert calls == [("conv2d", (1, 1, 1, 1), F.GATHER_ZERO, F.ACT_NONE, False, "batch"), ("norm", False, F.ACT_NONE, 0.8, True, "prelu"),
                     ("conv2d", (1, 1, 1, 1), F.GATHER_ZERO, F.ACT_NONE, False, "batch"),
                     ("norm", False, F.ACT_NONE, 1e-5, True, "prelu", "shuffle", 2)]
    # other upscale factors keep the shuffle as its own launch behind the fused BatchNorm+PReLU
    calls.clear()
    y = nn.Sequential(nn.BatchNorm2d(18), nn.PixelShuffle(upscale_factor=3), nn.PReLU())(torch.zeros(2, 18, 4, 4))
    assert tuple(y.shape) == (2, 2, 12, 12) and calls == [("norm", False, F.ACT_NONE, 1e-5, True, "prelu"), ("shuffle", 3)]
    # a per-channel PReLU (num_parameters = C) is not the reference's layer: stays a separate launch
    calls.clear()
    nn.Sequential(nn.BatchNorm2d(8), nn.PReLU(8))(torch.zeros(2, 8, 4, 4))
    assert calls == [("norm", False, F.ACT_NONE, 1e-5, True), ("prelu",)]

    # eval mode: Dropout2d is the identity, BatchNorm uses running statistics (no batch-stat kernel, no counter)
    calls.clear()
    blk = nn.Sequential(nn.Conv2d(4, 8, 3, 2, 1), nn.LeakyReLU(0.2), nn.Dropout2d(0.25), nn.BatchNorm2d(8)).eval()
    blk(torch.zeros(1, 4, 4, 4))
    assert calls == [("conv2d", (1, 1, 1, 1), F.GATHER_ZERO, F.ACT_LRELU, False), ("norm", False, F.ACT_NONE, 1e-5, False)]

    # fusion off: every layer is its own launch family
    calls.clear()
    nn.set_fusion(False)
    try:
        nn.Sequential(nn.ZeroPad2d((1, 0, 1, 0)), nn.Conv2d(8, 1, 4, padding=1))(torch.zeros(1, 8, 6, 6))
    finally:
        nn.set_fusion(True)
    assert [c[0] for c in calls] == ["gather2d", "conv2d"]


def test_dropout_mask_plan(monkeypatch):
    """One Philox launch per training step for all dropout masks: the sequence one step asks for is the plan of the next
    (views of one flat buffer); departures from the plan fall back to single launches (no kernels run here)."""
    import pytorch_gan_amd.functional as F
    import pytorch_gan_amd.nn as nn

    calls = []

    def fake(shape, p, seed, counter, device):
        calls.append(tuple(shape))
        return torch.arange(float(nn._numel(shape))).view(shape)

    monkeypatch.setattr(F, "rand_mask", fake)
    monkeypatch.setattr(nn._DropoutRNG, "counter", classmethod(lambda cls, d: None))
    monkeypatch.setattr(nn._MaskPlan, "plans", {})

    def step(shapes, p=0.25):
        with F.weight_cache_scope():
            return [nn._next_mask(s, p, "cpu") for s in shapes]

    seq = [(4, 16), (4, 32), (4, 16), (4, 32)]
    step(seq)
    assert calls == seq                                   # first step: recorded, single launches
    calls.clear()
    out = step(seq)
    assert calls == [(384,)]                              # second step: one launch for all four
    assert [tuple(o.shape) for o in out] == seq and [o.flatten()[0].item() for o in out] == [0.0, 64.0, 192.0, 256.0]
    calls.clear()
    step(seq + [(4, 8)])
    assert calls == [(384,), (4, 8)]                      # an extra request: its own launch ...
    calls.clear()
    step(seq + [(4, 8)])
    assert calls == [(416,)]                              # ... and part of the next plan
    calls.clear()
    step([(4, 32)])
    assert calls == [(4, 32)]                             # different first request: no batch
    calls.clear()
    assert nn._next_mask((4, 16), 0.25, "cpu") is not None and calls == [(4, 16)]   # outside a step: never batched


def test_splitk_rule_and_workspace_size():
    """Host-side rules of the split-K conv entry points (no launch): the workspace is 1024 tickets + 1024 64x64 fp32 slabs; a
    launch is a candidate only with 16-byte channel vectors, >= 32 source channels, more than 4 output channels and at most
    128 64x64 tiles (pix2pix/models.py:62-71 inner levels yes, a CycleGAN residual conv no)."""
    from pytorch_gan_amd._lib import lib

    assert lib.migan_conv_splitk_workspace() == 1024 * 4 + 1024 * 64 * 64 * 4
    assert lib.migan_conv_splitk_applies(1, 512, 512, 1) == 1          # 512->512 on one output pixel
    assert lib.migan_conv_splitk_applies(64, 512, 512, 1) == 1
    assert lib.migan_conv_splitk_applies(256, 256, 512, 4) == 1        # stride-2 dgrad, 4 parity classes of 256 pixels
    assert lib.migan_conv_splitk_applies(32768, 256, 256, 1) == 0      # R256 at batch 8: 2048 tiles
    assert lib.migan_conv_splitk_applies(64, 512, 16, 1) == 0          # Ci < 32: register-staged kernels
    assert lib.migan_conv_splitk_applies(64, 3, 512, 1) == 0           # thin-N output: VALU kernels
    assert lib.migan_conv_splitk_applies(0, 512, 512, 1) == 0


def test_bucket_layout_and_run_ranks_helper():
    """optim.bucket_layout: 256-byte aligned slots in parameter order; tests/util.run_ranks: a launcher whose grandchild
    outlives the time limit is killed with all its descendants (what keeps a hung multi-rank test from leaving ranks on the GPU)."""
    import subprocess
    import time

    from pytorch_gan_amd.optim import bucket_layout
    from util import run_ranks

    offs, total = bucket_layout([1, 64, 65, 128, 3])
    assert offs == [0, 64, 128, 256, 384] and total == 448
    assert bucket_layout([]) == ([], 0)
    marker = "sleep_marker_%d" % os.getpid()
    code = ("import subprocess, sys, time; subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(300)  # %s']); "
            "time.sleep(300)" % marker)
    t0 = time.time()
    rc, _, _ = run_ranks([sys.executable, "-c", code], ROOT, dict(os.environ), 2)
    assert rc is None and time.time() - t0 < 30
    time.sleep(0.2)
    left = subprocess.run("ps -eo args | grep '%s' | grep -v grep | wc -l" % marker, shell=True, capture_output=True, text=True)
    assert int(left.stdout.strip()) == 0
    rc, out, _ = run_ranks([sys.executable, "-c", "print('ok')"], ROOT, dict(os.environ), 30)
    assert rc == 0 and out.strip() == "ok"


def test_rocpd_stats_counts_steps_from_the_trace(tmp_path):
    """tools/rocpd_stats.py --per-step: the per-step header comes from the optimiser launches in the trace (warm-up steps
    are traced too), not from the --steps the bench was asked for."""
    import sqlite3
    import subprocess
    import sys

    db = str(tmp_path / "t_results.db")
    c = sqlite3.connect(db)
    c.execute("create table kernels(name text, start int, end int, grid_x int, grid_y int, grid_z int)")
    for _ in range(12):   # 12 steps in the trace: 3 optimiser launches + 10 conv launches each
        c.executemany("insert into kernels values(?,?,?,?,?,?)", [("adam_kernel(AdamTensor const*)", 0, 1000, 1, 1, 1)] * 3)
        c.executemany("insert into kernels values(?,?,?,?,?,?)", [("conv", 0, 5000, 2, 1, 1)] * 10)
    c.commit()
    c.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rocpd_stats.py"), db, "5", "5", "--by-grid", "--per-step",
                        "adam_kernel=3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    head = r.stdout.splitlines()[0]
    assert "13.0 launches per step" in head and "12 steps in the trace" in head, head


def test_pmc_step_joins_counter_rows_to_roofline_groups_by_launch_ordinal(tmp_path):
    """tools/pmc_step.py: bench.py --pmc-log names, per call of a roofline group, the ordinals of the library launches it issued; the
    pass's dispatch list filtered to library kernels (ATen / runtime kernels interleave freely) is the same sequence.  Synthetic pass:
    two calls of one group (main kernel + reduce), one norm call, ATen kernels in between; a pass with a missing dispatch is refused."""
    import csv
    import json
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake_root = tmp_path / "repo"
    (fake_root / "tools").mkdir(parents=True)
    (fake_root / "profiles").mkdir()
    shutil.copy(os.path.join(root, "tools", "pmc_step.py"), fake_root / "tools" / "pmc_step.py")
    seq = [("void at::native::fill<float>(x)", 1 << 20), ("pack_kernel(float*)", 256), ("void wgrad_dma_kernel<64, 128>(WgradGeom)", 49152),
           ("void at::native::add(x)", 256), ("upconv_wgrad_reduce_kernel(float const*)", 1024), ("void norm_apply_kernel<4>(float*)", 4096),
           ("void wgrad_dma_kernel<64, 128>(WgradGeom)", 49152), ("upconv_wgrad_reduce_kernel(float const*)", 1024)]
    log = {"workload": "dcgan", "steps": 2, "total_launches": 6, "segments": [
        {"group": "upconv_wgrad[2x 4->4 @8]", "first": 1, "last": 3, "dense": 9e9, "executed": 4e9},
        {"group": "norm_apply[1x64x4]", "first": 3, "last": 4, "dense": 64e6, "executed": -1.0},
        {"group": "upconv_wgrad[2x 4->4 @8]", "first": 4, "last": 6, "dense": 9e9, "executed": 4e9}]}

    def make_pass(name, counters, drop=None):
        d = fake_root / name
        (d / "sub").mkdir(parents=True)
        json.dump(log, open(d / "segments.json", "w"))
        with open(d / "sub" / "p_counter_collection.csv", "w", newline="") as fh:
            wr = csv.writer(fh)
            wr.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
            for i, (nm, grid) in enumerate(seq):
                if i == drop:
                    continue
                for cn, val in counters(nm).items():
                    wr.writerow([i + 1, nm, grid, cn, val, 1000 * i, 1000 * i + (300 if "wgrad_dma" in nm else 20)])
        return str(d)

    sq = make_pass("sq", lambda nm: {"GRBM_GUI_ACTIVE": 1000.0, "SQ_VALU_MFMA_BUSY_CYCLES": 64000.0 if "wgrad_dma" in nm else 0.0})
    fe = make_pass("fetch", lambda nm: {"FETCH_SIZE": 100.0 if "wgrad_dma" in nm else (31.25 if "norm" in nm else 4.0)})
    wr_ = make_pass("write", lambda nm: {"WRITE_SIZE": 50.0 if "wgrad_dma" in nm else (62.5 if "norm" in nm else 2.0)})
    bad = make_pass("bad", lambda nm: {"FETCH_SIZE": 1.0}, drop=4)
    r = subprocess.run([sys.executable, str(fake_root / "tools" / "pmc_step.py"), "r99", sq, fe, wr_, bad], capture_output=True, text=True)
    assert r.returncode == 0 and "REFUSED" in r.stdout and "bad" in r.stdout, r.stdout + r.stderr
    tab = json.load(open(fake_root / "profiles" / "r99_pmc_kernels.json"))
    g = tab["upconv_wgrad[2x 4->4 @8]"]
    assert g["symbol"].startswith("wgrad_dma_kernel<64, 128>") and g["kernels_per_launch"] == {"wgrad_dma_kernel<64, 128>": 1.0, "upconv_wgrad_reduce_kernel": 1.0}
    assert g["hbm_bytes_per_launch"] == int((2 * (100.0 + 4.0) + (50.0 + 2.0)) * 1024)      # main kernel + reduce, per call
    assert abs(g["mfma_busy_frac"] - 64000.0 / (128 * 2000.0)) < 1e-4                        # GUI_ACTIVE of both kernels of the call
    assert g["launch_us_under_pmc"] == 0.3 and g["executed_gflop_per_launch"] == 4.0
    n = tab["norm_apply[1x64x4]"]
    assert n["algorithmic_mb_per_call"] == 64.0 and n["hbm_bytes_per_launch"] == int((2 * 31.25 + 62.5) * 1024)
    assert [os.path.basename(p_) for p_ in tab["_source"]["passes"]] == ["fetch", "sq", "write"]


def test_weight_gradient_stream_protocol(monkeypatch):
    """functional._Fork in its deferred form (weight-gradient launches of a step body on their own stream): which launches fork, what is
    joined when - with recording stand-ins for the HIP streams (no GPU: the stream objects only have to remember who waited for whom)."""
    import pytorch_gan_amd.functional as F

    log = []

    class FakeStream:
        def __init__(self, name):
            self.name, self.cuda_stream = name, hash(name) & 0xffff

        def wait_stream(self, other):
            log.append((self.name, "waits", other.name))

        def __enter__(self):
            log.append((self.name, "enter"))

        def __exit__(self, *a):
            log.append((self.name, "exit"))

    main = FakeStream("main")
    made = []

    def new_stream(device=None):
        made.append(FakeStream("side%d" % len(made)))
        return made[-1]

    class FakeTensor:
        def __init__(self):
            self.recorded = []

        def record_stream(self, st):
            self.recorded.append(st.name)

    dev = torch.device("cuda", 0)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda d=None: main)
    monkeypatch.setattr(torch.cuda, "Stream", new_stream)
    monkeypatch.setattr(torch.cuda, "stream", lambda st: st)
    monkeypatch.setattr(F, "_SIDE_STREAMS", {})
    monkeypatch.setattr(F, "_PENDING_WGRAD", {})
    monkeypatch.setattr(F, "_PENDING_READS", [])
    capturing = [False]
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: capturing[0])
    monkeypatch.setattr(F, "ensure_splitk_ws", lambda device, stream: None)   # a new side stream gets its split-K workspace (device memory)
    big, small = F._WGRAD_STREAM_MIN, F._WGRAD_STREAM_MIN - 1
    with torch.no_grad():
        # outside a step body (no weight_cache_scope): never deferred - a bare loss.backward() reads .grad right away
        assert not F._Fork(dev, big, True).on
        with F.weight_cache_scope():
            f = F._Fork(dev, small, True)
            assert not f.on                                   # below the threshold: in line with its layer's backward
            f = F._Fork(dev, big, True)
            assert f.on and f.defer
            dy, xs = FakeTensor(), FakeTensor()
            with f:
                pass
            f.join((None, None), (dy, xs, None))              # everything went into gradient slots: join deferred
            assert log == [("side0", "waits", "main"), ("side0", "enter"), ("side0", "exit")]
            assert dy.recorded == xs.recorded == ["side0"] and len(F._PENDING_WGRAD) == 1
            g = F._Fork(dev, big, True)
            with g:
                pass
            g.join((object(), None), (dy,))                   # a gradient returned to autograd: joined at once
            assert log[-1] == ("main", "waits", "side0")
            F.join_wgrad_streams()
            assert log[-1] == ("main", "waits", "side0") and not F._PENDING_WGRAD
            n = len(log)
            F.join_wgrad_streams()                            # nothing pending: no wait
            assert len(log) == n
            # one_wgrad_stream(): every size, ONE stream per device whatever the forking stream, and a returned gradient is an error
            with F.one_wgrad_stream():
                h = F._Fork(dev, 16, True)
                assert h.on and h.defer and h.key == (0, "all")
                with h:
                    pass
                with pytest.raises(RuntimeError):
                    h.join((object(),), ())
                h.join((None,), ())
            assert (0, "all") in F._PENDING_WGRAD
            F.join_wgrad_streams()
            # while a hipGraph is being recorded the tensors a deferred launch reads are HELD until the join instead of marked with
            # record_stream (the allocator would keep a marked block for the whole recording)
            capturing[0] = True
            k = F._Fork(dev, big, True)
            t1 = FakeTensor()
            with k:
                pass
            k.join((None,), (t1, None))
            assert t1.recorded == [] and F._PENDING_READS == [t1]
            F.join_wgrad_streams()
            assert F._PENDING_READS == [] and not F._PENDING_WGRAD
            capturing[0] = False
    # second-order backward (grad mode on): never forked
    with F.weight_cache_scope():
        assert not F._Fork(dev, big, True).on


def test_fused_generator_plan_steps_aside_for_cross_replica_batchnorm(monkeypatch):
    """The fused WGAN-GP generator kernels (csrc/mlp_fused.hip) take BatchNorm1d statistics over the rows of ONE rank.  With
    dp.enable_sync_batchnorm() at world size > 1 the generator of wgan_gp.py:42-65 (Linear, BatchNorm1d(eps 0.8), LeakyReLU groups) must go
    through the modules, whose norm calls gather the moments - the plan reports itself unusable; a generator without BatchNorm and
    world size 1 keep the fused path."""
    import types

    import pytorch_gan_amd.functional as F
    from pytorch_gan_amd import steps

    def gen(bn):
        layers = [torch.nn.Linear(100, 128), torch.nn.LeakyReLU(0.2), torch.nn.Linear(128, 256)]
        if bn:
            layers.append(torch.nn.BatchNorm1d(256, 0.8))
        layers += [torch.nn.LeakyReLU(0.2), torch.nn.Linear(256, 1024), torch.nn.Tanh()]
        g = torch.nn.Module()
        g.model, g.img_shape = torch.nn.Sequential(*layers), (1, 32, 32)
        return g.train()

    z = torch.randn(64, 100)
    with_bn, without = steps._GeneratorFusedPlan(gen(True), z), steps._GeneratorFusedPlan(gen(False), z)
    assert with_bn.ok and without.ok
    assert with_bn.usable(z) and without.usable(z)
    monkeypatch.setattr(F, "_SYNC_BN", types.SimpleNamespace(world=2))
    assert not with_bn.usable(z) and without.usable(z)
    monkeypatch.setattr(F, "_SYNC_BN", types.SimpleNamespace(world=1))
    assert with_bn.usable(z)
