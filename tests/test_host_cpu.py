"""Host-side logic that needs no GPU: the packed-weight cache stamps, the direct-gradient slot rules and the
bookkeeping helpers of bench.py (kernel symbol names, committed PMC traffic table)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_weight_cache_stamps():
    import pytorch_gan_amd.functional as F
    from pytorch_gan_amd import optim

    calls = []

    def make():
        calls.append(1)
        return torch.full((1,), float(len(calls)))

    w = torch.zeros(4)
    F.set_weight_cache(True)
    try:
        assert F._packed(w, w, "k", make) is not F._packed(w, w, "k", make)  # no optimiser epoch -> never cached
        assert len(calls) == 2
        w._migan_epoch = next(optim._EPOCH)
        a = F._packed(w, w, "k", make)
        assert F._packed(w, w, "k", make) is a and len(calls) == 3            # same stamp -> hit
        assert F._packed(w, w, "other", make) is not a                        # cache is per pack kind
        w._migan_epoch = next(optim._EPOCH)                                   # an optimiser step
        b = F._packed(w, w, "k", make)
        assert b is not a
        w.add_(1.0)                                                            # version bump (copy_/load_state_dict)
        assert F._packed(w, w, "k", make) is not b
        e1, e2 = next(optim._EPOCH), next(optim._EPOCH)
        assert e2 > e1 > w._migan_epoch                                        # stamps are never re-used
    finally:
        F.set_weight_cache(False)
    n = len(calls)
    F._packed(w, w, "k", make)
    F._packed(w, w, "k", make)
    assert len(calls) == n + 2                                                 # cache off: always repack


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


def test_bench_symbols_and_traffic_table():
    import bench

    assert bench.kernel_symbol("upconv_wgrad_64x128[128->64@64]") == "wgrad_inc_kernel<64, 128, true, false>"
    assert bench.kernel_symbol("upconv_fwd_igemm_1128064[128->64@64]") == "igemm_pipe_kernel<128, 64, 2, 2, false, true>"
    assert bench.kernel_symbol("upconv_fwd_igemm_1128128[128->128@32]") == "igemm_pipe_kernel<128, 128, 2, 2, false, false>"
    assert bench.kernel_symbol("igemm_1128032") == "igemm_pipe_kernel<128, 32, 4, 1, false, false>"
    tab = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
    for name, ent in tab.items():
        assert bench.kernel_symbol(name) == ent["symbol"], name
        b, src = bench.pmc_traffic(name)
        assert b == ent["hbm_bytes_per_launch"] and "r01_pmc_traffic.json" in src
        rebuilt = (2 * ent["fetch_size_kb_reported"] + ent["write_size_kb_reported"]) * 1024  # FETCH x2 (gfx950), WRITE x1
        assert abs(ent["hbm_bytes_per_launch"] - rebuilt) <= 1e-5 * rebuilt
        assert ent["hbm_bytes_per_launch"] >= ent["algorithmic_bytes_per_launch"]
    assert bench.pmc_traffic("no_such_kernel") == (None, None)
    # 3 G forwards-equivalents + 8 D forwards-equivalents: 2.811 GFLOP per image
    assert abs(bench.dcgan_flops_per_image() / 1e9 - 2.8107) < 1e-3
