#!/usr/bin/env python
"""Per-shape timing of the conv-family launchers (C ABI called directly, HIP events on the launch stream).

    python tools/conv_microbench.py [--shapes dcgan|cyclegan|srgan] [--iters 20] [--only fwd,dgrad,wgrad]

Prints one line per (layer, direction): time, algorithmic TFLOP/s, fraction of the 157.3 TF fp32-MFMA peak.
Used for tuning and as the target command of the rocprofv3 --pmc passes (profiles/).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

# name, N, Ci, H, W, Co, k, stride, pad, gather(0 zero,1 reflect,2 up2)
SHAPES = {
    "dcgan": [
        ("G.conv1 up2 128->128 @32", 128, 128, 16, 16, 128, 3, 1, 1, 2),
        ("G.conv2 up2 128->64 @64", 128, 128, 32, 32, 64, 3, 1, 1, 2),
        ("G.conv3 64->1 @64", 128, 64, 64, 64, 1, 3, 1, 1, 0),
        ("D.conv1 1->16 s2", 128, 1, 64, 64, 16, 3, 2, 1, 0),
        ("D.conv2 16->32 s2", 128, 16, 32, 32, 32, 3, 2, 1, 0),
        ("D.conv3 32->64 s2", 128, 32, 16, 16, 64, 3, 2, 1, 0),
        ("D.conv4 64->128 s2", 128, 64, 8, 8, 128, 3, 2, 1, 0),
        ("D2.conv3 32->64 s2 b256", 256, 32, 16, 16, 64, 3, 2, 1, 0),
        ("D2.conv4 64->128 s2 b256", 256, 64, 8, 8, 128, 3, 2, 1, 0),
    ],
    "cyclegan": [
        ("c7s1-64 3->64 reflect", 8, 3, 256, 256, 64, 7, 1, 3, 1),
        ("d128 64->128 s2", 8, 64, 256, 256, 128, 3, 2, 1, 0),
        ("d256 128->256 s2", 8, 128, 128, 128, 256, 3, 2, 1, 0),
        ("R256 reflect 256->256 @64", 8, 256, 64, 64, 256, 3, 1, 1, 1),
        ("R256 bs1 256->256 @64", 1, 256, 64, 64, 256, 3, 1, 1, 1),
        ("u128 up2 256->128 @128", 8, 256, 64, 64, 128, 3, 1, 1, 2),
        ("u64 up2 128->64 @256", 8, 128, 128, 128, 64, 3, 1, 1, 2),
        ("c7s1-3 64->3 reflect", 8, 64, 256, 256, 3, 7, 1, 3, 1),
        ("D.c1 3->64 k4s2", 8, 3, 256, 256, 64, 4, 2, 1, 0),
        ("D.c2 64->128 k4s2", 8, 64, 128, 128, 128, 4, 2, 1, 0),
        ("D.c3 128->256 k4s2", 8, 128, 64, 64, 256, 4, 2, 1, 0),
        ("D.c4 256->512 k4s2", 8, 256, 32, 32, 512, 4, 2, 1, 0),
    ],
    # pix2pix/models.py:62-78 at 256x256, bs 1: "dgrad" of a down-conv row is also the ConvTranspose2d forward of the same
    # geometry (UNetUp), e.g. "u2" = ConvTranspose2d(1024, 512, 4, 2, 1) on 2x2
    "pix2pix": [
        ("d2 64->128 @128", 1, 64, 128, 128, 128, 4, 2, 1, 0),
        ("d3 128->256 @64", 1, 128, 64, 64, 256, 4, 2, 1, 0),
        ("d4 256->512 @32", 1, 256, 32, 32, 512, 4, 2, 1, 0),
        ("d5 512->512 @16", 1, 512, 16, 16, 512, 4, 2, 1, 0),
        ("d6 512->512 @8", 1, 512, 8, 8, 512, 4, 2, 1, 0),
        ("d7 512->512 @4", 1, 512, 4, 4, 512, 4, 2, 1, 0),
        ("d8 512->512 @2", 1, 512, 2, 2, 512, 4, 2, 1, 0),
        ("u2 512<-1024 @4", 1, 512, 4, 4, 1024, 4, 2, 1, 0),
        ("u3 512<-1024 @8", 1, 512, 8, 8, 1024, 4, 2, 1, 0),
        ("u4 512<-1024 @16", 1, 512, 16, 16, 1024, 4, 2, 1, 0),
        ("u5 256<-1024 @32", 1, 256, 32, 32, 1024, 4, 2, 1, 0),
        ("u6 128<-512 @64", 1, 128, 64, 64, 512, 4, 2, 1, 0),
        ("u7 64<-256 @128", 1, 64, 128, 128, 256, 4, 2, 1, 0),
        ("D.c4 256->512 @32", 1, 256, 32, 32, 512, 4, 2, 1, 0),
    ],
    "srgan": [
        ("D.c1 3->64 @384", 16, 3, 384, 384, 64, 3, 1, 1, 0),
        ("G.c1 3->64 k9 @96", 16, 3, 96, 96, 64, 9, 1, 4, 0),
        ("res 64->64 @96", 16, 64, 96, 96, 64, 3, 1, 1, 0),
        ("up 64->256 @192", 16, 64, 192, 192, 256, 3, 1, 1, 0),
        ("conv3 9x9 64->3 @384", 16, 64, 384, 384, 3, 9, 1, 4, 0),
        ("vgg 64->64 @384", 16, 64, 384, 384, 64, 3, 1, 1, 0),
        ("vgg 256->256 @96", 16, 256, 96, 96, 256, 3, 1, 1, 0),
        ("D 512->512 s2 @48", 16, 512, 48, 48, 512, 3, 2, 1, 0),
        ("D 64->64 s2 @384", 16, 64, 384, 384, 64, 3, 2, 1, 0),
        ("D 128->128 s2 @192", 16, 128, 192, 192, 128, 3, 2, 1, 0),
        ("D 256->256 s2 @96", 16, 256, 96, 96, 256, 3, 2, 1, 0),
    ],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="dcgan")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--match", default="")
    ap.add_argument("--repeat", type=int, default=1, help="time every direction this many times; print min and median")
    ap.add_argument("--dirs", default="", help="exact direction names (fwd,ufwd,rdgrad,twgrad,...) instead of --only's families")
    ap.add_argument("--pmc-log", default="", help="write one JSON line per timed (layer, direction) and launch a 1-element "
                                                  "axpby marker kernel in front of each, so that a rocprofv3 --pmc pass over "
                                                  "this command can be cut into per-direction segments (tools/pmc_kernels.py)")
    args = ap.parse_args()
    import pytorch_gan_amd  # noqa: F401
    from pytorch_gan_amd._lib import check, lib

    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    only = set(args.only.split(","))
    exact = set(args.dirs.split(",")) if args.dirs else None
    marker = torch.zeros(4, device=dev)
    plog = open(args.pmc_log, "w") if args.pmc_log else None
    for name, N, Ci, H, W, Co, k, s, p, gth in SHAPES[args.shapes]:
        if args.match and args.match not in name:
            continue
        HL, WL = (2 * H, 2 * W) if gth == 2 else (H, W)
        Ho, Wo = (HL + 2 * p - k) // s + 1, (WL + 2 * p - k) // s + 1
        flops = 2.0 * N * Ho * Wo * Co * Ci * k * k
        x = torch.rand(N * H * W * Ci, device=dev) * 2 - 1
        w = (torch.rand(Co * k * k * Ci, device=dev) * 2 - 1) * 0.05
        y = torch.empty(N * Ho * Wo * Co, device=dev)
        dy = torch.rand(N * Ho * Wo * Co, device=dev) * 2 - 1
        # dgrad of a gathered conv targets the logical (padded / upsampled) extent, as functional.py does
        if gth == 1:
            Hd, Wd, pd = H + 2 * p, W + 2 * p, 0
        elif gth == 2:
            Hd, Wd, pd = HL, WL, p
        else:
            Hd, Wd, pd = H, W, p
        dx = torch.empty(N * Hd * Wd * Ci, device=dev)
        dw = torch.empty_like(w)
        nb = lib.migan_conv2d_wgrad_workspace(N, Ho, Wo, Co, k, k, Ci)
        ws = torch.empty(max(nb // 4, 1), device=dev)
        skb = lib.migan_conv_splitk_workspace() if os.environ.get("MIGAN_SPLITK", "1") == "1" else 0
        sk = torch.zeros(max(skb // 4, 1), device=dev)
        skp = sk.data_ptr() if skb else None
        calls = {
            "fwd": lambda: lib.migan_conv2d_fwd_ws(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), N, H, W, Ci, Ho, Wo,
                                                   Co, k, k, s, p, p, gth, 0, 0.0, skp, skb, st),
            "dgrad": lambda: lib.migan_conv2d_dgrad_ws(dy.data_ptr(), w.data_ptr(), None, dx.data_ptr(), N, Hd, Wd, Ci, Ho,
                                                       Wo, Co, k, k, s, pd, pd, 0, 0.0, skp, skb, st),
            "wgrad": lambda: lib.migan_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, H,
                                                    W, Ci, Ho, Wo, Co, k, k, s, p, p, gth, 0, None, 0, None, 0, st),
        }
        dirs = ["fwd", "dgrad", "wgrad"]
        if gth == 1 and k == 3 and s == 1 and p == 1 and Co % 4 == 0 and Co >= 8 and Ci > 4:
            # ReflectionPad2d(1)+Conv3x3 input gradient straight into H x W (pad-1 dgrad + added ring terms); "dgrad" above
            # is the padded-extent launch it replaces (which additionally needs the fold pass, timed as "fold")
            dxr = torch.empty(N * H * W * Ci, device=dev)
            calls["rdgrad"] = lambda: lib.migan_conv2d_dgrad_reflect1_ws(dy.data_ptr(), w.data_ptr(), dxr.data_ptr(), N, H, W,
                                                                         Ci, Co, skp, skb, st)
            calls["fold"] = lambda: lib.migan_gather2d_bwd(dx.data_ptr(), dxr.data_ptr(), N, H, W, Ci, Hd, Wd, p, p, 1, st)
            dirs += ["rdgrad", "fold"]
        if gth == 2 and k == 3 and s == 1 and p == 1 and Co % 4 == 0 and Ci % 4 == 0:
            # phase-collapsed Upsample(2)->Conv3x3 (what the product path runs); TF stays ALGORITHMIC (dense FLOPs)
            wf = torch.empty(Co * 16 * Ci, device=dev)
            wd = torch.empty_like(wf)
            check(lib.migan_upconv3x3_pack(w.data_ptr(), wf.data_ptr(), wd.data_ptr(), Co, Ci, st), "pack")
            dxs = torch.empty(N * H * W * Ci, device=dev)
            nbu = lib.migan_upconv3x3_wgrad_workspace(N, H, W, Co, Ci)
            wsu = torch.empty(max(nbu // 4, 1), device=dev)
            calls["ufwd"] = lambda: lib.migan_upconv3x3_fwd(x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), N, H, W, Ci,
                                                            Co, 0, 0.0, st)
            calls["udgrad"] = lambda: lib.migan_upconv3x3_dgrad(dy.data_ptr(), wd.data_ptr(), dxs.data_ptr(), N, H, W,
                                                                Ci, Co, st)
            calls["uwgrad"] = lambda: lib.migan_upconv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(),
                                                                wsu.data_ptr(), nbu, N, H, W, Ci, Co, 0, None, 0, None, 0, st)
            dirs += ["ufwd", "udgrad", "uwgrad"]
        if lib.migan_thin_toeplitz_ok(Co, k, k, Ci, s, gth) == 1:
            # width-Toeplitz expansion of the thin-N conv (what the product path runs above 64k pixels): tfwd = R x 1 GEMM +
            # diagonal sum; texpand = dy -> q (shared by both gradients); twgrad / tdgrad consume q
            cop = lib.migan_thin_toeplitz_cols(Co, k)
            wtd = torch.empty(2 * cop * k * Ci, device=dev)
            check(lib.migan_thin_toeplitz_pack(w.data_ptr(), wtd.data_ptr(), wtd.data_ptr() + 4 * cop * k * Ci, Co, Ci, k, k, st),
                  "tpack")
            nbp = lib.migan_thin_toeplitz_workspace(N, Ho, W, Co, k)
            pq = torch.empty(nbp // 4, device=dev)
            nbw = lib.migan_thin_toeplitz_wgrad_workspace(N, Ho, W, Ci, Co, k, k)
            wsw = torch.empty(max(nbw // 4, 1), device=dev)
            nbd = lib.migan_thin_toeplitz_dgrad_workspace(N, H, W, Ci, Ho, k, gth)
            wsd = torch.empty(max(nbd // 4, 1), device=dev)
            dxt = torch.empty(N * H * W * Ci, device=dev)
            calls["tfwd"] = lambda: lib.migan_thin_toeplitz_fwd(x.data_ptr(), wtd.data_ptr(), None, y.data_ptr(), pq.data_ptr(),
                                                                nbp, N, H, W, Ci, Ho, Wo, Co, k, k, p, p, gth, 0, 0.0, st)
            calls["texpand"] = lambda: lib.migan_thin_toeplitz_expand(dy.data_ptr(), pq.data_ptr(), N, Ho, Wo, Co, W, k, p, gth, st)
            calls["twgrad"] = lambda: lib.migan_thin_toeplitz_wgrad(x.data_ptr(), pq.data_ptr(), dw.data_ptr(), wsw.data_ptr(), nbw,
                                                                    N, H, W, Ci, Ho, Co, k, k, p, gth, 0, st)
            calls["tdgrad"] = lambda: lib.migan_thin_toeplitz_dgrad(pq.data_ptr(), wtd.data_ptr() + 4 * cop * k * Ci, dxt.data_ptr(),
                                                                    wsd.data_ptr(), nbd, N, H, W, Ci, Ho, Co, k, k, p, gth, st)
            dirs += ["tfwd", "texpand", "twgrad", "tdgrad"]
        if lib.migan_rgb_conv_ok(Ci, Co, k, k, s, gth, N * Ho * Wo) == 1:
            # image-input layer (csrc/rgb_conv.hip): xfwd = forward with LeakyReLU; xwgrad = activation backward + bias + weight gradient
            # in one launch (what replaces act_bwd_colsum + wgrad); abwd = the activation-backward + column-sum pass it makes unnecessary
            wk = w.view(Co, Ci, k, k).permute(2, 3, 1, 0).contiguous()
            bz = torch.zeros(Co, device=dev)
            calls["xfwd"] = lambda: lib.migan_rgb_conv_fwd(x.data_ptr(), wk.data_ptr(), bz.data_ptr(), y.data_ptr(), N, H, W, Ci, Ho, Wo, Co,
                                                           k, k, p, p, gth, 1, 0.2, 0, st)
            dirs += ["xfwd"]
            if lib.migan_rgb_conv_wgrad_ok(Ci, Co, k, k, s, gth, N * Ho * Wo) == 1:
                nbx = lib.migan_rgb_conv_wgrad_workspace(Co, k, k)
                wsx = torch.empty(nbx // 4, device=dev)
                db = torch.empty(Co, device=dev)
                calls["xwgrad"] = lambda: lib.migan_rgb_conv_wgrad(x.data_ptr(), dy.data_ptr(), y.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                                   wsx.data_ptr(), nbx, N, H, W, Ho, Wo, Co, k, k, p, p, gth, 1, 0.2, 0, 0, st)
                dirs += ["xwgrad"]
        if lib.migan_c64_conv_ok(N, H, W, Ci, Co, k, k, s, p, p, p, p, gth) == 1:
            # weight-stationary 64 -> 64 kernel (csrc/conv_c64.hip): cfwd = forward, cdgrad = input gradient (flipped pack); both are checked
            # against the general kernels' results here before they are timed
            wo = w.view(Co, k, k, Ci).permute(0, 3, 1, 2).contiguous()   # `w` above is OHWI for the general forward: OIHW for the pack
            wpf = torch.empty(lib.migan_c64_pack_floats(), device=dev)
            wpd = torch.empty_like(wpf)
            check(lib.migan_c64_pack(wo.data_ptr(), wpf.data_ptr(), 0, st), "c64 pack")
            yc = torch.empty_like(y)
            check(lib.migan_c64_conv_fwd(x.data_ptr(), wpf.data_ptr(), None, yc.data_ptr(), N, H, W, 0, 0.0, 0, None, None, None, None, 0, 0.0, None, st), "c64")
            check(calls["fwd"](), "fwd")
            torch.cuda.synchronize()
            print("   c64 fwd vs general kernel: rel %.2e" % float((yc - y).norm() / y.norm()))
            # the general dgrad takes IHWO weights; interpret `w` as OIHW-permuted for a consistent pair: w_ihwo[i][r][s][o] := wo[o][i][r][s]
            wi = wo.permute(1, 2, 3, 0).contiguous()
            check(lib.migan_c64_pack(wo.data_ptr(), wpd.data_ptr(), 1, st), "c64 pack flip")
            dxc = torch.empty_like(dx)
            check(lib.migan_c64_conv_fwd(dy.data_ptr(), wpd.data_ptr(), None, dxc.data_ptr(), N, H, W, 0, 0.0, 0, None, None, None, None, 0, 0.0, None, st), "c64d")
            check(lib.migan_conv2d_dgrad_ws(dy.data_ptr(), wi.data_ptr(), None, dx.data_ptr(), N, Hd, Wd, Ci, Ho, Wo, Co, k, k, s, pd, pd, 0, 0.0,
                                            skp, skb, st), "dgrad")
            torch.cuda.synchronize()
            print("   c64 dgrad vs general kernel: rel %.2e" % float((dxc - dx).norm() / dx.norm()))
            calls["cfwd"] = lambda: lib.migan_c64_conv_fwd(x.data_ptr(), wpf.data_ptr(), None, y.data_ptr(), N, H, W, 0, 0.0, 0, None, None,
                                                           None, None, 0, 0.0, None, st)
            calls["cdgrad"] = lambda: lib.migan_c64_conv_fwd(dy.data_ptr(), wpd.data_ptr(), None, dx.data_ptr(), N, H, W, 0, 0.0, 0, None,
                                                             None, None, None, 0, 0.0, None, st)
            nbc = lib.migan_c64_wgrad_workspace(N, H, W)
            wsc = torch.empty(nbc // 4, device=dev)
            dwc = torch.empty(Co * Ci * k * k, device=dev)
            check(lib.migan_c64_conv_wgrad(x.data_ptr(), dy.data_ptr(), dwc.data_ptr(), wsc.data_ptr(), nbc, N, H, W, 0, None, 0, None, 0, None, None,
                                           None, None, 0, 0.0, None, st), "c64w")
            check(calls["wgrad"](), "wgrad")
            torch.cuda.synchronize()
            print("   c64 wgrad vs general kernel: rel %.2e" % float((dwc - dw).norm() / dw.norm()))
            calls["cwgrad"] = lambda: lib.migan_c64_conv_wgrad(x.data_ptr(), dy.data_ptr(), dwc.data_ptr(), wsc.data_ptr(), nbc, N, H, W, 0, None, 0,
                                                               None, 0, None, None, None, None, 0, 0.0, None, st)
            dirs += ["cfwd", "cdgrad", "cwgrad"]
        if Co <= 3 and lib.migan_rgb_conv_ok(Co, Ci, k, k, s, gth, N * H * W) == 1 and p == 1 and gth == 0:
            # thin-OUTPUT layer (dcgan.py:62): odgrad = its input gradient on the image-input forward kernel, taps reversed
            wko = w.view(Co, Ci, k, k).permute(2, 3, 0, 1).contiguous()
            calls["odgrad"] = lambda: lib.migan_rgb_conv_fwd(dy.data_ptr(), wko.data_ptr(), None, dx.data_ptr(), N, H, W, Co, H, W, Ci, k, k,
                                                             p, p, 0, 0, 0.0, 1, st)
            dirs += ["odgrad"]
        for d in dirs:
            if exact is not None:
                if d not in exact:
                    continue
            elif d == "texpand":
                if not ({"wgrad", "dgrad"} & only):
                    continue
            elif d[0] == "t" and d[1:] in ("fwd", "wgrad", "dgrad"):
                if d[1:] not in only:
                    continue
            elif d.lstrip("urxoc") not in only and not (d == "fold" and "dgrad" in only):
                continue
            fn = calls[d]
            if plog:
                import json

                check(lib.migan_axpby(marker.data_ptr(), 1.0, None, 0.0, marker.data_ptr(), 1, st), "marker")
                up = d in ("ufwd", "udgrad", "uwgrad")
                plog.write(json.dumps({"layer": name, "dir": d, "calls": 3 + args.iters, "N": N, "Ci": Ci, "H": H, "W": W, "Co": Co,
                                       "k": k, "stride": s, "gather": gth, "Ho": Ho, "dense_flops": flops,
                                       "executed_flops": flops * (16.0 / 36.0 if up else 1.0)}) + "\n")
                plog.flush()
            for _ in range(3):
                check(fn(), d)
            torch.cuda.synchronize()
            times = []
            for _rep in range(max(1, args.repeat)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) / args.iters)
            times.sort()
            ms = times[0]
            tf = flops / (ms * 1e-3) / 1e12
            if args.repeat > 1:
                print("%-28s %-6s %9.1f us  %7.2f TF  %5.1f%%  median %9.1f us" % (name, d, ms * 1e3, tf, 100 * tf / 157.3,
                                                                                  times[len(times) // 2] * 1e3), flush=True)
                continue
            print("%-28s %-6s %9.1f us  %7.2f TF  %5.1f%%  (%.2f GFLOP)" % (name, d, ms * 1e3, tf, 100 * tf / 157.3, flops / 1e9),
                  flush=True)


if __name__ == "__main__":
    main()
