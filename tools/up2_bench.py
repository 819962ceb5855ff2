#!/usr/bin/env python3
"""Upsample2D conv (nearest x2 + 3x3) on one MI355X: the 3x3 form with the upsampling fused into the gather (conv_mode 2) against
the four 2x2 phase convs on the low-resolution image (conv_mode 5..12), forward / input gradient / weight gradient, at the
three upsampler shapes of SD-2.1 (B = 8).  python tools/up2_bench.py"""
import os, sys
os.environ["PDMK_ENV_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k
dev = torch.device("cuda:0"); dt = torch.bfloat16
REP = 5


def gtime(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (4 * REP) * 1e3


for (B, H, C) in [(8, 32, 640), (8, 16, 1280), (8, 8, 1280)]:
    W = H
    Ml, Mh = B * H * W, B * 4 * H * W
    x = torch.randn(Ml, C, device=dev).to(dt)
    w3m = torch.randn(C, 9 * C, device=dev) * 0.02
    w3 = w3m.to(dt)
    bias = torch.randn(C, device=dev)
    wp = torch.empty(4, C, 4 * C, device=dev, dtype=dt); wpt = torch.empty(C, 16 * C, device=dev, dtype=dt)
    k.up2_pack_weights(w3m, wp, wpt, C, C)
    y = torch.empty(Mh, C, device=dev, dtype=dt)
    dy = torch.randn(Mh, C, device=dev).to(dt)
    geo = lambda m, ld: (B, H, W, C, H, W, m, ld)
    fl3 = 2.0 * Mh * C * 9 * C
    fl2 = fl3 * 16 / 36
    print(f"== B{B} {H}x{H} -> {2*H}x{2*H}, C={C}: 3x3 form {fl3/1e9:.1f} GFLOP, phase form {fl2/1e9:.1f} GFLOP")
    t = gtime(lambda: k.gemm_auto(x, w3, y, Mh, C, 9 * C, 0, 9 * C, C, a_mode=k.A_CONV, conv=(B, H, W, C, 2 * H, 2 * W, 2, C), bias=bias))
    print(f"  fwd 3x3 (mode 2, tuned)            {t:8.1f} us  {fl3/t/1e6:7.1f} TF/s(3x3 flops)")

    def fwd_phases(grouped):
        with k.Recorder() as r:
            for p in range(4):
                k.gemm(x, wp[p], y, Ml, C, 4 * C, 0, 4 * C, C, a_mode=k.A_CONV, conv=geo(5 + p, C), bias=bias)
        if grouped:
            k.gemm_group(r.recs)
        else:
            for rec in r.recs: rec.run()
    for cand in (15, 16):
        os.environ["PDMK_RING_CFG"] = str(cand)
        t = gtime(lambda: fwd_phases(False))
        os.environ.pop("PDMK_RING_CFG")
        os.environ["PDMK_GROUP_CFG"] = str(cand)
        tg = gtime(lambda: fwd_phases(True))
        os.environ.pop("PDMK_GROUP_CFG")
        print(f"  fwd 4 phases cand {cand}: separate {t:8.1f} us ({fl2/t/1e6:6.1f} TF/s exec)   grouped {tg:8.1f} us ({fl2/tg/1e6:6.1f} TF/s exec)")
    # input gradient
    dxh = torch.empty(Mh, C, device=dev, dtype=dt); dxl = torch.empty(Ml, C, device=dev, dtype=dt)
    wt = torch.randn(C, 9 * C, device=dev).to(dt)

    def dgrad_old():
        k.gemm_auto(dy, wt, dxh, Mh, C, 9 * C, 0, 9 * C, C, a_mode=k.A_CONV, conv=(B, 2 * H, 2 * W, C, 2 * H, 2 * W, 0, C))
        k.pool2x2_sum(dxh, dxl, B, H, W, C)
    t = gtime(dgrad_old)
    print(f"  dgrad 3x3 at 2Hx2W + pool          {t:8.1f} us")
    for cand in (15, 16):
        os.environ["PDMK_RING_CFG"] = str(cand)
        def dgrad_new():
            for p in range(4):
                k.gemm(dy, wpt[:, p * 4 * C:(p + 1) * 4 * C], dxl, Ml, C, 4 * C, 0, 16 * C, C, a_mode=k.A_CONV, conv=geo(9 + p, C), accumulate=p > 0)
        t = gtime(dgrad_new)
        tm = gtime(lambda: k.gemm(dy, wpt, dxl, Ml, C, 16 * C, 0, 16 * C, C, a_mode=k.A_CONV, conv=geo(13, C)))
        os.environ.pop("PDMK_RING_CFG")
        print(f"  dgrad cand {cand}: 4 phases {t:8.1f} us ({fl2/t/1e6:6.1f} TF/s exec)   merged (mode 13) {tm:8.1f} us ({fl2/tm/1e6:6.1f} TF/s exec)")
    # weight gradient
    dw = torch.zeros(C, 9 * C, device=dev); db = torch.zeros(C, device=dev)
    t = gtime(lambda: k.wgrad(dy, x, dw, C, 9 * C, Mh, C, 0, b_mode=k.B_COLK_CONV, conv=(B, H, W, C, 2 * H, 2 * W, 2, C), colsum_out=db))
    print(f"  wgrad 3x3 (mode 2, tuned)          {t:8.1f} us  {fl3/t/1e6:7.1f} TF/s(3x3 flops)")
    dwp = torch.zeros(4, C, 4 * C, device=dev)
    sk = k.wgrad_plan(dy, x, C, 4 * C, Ml, C, 0, k.B_COLK_CONV, geo(5, C))

    def wgrad_new(grouped):
        k.zero_(dwp)
        with k.Recorder() as r:
            for p in range(4):
                k.gemm(dy, x, dwp[p], C, 4 * C, Ml, C, 0, 4 * C, a_mode=k.A_COLK, b_mode=k.B_COLK_CONV, conv=geo(5 + p, C),
                       out_f32=True, splitk=sk, accumulate=(sk == 1), dtype=k.BF16, colsum_out=db)
        if grouped:
            k.gemm_group(r.recs)
        else:
            for rec in r.recs: rec.run()
        k.up2_combine_wgrad(dwp, dw, C, C)
    t, tg = gtime(lambda: wgrad_new(False)), gtime(lambda: wgrad_new(True))
    print(f"  wgrad 4 phases (sk {sk}) + combine: separate {t:8.1f} us   grouped(tuned choice) {tg:8.1f} us ({fl2/tg/1e6:6.1f} TF/s exec)")
