#!/usr/bin/env python3
"""A/B sweep of the LDS-DMA ring GEMM tile shapes on the hot shapes of the step (graph-replayed launches: GPU time
without Python launch overhead), each checked against torch on a row sample.  PDMK_ENV_DYNAMIC=1 is set here."""
import os, sys
os.environ["PDMK_ENV_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
import torch.nn.functional as F
from pdm import _pdmk as k

dev = torch.device("cuda:0")
dt = torch.bfloat16
REP = 10


def gtime(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REP) * 1e3   # us


def mk_lin(M, N, K, res=True):
    x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    r = torch.randn(M, N, device=dev).to(dt) if res else None
    bias = torch.randn(N, device=dev)
    fn = lambda: k.gemm(x, w, y, M, N, K, K, K, N, R=r, ldr=N, bias=bias)
    rows = torch.randint(0, M, (64,), device=dev)
    def check():
        ref = x[rows].float() @ w.float().t() + bias + (r[rows].float() if res else 0)
        err = (y[rows].float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
        return err
    return f"lin M{M} N{N} K{K}", 2.0 * M * N * K, fn, check


def mk_conv(B, H, Ci, Co, sk=1):
    x = torch.randn(B * H * H, Ci, device=dev).to(dt)
    w4 = (torch.randn(Co, Ci, 3, 3, device=dev) * (9 * Ci) ** -0.5).to(dt)
    w = w4.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
    M = B * H * H
    y = torch.empty(M, Co, device=dev, dtype=dt)
    yf = torch.zeros(M, Co, device=dev)
    bias = torch.randn(Co, device=dev)
    if sk == 1:
        fn = lambda: k.gemm(x, w, y, M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(B, H, H, Ci, H, H, 0, Ci), bias=bias)
    else:
        def fn():
            k.gemm(x, w, yf, M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(B, H, H, Ci, H, H, 0, Ci), bias=bias, out_f32=True, splitk=sk)
    def check():
        if sk > 1:
            yf.zero_(); fn(); out = yf
        else:
            out = y
        ref = F.conv2d(x.float().view(B, H, H, Ci).permute(0, 3, 1, 2)[:1], w4.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(H * H, Co)
        return (out[:H * H].float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
    return f"conv B{B} {H}x{H} {Ci}->{Co} sk{sk}", 2.0 * M * Co * 9 * Ci, fn, check


shapes = [mk_lin(32768, 320, 320), mk_lin(32768, 320, 1280), mk_lin(32768, 2560, 320, False), mk_lin(32768, 960, 320, False),
          mk_lin(8192, 640, 640), mk_lin(8192, 640, 2560), mk_lin(8192, 5120, 640, False), mk_lin(8192, 320, 640),
          mk_lin(2048, 1280, 1280), mk_lin(2048, 1280, 5120), mk_lin(2048, 10240, 1280, False), mk_lin(2048, 640, 1280),
          mk_lin(512, 1280, 1280), mk_lin(616, 1280, 1024, False), mk_lin(8192, 352, 608), mk_lin(32768, 224, 288),
          mk_conv(8, 64, 320, 320), mk_conv(8, 64, 640, 320), mk_conv(8, 32, 640, 640), mk_conv(8, 32, 320, 640), mk_conv(8, 32, 1280, 640),
          mk_conv(8, 16, 1280, 1280), mk_conv(8, 16, 640, 1280), mk_conv(8, 16, 2560, 1280, 4), mk_conv(8, 8, 1280, 1280, 5),
          mk_conv(8, 8, 2560, 1280, 11), mk_conv(8, 32, 352, 608), mk_conv(8, 64, 160, 288)]
names = ["old", "256x128", "256x160", "128x128", "128x160", "64x128", "64x160", "s128x128", "s64x128", "s64x160", "128x192", "64x192",
         "s128x160", "h256x160", "h256x128", "h128x160", "h128x128"]
cfgs = [(n, i) for i, n in enumerate(names)] + [("tuned", -1)]
print(f"{'shape':34s} " + " ".join(f"{c[0]:>8s}" for c in cfgs) + "   (us; TF/s of best)")
for name, fl, fn, check in shapes:
    row, errs = [], []
    for cname, cfg in cfgs:
        os.environ["PDMK_RING_CFG"] = str(cfg)
        if cname.startswith("h") and not name.startswith("conv"):
            row.append(float("inf")); errs.append(0.0)
            continue
        fn(); torch.cuda.synchronize()
        errs.append(check())
        row.append(gtime(fn))
    best = min(row)
    flag = "" if max(errs) < 2e-2 else f"  !! err {max(errs):.3f} @ {cfgs[errs.index(max(errs))][0]}"
    print(f"{name:34s} " + " ".join(f"{t:8.1f}" if t < 1e9 else "       -" for t in row) + f"   {fl / best / 1e6:7.1f} best={cfgs[row.index(best)][0]}{flag}")
