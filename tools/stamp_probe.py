#!/usr/bin/env python3
"""Diagnostic (needs `make -C unlearn-ft_amd/csrc EXTRA=-DPDMK_STAMPS`): per-workgroup phase times of the LDS-DMA GEMM.
Stamps: 0 entry, 1 first K-tile landed, 2 main loop done, 3 epilogue done (shader cycles); 5/4 entry/exit (100 MHz)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import numpy as np
import torch
from pdm import _pdmk as k

dev = torch.device("cuda:0")
dt = torch.bfloat16


def stamps(nwg):
    buf = (ctypes.c_ulonglong * (nwg * 6))()
    assert k._lib.pdmk_debug_read_stamps(buf, nwg * 6) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 6).astype(np.int64)


def report(name, fn, nwg, flops):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    s = stamps(nwg)
    pro, main, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    life_rt = (s[:, 4] - s[:, 5]) * 10.0            # ns
    span = (s[:, 4].max() - s[:, 5].min()) * 10.0
    start_skew = (s[:, 5] - s[:, 5].min()) * 10.0
    clk = np.median((s[:, 3] - s[:, 0]) / np.maximum(life_rt, 1)) # cycles per ns
    md = lambda a: float(np.median(a))
    print(f"{name:40s} wg={nwg:5d} event {e0.elapsed_time(e1)*1e3:7.1f} us | span {span/1e3:6.1f} us, wg life med {md(life_rt)/1e3:6.1f} max {life_rt.max()/1e3:6.1f} us, "
          f"start skew med {md(start_skew)/1e3:5.1f} max {start_skew.max()/1e3:5.1f} us | cycles: prologue {md(pro):7.0f} main {md(main):8.0f} epilogue {md(epi):7.0f} | clk {clk:4.2f} GHz "
          f"| {flops/ (e0.elapsed_time(e1)*1e-3) / 1e12:6.1f} TF/s")


def lin(M, N, K, res=True):
    x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    r = torch.randn(M, N, device=dev).to(dt) if res else None
    bm = 256 if ((M + 255) // 256) * ((N + 127) // 128) >= 320 else 128
    nwg = ((M + bm - 1) // bm) * ((N + 127) // 128)
    report(f"linear M{M} N{N} K{K} bm{bm}", lambda: k.gemm(x, w, y, M, N, K, K, K, N, R=r, ldr=N), nwg, 2.0 * M * N * K)


def conv(B, H, Ci, Co):
    x = torch.randn(B * H * H, Ci, device=dev).to(dt)
    w = (torch.randn(Co, 9 * Ci, device=dev) * 0.02).to(dt)
    M = B * H * H
    y = torch.empty(M, Co, device=dev, dtype=dt)
    bias = torch.zeros(Co, device=dev)
    bm = 256 if ((M + 255) // 256) * ((Co + 127) // 128) >= 320 else 128
    nwg = ((M + bm - 1) // bm) * ((Co + 127) // 128)
    report(f"conv B{B} {H}x{H} {Ci}->{Co} bm{bm}", lambda: k.gemm(x, w, y, M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(B, H, H, Ci, H, H, 0, Ci), bias=bias),
           nwg, 2.0 * M * Co * 9 * Ci)


lin(8192, 8192, 8192, False)
lin(32768, 320, 320)
lin(32768, 320, 1280)
lin(32768, 2560, 320, False)
lin(8192, 640, 640)
lin(8192, 640, 2560)
lin(2048, 1280, 1280)
lin(2048, 1280, 5120)
lin(512, 1280, 1280)
conv(8, 64, 320, 320)
conv(8, 32, 640, 640)
conv(8, 16, 1280, 1280)
