#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM kernel on the SD-2.1 hot shapes (HIP-event timed, random data)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k

dev = torch.device("cuda:0")
dt = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def conv(B, H, Ci, Co, mode=0):
    x = torch.randn(B * H * H, Ci, device=dev).to(dt)
    w = (torch.randn(Co, 9 * Ci, device=dev) * 0.02).to(dt)
    Ho = H // 2 if mode == 1 else (2 * H if mode == 2 else H)
    M = B * Ho * Ho
    y = torch.empty(M, Co, device=dev, dtype=dt)
    bias = torch.zeros(Co, device=dev)
    t = timeit(lambda: k.gemm(x, w, y, M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(B, H, H, Ci, Ho, Ho, mode, Ci), bias=bias))
    return f"conv{mode} B{B} {H}x{H} {Ci}->{Co}", 2.0 * M * Co * 9 * Ci, t


def lin(M, N, K, res=False):
    x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    r = torch.randn(M, N, device=dev).to(dt) if res else None
    t = timeit(lambda: k.gemm(x, w, y, M, N, K, K, K, N, R=r, ldr=N))
    return f"linear M{M} N{N} K{K}", 2.0 * M * N * K, t


def wgrad_lin(P, No, Ki, sk):
    dy = torch.randn(P, No, device=dev).to(dt)
    x = torch.randn(P, Ki, device=dev).to(dt)
    dw = torch.zeros(No, Ki, device=dev)
    t = timeit(lambda: k.gemm(dy, x, dw, No, Ki, P, No, Ki, Ki, a_mode=k.A_COLK, b_mode=k.B_COLK, out_f32=True, splitk=sk, accumulate=(sk == 1)))
    return f"wgrad-lin P{P} {No}x{Ki} sk{sk}", 2.0 * P * No * Ki, t


def wgrad_conv(B, H, Ci, Co, sk):
    dy = torch.randn(B * H * H, Co, device=dev).to(dt)
    x = torch.randn(B * H * H, Ci, device=dev).to(dt)
    dw = torch.zeros(Co, 9 * Ci, device=dev)
    P = B * H * H
    t = timeit(lambda: k.gemm(dy, x, dw, Co, 9 * Ci, P, Co, 0, 9 * Ci, a_mode=k.A_COLK, b_mode=k.B_COLK_CONV, out_f32=True, splitk=sk,
                              accumulate=(sk == 1), conv=(B, H, H, Ci, H, H, 0, Ci)))
    return f"wgrad-conv B{B} {H}x{H} {Ci}->{Co} sk{sk}", 2.0 * P * Co * 9 * Ci, t


rows = [lin(8192, 8192, 8192), lin(4096, 4096, 4096), lin(32768, 1024, 1024), lin(16384, 2048, 512), conv(8, 64, 320, 320), conv(8, 64, 960, 176), conv(8, 32, 640, 640), conv(8, 16, 1280, 1280), conv(8, 8, 1280, 1280),
        conv(8, 8, 2560, 1280), conv(8, 64, 320, 320, 1), conv(8, 32, 640, 640, 2),
        lin(32768, 320, 320, True), lin(32768, 960, 320), lin(32768, 2560, 320), lin(32768, 320, 1280, True), lin(8192, 640, 640, True),
        lin(2048, 1280, 1280, True), lin(616, 640, 1024), lin(32768, 320, 960),
        wgrad_lin(32768, 320, 320, 4), wgrad_lin(32768, 320, 320, 8), wgrad_lin(32768, 320, 320, 16), wgrad_lin(32768, 320, 320, 32), wgrad_lin(32768, 320, 320, 64),
        wgrad_lin(32768, 2560, 320, 2), wgrad_lin(32768, 2560, 320, 4), wgrad_lin(32768, 2560, 320, 8),
        wgrad_lin(8192, 640, 640, 4), wgrad_lin(8192, 640, 640, 8), wgrad_lin(8192, 640, 640, 16), wgrad_lin(2048, 1280, 1280, 1), wgrad_lin(2048, 1280, 1280, 2), wgrad_lin(2048, 1280, 1280, 4),
        wgrad_conv(8, 64, 320, 320, 4), wgrad_conv(8, 64, 320, 320, 8), wgrad_conv(8, 64, 320, 320, 16), wgrad_conv(8, 32, 640, 640, 2), wgrad_conv(8, 16, 1280, 1280, 1),
        wgrad_conv(8, 8, 1280, 1280, 1)]
for name, fl, t in rows:
    print(f"{name:42s} {t*1e6:9.1f} us  {fl/t/1e12:8.1f} TFLOP/s")
