#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k
dev = torch.device("cuda:0"); dt = torch.bfloat16

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, N, K in [(8, 1280, 1280), (8, 640, 1280), (8, 320, 1280), (8, 1280, 320), (8, 1280, 640)]:
    x = torch.randn(M, K, device=dev).to(dt); w = torch.randn(N, K, device=dev).to(dt)
    y = torch.empty(M, N, device=dev); bias = torch.zeros(N, device=dev)
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev); dy = torch.randn(M, N, device=dev)
    t0 = timeit(lambda: k.gemm(x, w, y, M, N, K, K, K, N, bias=bias, out_f32=True))
    t1 = timeit(lambda: k.skinny_gemm(x, w, y, M, N, K, K, K, N, bias=bias))
    dyb = dy.to(dt)
    t2 = timeit(lambda: k.gemm(dyb, x, dw, N, K, M, N, K, K, a_mode=k.A_COLK, b_mode=k.B_COLK, out_f32=True, accumulate=True, colsum_out=db))
    t3 = timeit(lambda: k.skinny_wgrad(dy, x, dw, db, M, N, K, N, K, K))
    print(f"M{M} N{N} K{K}: gemm {t0:6.1f} us  skinny {t1:6.1f} us | wgrad gemm {t2:6.1f} us  skinny {t3:6.1f} us")
