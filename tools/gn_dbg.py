#!/usr/bin/env python3
"""Apply-only GroupNorm timing (statistics from the GEMM epilogue) against a plain copy of the same tensor."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k
dev = torch.device("cuda:0"); dt = torch.bfloat16
REP = 10
def gtime(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REP) * 1e3
for B, HW, C in [(8, 4096, 320), (8, 4096, 960), (8, 1024, 640), (8, 1024, 1280), (8, 256, 1280), (8, 64, 1280)]:
    G, gs = 32, C // 32
    x = torch.randn(B * HW, C, device=dev).to(dt); y = torch.empty_like(x)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    stats = torch.zeros(B, G, 2, device=dev)
    acc = torch.zeros(B, 4, C, device=dev, dtype=torch.int64)
    ta = gtime(lambda: k.groupnorm_apply_colstat(x, y, gamma, beta, stats, acc, 0, B, HW, C, C, C, G, gs, 1e-5, True))
    print(f"B{B} HW{HW} C{C}: apply {ta:6.1f} us   copy {gtime(lambda: y.copy_(x)):5.1f}", flush=True)
