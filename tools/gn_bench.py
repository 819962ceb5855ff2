#!/usr/bin/env python3
"""Graph-timed GroupNorm(+SiLU) forward / backward on the step's shapes (GB/s = algorithmic bytes / time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
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
for B, HW, C in [(8, 4096, 320), (8, 4096, 640), (8, 4096, 960), (8, 1024, 640), (8, 1024, 1280), (8, 256, 1280), (8, 256, 2560), (8, 64, 1280)]:
    G, gs = 32, C // 32
    x = torch.randn(B * HW, C, device=dev).to(dt); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    stats = torch.zeros(B, G, 2, device=dev); ws = torch.zeros(B * G * 64, device=dev, dtype=torch.float64)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    tf = gtime(lambda: k.groupnorm_fwd(x, y, gamma, beta, stats, ws, B, HW, C, C, C, G, gs, 1e-5, True))
    tb = gtime(lambda: k.groupnorm_bwd(x, dy, dx, gamma, beta, stats, dg, db, ws, B, HW, C, C, C, C, G, gs, True, False))
    nb = B * HW * C * 2
    acc = torch.zeros(B, 4, C, device=dev, dtype=torch.int64)       # (the values do not matter for the timing)
    ta = gtime(lambda: k.groupnorm_apply_colstat(x, y, gamma, beta, stats, acc, 0, B, HW, C, C, C, G, gs, 1e-5, True))
    tc = gtime(lambda: y.copy_(x))
    print(f"B{B} HW{HW} C{C}: fwd {tf:6.1f} us ({3 * nb / tf / 1e6:5.2f} TB/s of 2R+1W)   bwd {tb:6.1f} us ({5 * nb / tb / 1e6:5.2f} TB/s of 4R+1W)   apply-only {ta:6.1f} us ({2 * nb / ta / 1e6:5.2f} TB/s of 1R+1W; torch copy {tc:6.1f} us)")
