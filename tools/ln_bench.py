#!/usr/bin/env python3
"""Graph-timed LayerNorm forward / backward on the step's shapes against a plain copy of the same tensor."""
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
for M, C in [(32768, 320), (8192, 640), (2048, 1280), (65536, 320), (512, 1280)]:
    x = torch.randn(M, C, device=dev).to(dt); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    stats = torch.zeros(M, 2, device=dev)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    tf = gtime(lambda: k.layernorm_fwd(x, y, gamma, beta, stats, M, C, C, C, 1e-5))
    tb = gtime(lambda: k.layernorm_bwd(x, dy, dx, gamma, stats, dg, db, M, C, C, C, C, False))
    nb = M * C * 2
    print(f"M{M} C{C}: fwd {tf:6.1f} us ({2 * nb / tf / 1e6:5.2f} TB/s of 1R+1W)   bwd {tb:6.1f} us ({3 * nb / tb / 1e6:5.2f} TB/s of 2R+1W)   copy {gtime(lambda: y.copy_(x)):5.1f}", flush=True)
