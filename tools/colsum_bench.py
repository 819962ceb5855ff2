#!/usr/bin/env python3
"""Graph-timed per-image column sums (the time-embedding gradient of a ResBlock) against a plain read of the tensor."""
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
for B, HW, C in [(8, 4096, 320), (8, 1024, 640), (8, 256, 1280), (8, 64, 1280)]:
    x = torch.randn(B * HW, C, device=dev).to(dt)
    out = torch.zeros(B, C, device=dev)
    t = gtime(lambda: k.colsum(x, out, HW, C, C, accumulate=True, nbatch=B, ldo=C))
    ref = x.float().view(B, HW, C).sum(1)
    out.zero_(); k.colsum(x, out, HW, C, C, accumulate=True, nbatch=B, ldo=C); torch.cuda.synchronize()
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"B{B} HW{HW} C{C}: colsum {t:6.1f} us ({B * HW * C * 2 / t / 1e6:5.2f} TB/s)  err {err:.1e}", flush=True)
