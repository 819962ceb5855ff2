#!/usr/bin/env python3
"""Graph-timed fused AdamW (fp32 master / m / v / gradient streams + bf16 weight copy) on one optimiser share."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k
dev = torch.device("cuda:0")
n = 73 * 1024 * 1024
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3; m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
w = torch.empty(n, device=dev, dtype=torch.bfloat16)
lr = torch.tensor([1e-4], device=dev); bc = torch.tensor([0.1, 0.001], device=dev)
def run(): k.adamw(p, g, m, v, n, lr, 0.9, 0.999, 1e-8, 0.01, bc, 1.0, True, w)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e3
print(f"adamw n={n/1e6:.0f}M: {t:7.1f} us  {34 * n / t / 1e6:5.2f} TB/s of 34 B/param", flush=True)
