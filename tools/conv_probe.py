#!/usr/bin/env python3
"""Runs one GEMM shape repeatedly (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k
dev = torch.device("cuda:0"); dt = torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
B, H, Ci, Co = 8, 64, 320, 320
if which == "conv":
    x = torch.randn(B * H * H, Ci, device=dev).to(dt); w = (torch.randn(Co, 9 * Ci, device=dev) * 0.02).to(dt)
    y = torch.empty(B * H * H, Co, device=dev, dtype=dt)
    f = lambda: k.gemm(x, w, y, B * H * H, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(B, H, H, Ci, H, H, 0, Ci))
elif which == "lin":
    M, N, K = 32768, 2560, 320
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.02).to(dt); y = torch.empty(M, N, device=dev, dtype=dt)
    f = lambda: k.gemm(x, w, y, M, N, K, K, K, N)
else:
    M, N, K = 8192, 8192, 8192
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.02).to(dt); y = torch.empty(M, N, device=dev, dtype=dt)
    f = lambda: k.gemm(x, w, y, M, N, K, K, K, N)
for _ in range(10):
    f()
torch.cuda.synchronize()
