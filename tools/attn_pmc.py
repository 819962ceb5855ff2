#!/usr/bin/env python3
"""A few eager attention forward / backward launches on the step's largest shape, for a rocprofv3 --pmc pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k
dev = torch.device("cuda:0"); dt = torch.bfloat16
B, H, N, D = 8, 5, 4096, 64
qkv = torch.randn(B, N, 3 * H * D, device=dev).to(dt)
q, kk, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
qs = (N * 3 * H * D, 3 * H * D)
o = torch.zeros(B, N, H * D, device=dev, dtype=dt); lse = torch.zeros(B, H, N, device=dev)
os_ = (N * H * D, H * D)
do = torch.randn(B, N, H * D, device=dev).to(dt)
dq = torch.zeros(B, N, H * D, device=dev, dtype=dt); dk = torch.zeros_like(dq); dv = torch.zeros_like(dq)
delta = torch.zeros(B, H, N, device=dev)
for _ in range(4):
    k.attn_fwd(q, kk, v, o, lse, B, H, N, N, qs, qs, qs, os_, D ** -0.5)
    k.attn_bwd(q, kk, v, o, do, lse, delta, dq, dk, dv, B, H, N, N, qs, qs, qs, os_, os_, os_, os_, D ** -0.5)
torch.cuda.synchronize()
