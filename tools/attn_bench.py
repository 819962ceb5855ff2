#!/usr/bin/env python3
"""Graph-timed attention forward / backward on the step's shapes; PDMK_ATTN_NQ=1/2 forces 16 / 32 queries per wave."""
import os, sys
os.environ["PDMK_ENV_DYNAMIC"] = "1"
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

for B, H, Nq, Nk in [(8, 5, 4096, 4096), (8, 2, 4096, 4096), (16, 5, 4096, 4096), (8, 10, 1024, 1024), (8, 5, 1024, 1024), (8, 20, 256, 256), (8, 5, 4096, 77), (8, 2, 4096, 77)]:
    D = 64
    selfa = Nq == Nk
    if selfa:
        qkv = torch.randn(B, Nq, 3 * H * D, device=dev).to(dt)
        q, kk, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
        qs = ks = vs = (Nq * 3 * H * D, 3 * H * D)
    else:
        q = torch.randn(B, Nq, H * D, device=dev).to(dt)
        kv = torch.randn(B, Nk, 2 * H * D, device=dev).to(dt)
        kk, v = kv[..., :H * D], kv[..., H * D:]
        qs, ks, vs = (Nq * H * D, H * D), (Nk * 2 * H * D, 2 * H * D), (Nk * 2 * H * D, 2 * H * D)
    o = torch.zeros(B, Nq, H * D, device=dev, dtype=dt); lse = torch.zeros(B, H, Nq, device=dev)
    os_ = (Nq * H * D, H * D)
    do = torch.randn(B, Nq, H * D, device=dev).to(dt)
    dq = torch.zeros_like(q.contiguous()) if not selfa else torch.zeros(B, Nq, H * D, device=dev, dtype=dt)
    dk = torch.zeros(B, Nk, H * D, device=dev, dtype=dt); dv = torch.zeros(B, Nk, H * D, device=dev, dtype=dt)
    delta = torch.zeros(B, H, Nq, device=dev)
    fl = 4.0 * B * H * Nq * Nk * D
    row = []
    for nq in ("1", "2"):
        os.environ["PDMK_ATTN_NQ"] = nq
        tf = gtime(lambda: k.attn_fwd(q, kk, v, o, lse, B, H, Nq, Nk, qs, ks, vs, os_, D ** -0.5))
        tb = gtime(lambda: k.attn_bwd(q, kk, v, o, do, lse, delta, dq, dk, dv, B, H, Nq, Nk, qs, ks, vs, os_, os_, (Nk * H * D, H * D), (Nk * H * D, H * D), D ** -0.5))
        row.append(f"NQ{nq}: fwd {tf:7.1f} us {fl / tf / 1e6:6.1f} TF/s | bwd {tb:7.1f} us {2.5 * fl / tb / 1e6:6.1f} TF/s")
    os.environ.pop("PDMK_ATTN_NQ")
    for a_, b_ in (("1", "2"), ("2", "1"), ("0", "0")):         # dQ / dK,dV forms mixed; 0 = the library's own choice
        os.environ["PDMK_ATTN_NQ_DQ"], os.environ["PDMK_ATTN_NQ_DKV"] = a_, b_
        tb = gtime(lambda: k.attn_bwd(q, kk, v, o, do, lse, delta, dq, dk, dv, B, H, Nq, Nk, qs, ks, vs, os_, os_, (Nk * H * D, H * D), (Nk * H * D, H * D), D ** -0.5))
        row.append(f"dq{a_}/dkv{b_}: bwd {tb:7.1f} us")
    tf = gtime(lambda: k.attn_fwd(q, kk, v, o, lse, B, H, Nq, Nk, qs, ks, vs, os_, D ** -0.5))
    row.append(f"default fwd {tf:7.1f} us")
    print(f"B{B} H{H} Nq{Nq} Nk{Nk}: " + "   ".join(row))
