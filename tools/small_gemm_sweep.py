#!/usr/bin/env python3
"""Sweep of the ring candidates (incl. the small tiles 17..19) x split-K on the small / short linear GEMMs of the step.
Operands rotate over NBUF distinct buffers so that weights are not L2-resident from the previous launch (as in the step)."""
import os, sys
os.environ["PDMK_ENV_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k

dev, dt, NBUF = torch.device("cuda:0"), torch.bfloat16, 6


def gtime(fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (8 * len(fns)) * 1e3


shapes = [(2048, 1280, 640), (2048, 640, 1280), (8192, 320, 640), (8192, 640, 320), (2048, 1280, 1280), (8192, 640, 640),
          (512, 1280, 1280), (512, 1280, 2560), (616, 1280, 1024), (32768, 320, 128), (32768, 128, 320), (32768, 320, 320),
          (2048, 1280, 2560), (2048, 1280, 5120)]
cands = [0, 3, 4, 6, 7, 8, 9, 12, 17, 18, 19]
print(f"{'M N K':22s} " + " ".join(f"c{c:<2d}sk1" for c in cands) + " | best split (cand, sk, us)")
for M, N, K in shapes:
    xs = [torch.randn(M, K, device=dev).to(dt) for _ in range(NBUF)]
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(dt) for _ in range(NBUF)]
    ys = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(NBUF)]
    rs = [torch.randn(M, N, device=dev).to(dt) for _ in range(NBUF)]
    bias = torch.randn(N, device=dev)
    row = []
    for c in cands:
        os.environ["PDMK_RING_CFG"] = str(c)
        fns = [(lambda i=i: k.gemm(xs[i], ws[i], ys[i], M, N, K, K, K, N, R=rs[i], ldr=N, bias=bias)) for i in range(NBUF)]
        row.append(gtime(fns))
    ref = xs[0].float() @ ws[0].float().t() + bias + rs[0].float()
    err = (ys[0].float() - ref).abs().max().item() / ref.abs().max().item()
    best = (None, 1, 1e9)
    nk = K // 64
    for sk in (2, 3, 4, 6, 8):
        if nk // sk < 2:
            continue
        wsl = [torch.empty(sk, M, N, device=dev) for _ in range(NBUF)]
        for c in (7, 8, 9, 12, 17, 18, 19):
            os.environ["PDMK_RING_CFG"] = str(c)

            def mk(i):
                def f():
                    k.gemm(xs[i], ws[i], wsl[i], M, N, K, K, K, N, out_f32=True, splitk=sk, accumulate=2)
                    k.splitk_finish(wsl[i], ys[i], M, N, N, sk, bias=bias, R=rs[i], ldr=N)
                return f
            t = gtime([mk(i) for i in range(NBUF)])
            if t < best[2]:
                best = (c, sk, t)
        del wsl
    fl = 2.0 * M * N * K
    b1 = min(row)
    print(f"{M:6d}{N:6d}{K:6d}     " + " ".join(f"{t:7.1f}" for t in row) + f" | {best[0]} sk{best[1]} {best[2]:.1f}us  "
          f"(best sk1 c{cands[row.index(b1)]} {b1:.1f}us = {fl / b1 / 1e6:.0f} TF/s; err {err:.1e})")
os.environ.pop("PDMK_RING_CFG", None)
