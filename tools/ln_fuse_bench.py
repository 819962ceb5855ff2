#!/usr/bin/env python3
"""LayerNorm + Linear as two launches (pdmk_layernorm_fwd, then the tuned plan's GEMM) against the LayerNorm-prologue launch of the
row-block kernel (pdmk_gemm_args.ln_gamma), on the transformer-block shapes of the 64x64 / 32x32 latent levels at B = 8; inference
form (teacher: nothing but the product leaves) and training form (student: + mean / rstd + the normalised rows).  Operands rotate
over NBUF buffers; launches are graph-replayed."""
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


shapes = [(32768, 960, 320, "plain"), (32768, 320, 320, "plain"), (32768, 2560, 320, "geglu"), (32768, 1408, 320, "geglu"),
          (32768, 576, 320, "plain"), (32768, 192, 320, "plain"),
          (8192, 1920, 640, "plain"), (8192, 640, 640, "plain"), (8192, 2816, 640, "geglu")]
print(f"{'M N K mode':30s} {'form':9s} LN us + GEMM us = two launches | fused us | saved")
for M, N, K, mode in shapes:
    geglu = mode == "geglu"
    xs = [torch.randn(M, K, device=dev).to(dt) for _ in range(NBUF)]
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(dt) for _ in range(NBUF)]
    gamma, beta = torch.randn(K, device=dev), torch.randn(K, device=dev)
    bias = torch.randn(N, device=dev) if geglu else None
    ls = [torch.zeros(M, K, device=dev, dtype=dt) for _ in range(NBUF)]
    sts = [torch.zeros(M, 2, device=dev) for _ in range(NBUF)]
    ys = [torch.zeros(M, N // 2 if geglu else N, device=dev, dtype=dt) for _ in range(NBUF)]
    pres = [torch.zeros(M, N, device=dev, dtype=dt) for _ in range(NBUF)] if geglu else [None] * NBUF
    for train in (False, True):
        def ln(i):
            return lambda: k.layernorm_fwd(xs[i], ls[i], gamma, beta, sts[i], M, K, K, K, 1e-5)

        def mm(i):
            if geglu:
                return lambda: k.gemm_geglu(ls[i], ws[i], ys[i], pres[i] if train else None, M, N, K, K, K, bias=bias)
            return lambda: k.gemm(ls[i], ws[i], ys[i], M, N, K, K, K, N)

        def fused(i):
            a = (gamma, beta, sts[i] if train else None, ls[i] if train else None, 1e-5)
            if geglu:
                return lambda: k.gemm_geglu(xs[i], ws[i], ys[i], pres[i] if train else None, M, N, K, K, K, bias=bias, ln=a)
            return lambda: k.gemm(xs[i], ws[i], ys[i], M, N, K, K, K, N, ln=a)
        if not k.gemm_ln_supported(xs[0], ws[0], M, N, K, K, K, geglu=geglu, bias=geglu):
            print(f"{M:6d}{N:6d}{K:5d} {mode:8s}   not supported")
            break
        t_ln = gtime([ln(i) for i in range(NBUF)])
        t_mm = gtime([mm(i) for i in range(NBUF)])
        t_f = gtime([fused(i) for i in range(NBUF)])
        print(f"{M:6d}{N:6d}{K:5d} {mode:8s}   {'train' if train else 'inference':9s} {t_ln:6.1f} + {t_mm:6.1f} = {t_ln + t_mm:6.1f} | {t_f:6.1f} | {t_ln + t_mm - t_f:+6.1f}")
