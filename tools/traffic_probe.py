#!/usr/bin/env python3
"""Isolated HBM-traffic probe of the Linear forward kernels (is the A operand re-read once per n-tile?).
Launches ONE forced candidate on the dominant shapes, operands rotating over 6 buffers (not cache-resident from the launch
before, as in the step), `REP` launches per variant, each variant under its own kernel symbol count so that the PMC CSV can be
split by launch order.  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; tools/traffic_probe_sum.py
prints bytes per launch next to the algorithmic bytes (gfx950: read bytes = 2 x FETCH_SIZE).
usage: traffic_probe.py <cand> [<cand> ...]"""
import os, sys
os.environ["PDMK_ENV_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k
dev, dt, NBUF, REP = torch.device("cuda:0"), torch.bfloat16, 6, 12
# (M, N, K, mode)
SHAPES = [(32768, 320, 320, "plain"), (32768, 320, 320, "bias+res"), (32768, 320, 320, "acc"), (32768, 960, 320, "plain"),
          (32768, 160, 320, "plain"), (8192, 640, 640, "plain"), (8192, 640, 640, "bias+res")]
CONVS = [(8, 64, 320, 320), (8, 64, 640, 320), (8, 32, 640, 640), (8, 64, 960, 320)]      # (B, H, Ci, Co), stride-1 3x3
conv_cands = [int(c[1:]) for c in sys.argv[1:] if c.startswith("c")]
cands = [int(c) for c in sys.argv[1:] if not c.startswith("c")] or ([] if conv_cands else [12])
print("order of launches (REP each):")
for c in cands:
    os.environ["PDMK_RING_CFG"] = str(c)
    for M, N, K, mode in SHAPES:
        xs = [torch.randn(M, K, device=dev).to(dt) for _ in range(NBUF)]
        ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(dt) for _ in range(NBUF)]
        ys = [torch.zeros(M, N, device=dev, dtype=dt) for _ in range(NBUF)]
        rs = [torch.randn(M, N, device=dev).to(dt) for _ in range(NBUF)] if "res" in mode else [None] * NBUF
        bias = torch.randn(N, device=dev) if "bias" in mode else None
        torch.cuda.synchronize()
        for r in range(REP):
            i = r % NBUF
            k.gemm(xs[i], ws[i], ys[i], M, N, K, K, K, N, R=rs[i], ldr=N if rs[i] is not None else 0, bias=bias, accumulate=(mode == "acc"))
        torch.cuda.synchronize()
        name = k.candidate_name(k.A_ROWK, k.B_ROWK, k.last_candidate())
        alg_r = 2 * (M * K + N * K + (M * N if mode in ("bias+res", "acc") else 0))
        alg_w = 2 * M * N
        print(f"VARIANT cand={c} M={M} N={N} K={K} mode={mode} reps={REP} alg_read={alg_r} alg_write={alg_w} kernel={name}", flush=True)

for c in conv_cands:
    os.environ["PDMK_RING_CFG"] = str(c)
    for B, H, Ci, Co in CONVS:
        M = B * H * H
        xs = [torch.randn(M, Ci, device=dev).to(dt) for _ in range(NBUF)]
        ws = [(torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5).to(dt) for _ in range(NBUF)]
        ys = [torch.zeros(M, Co, device=dev, dtype=dt) for _ in range(NBUF)]
        bias = torch.randn(Co, device=dev)
        torch.cuda.synchronize()
        for r in range(REP):
            i = r % NBUF
            k.gemm(xs[i], ws[i], ys[i], M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(B, H, H, Ci, H, H, 0, Ci), bias=bias)
        torch.cuda.synchronize()
        name = k.candidate_name(k.A_CONV, k.B_ROWK, k.last_candidate())
        print(f"VARIANT cand={c} M={M} N={Co} K={9 * Ci} mode=conv{H}x{H} reps={REP} alg_read={2 * (M * Ci + Co * 9 * Ci)} alg_write={2 * M * Co} kernel={name}", flush=True)
