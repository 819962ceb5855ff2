#!/usr/bin/env python3
"""A/B of the weight-gradient GEMM candidates (0 = register-staged K-step-64 kernel, 1/2 = 128x128 deep/shallow LDS-DMA
ring, 3 = 64x128, 4 = 128x64, 5 = 64x64) over split-K factors on the hot wgrad shapes; graph-replayed launches."""
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

SLAB = "--slabs" in sys.argv      # splits store partial slabs (accumulate = 2) instead of adding with atomics; the finish is not timed


def lin(P, No, Ki):
    dy = torch.randn(P, No, device=dev).to(dt); x = torch.randn(P, Ki, device=dev).to(dt)
    dw = torch.zeros(No, Ki, device=dev); db = torch.zeros(No, device=dev)
    ws = torch.zeros(64 * No * Ki, device=dev) if SLAB else None
    def run(sk):
        if SLAB and sk > 1:
            return k.gemm(dy, x, ws, No, Ki, P, No, Ki, Ki, a_mode=k.A_COLK, b_mode=k.B_COLK, out_f32=True, splitk=sk, accumulate=2, colsum_out=db)
        return k.gemm(dy, x, dw, No, Ki, P, No, Ki, Ki, a_mode=k.A_COLK, b_mode=k.B_COLK, out_f32=True, splitk=sk, accumulate=(sk == 1), colsum_out=db)
    return f"lin P{P} {No}x{Ki}", 2.0 * P * No * Ki, run

def conv(B, H, Ci, Co):
    P = B * H * H
    dy = torch.randn(P, Co, device=dev).to(dt); x = torch.randn(P, Ci, device=dev).to(dt)
    dw = torch.zeros(Co, 9 * Ci, device=dev); db = torch.zeros(Co, device=dev)
    return f"conv B{B} {H}x{H} {Ci}->{Co}", 2.0 * P * Co * 9 * Ci, lambda sk: k.gemm(dy, x, dw, Co, 9 * Ci, P, Co, 0, 9 * Ci, a_mode=k.A_COLK, b_mode=k.B_COLK_CONV, out_f32=True, splitk=sk, accumulate=(sk == 1), conv=(B, H, H, Ci, H, H, 0, Ci), colsum_out=db)

LIN_ONLY = SLAB
shapes = [lin(32768, 320, 320), lin(32768, 960, 320), lin(32768, 2560, 320), lin(32768, 320, 1280), lin(8192, 640, 640), lin(8192, 5120, 640), lin(2048, 1280, 1280), lin(2048, 1280, 5120),
          conv(8, 64, 320, 320), conv(8, 32, 640, 640), conv(8, 16, 1280, 1280), conv(8, 8, 1280, 1280), conv(8, 32, 352, 608)]
sks = [1, 2, 4, 8, 16, 32, 64]
for name, fl, fn in (shapes[:8] if LIN_ONLY else shapes):
    print(name)
    for cand in (0, 1, 2, 3, 4, 5):
        os.environ["PDMK_WGRAD_CFG"] = str(cand)
        row = []
        for sk in sks:
            try:
                row.append(gtime(lambda: fn(sk)))
            except Exception as e:
                row.append(float("nan"))
        best = min(r for r in row if r == r)
        print(f"   cand {cand}: " + " ".join(f"sk{s}={t:7.1f}" for s, t in zip(sks, row)) + f"   best {best:7.1f} us = {fl / best / 1e6:6.1f} TF/s")
