#!/usr/bin/env python3
"""Diagnostic (needs gemm_rowblock.o built with -DPDMK_RB_STAMPS): per-workgroup phase times of rowblock_kernel (wave 0).
Stamps (shader clock): 0 entry, 1 A + first weight stages issued, 2 A landed, 3 A fragments in registers, 4+s wait+barrier of
weight stage s passed (s < 16), 20 first epilogue done, 21 exit; 22 / 23 entry / exit in 100 MHz ticks."""
import ctypes, os, sys
os.environ["PDMK_ENV_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import numpy as np
import torch
from pdm import _pdmk as k

dev, dt = torch.device("cuda:0"), torch.bfloat16


def stamps(nwg):
    buf = (ctypes.c_ulonglong * (nwg * 24))()
    assert k._lib.pdmk_debug_rb_read_stamps(buf, nwg * 24) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 24).astype(np.int64)


def run(M, N, K, cand, res=False):
    os.environ["PDMK_RING_CFG"] = str(cand)
    xs = [torch.randn(M, K, device=dev).to(dt) for _ in range(4)]
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(dt) for _ in range(4)]
    ys = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(4)]
    r = torch.randn(M, N, device=dev).to(dt) if res else None
    for i in range(4):
        k.gemm(xs[i], ws[i], ys[i], M, N, K, K, K, N, R=r, ldr=N if res else 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); k.gemm(xs[0], ws[0], ys[0], M, N, K, K, K, N, R=r, ldr=N if res else 0); e1.record(); torch.cuda.synchronize()
    name = k.candidate_name(k.A_ROWK, k.B_ROWK, k.last_candidate())
    bm = 128 if cand == 20 else 64
    nwg = min(1024, (M + bm - 1) // bm * max(1, int(os.environ.get("PDMK_RB_GRP", "1"))))
    s = stamps(nwg)
    life = (s[:, 23] - s[:, 22]) * 10.0
    clk = np.median((s[:, 21] - s[:, 0]) / np.maximum(life, 1))
    md = lambda a: float(np.median(a))
    print(f"{name} M{M} N{N} K{K} res={res}: event {e0.elapsed_time(e1)*1e3:.1f} us, wg life med {md(life)/1e3:.2f} max {life.max()/1e3:.2f} us, "
          f"span {(s[:,23].max()-s[:,22].min())*10/1e3:.2f} us, clk {clk:.2f} GHz")
    print("   cycles: A landed", md(s[:, 2] - s[:, 0]), "A->regs", md(s[:, 3] - s[:, 2]), "| first step", md(s[:, 4] - s[:, 3]))
    steps = [md(s[:, 5 + i] - s[:, 4 + i]) for i in range(15)]
    print("   step-to-step:", " ".join(f"{x:.0f}" for x in steps))
    print("   exit at", md(s[:, 21] - s[:, 0]))


run(32768, 960, 320, 20)
run(32768, 320, 320, 20)
run(8192, 1920, 640, 21)
