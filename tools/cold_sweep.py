#!/usr/bin/env python3
"""Do the plan cache's timings (same operands three times in a row: L2 / MALL-warm) rank the Linear candidates the way the step
sees them (every operand cold: written by another kernel tens of MB ago)?  Every ring / row-block candidate on the C x C
projection shapes that carry the lowest MFMA utilisation of the step, timed WARM (one buffer set) and COLD (rotating over NBUF
sets, > 256 MiB in total so that the Infinity Cache cannot hold them), graph-replayed."""
import os, sys
os.environ["PDMK_ENV_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k
dev, dt = torch.device("cuda:0"), torch.bfloat16
CANDS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 17, 18, 19, 20, 21]


def gtime(fns, reps=6):
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
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3


shapes = [(2048, 1280, 1280, "plain"), (2048, 1280, 1280, "bias+res"), (8192, 640, 640, "plain"), (8192, 640, 640, "bias+res"),
          (32768, 320, 320, "plain"), (32768, 320, 320, "bias+res"), (2048, 640, 1280, "plain"), (8192, 320, 640, "plain")]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if str(s[0]) in sys.argv[1:]]
for M, N, K, mode in shapes:
    per_set = 2 * (M * K + N * K + 2 * M * N)
    NBUF = max(6, int(300e6 // per_set) + 1)
    xs = [torch.randn(M, K, device=dev).to(dt) for _ in range(NBUF)]
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(dt) for _ in range(NBUF)]
    ys = [torch.zeros(M, N, device=dev, dtype=dt) for _ in range(NBUF)]
    rs = [torch.randn(M, N, device=dev).to(dt) for _ in range(NBUF)] if "res" in mode else [None] * NBUF
    bias = torch.randn(N, device=dev) if "bias" in mode else None

    def call(i):
        return lambda: k.gemm(xs[i], ws[i], ys[i], M, N, K, K, K, N, R=rs[i], ldr=N if rs[i] is not None else 0, bias=bias)
    os.environ.pop("PDMK_RING_CFG", None)
    call(0)()
    torch.cuda.synchronize()
    tuned = k.last_candidate()
    row = []
    for c in CANDS:
        os.environ["PDMK_RING_CFG"] = str(c)
        try:
            call(0)()
            torch.cuda.synchronize()
            if k.last_candidate() != c:
                continue
            warm = gtime([call(0)] * 4)
            cold = gtime([call(i) for i in range(NBUF)], reps=3)
            row.append((c, warm, cold))
        except Exception as e:
            pass
    os.environ.pop("PDMK_RING_CFG", None)
    bw, bc = min(row, key=lambda r: r[1]), min(row, key=lambda r: r[2])
    tc = [r for r in row if r[0] == tuned]
    print(f"{M}x{N}x{K} {mode:9s} NBUF={NBUF} tuned=c{tuned} ({tc[0][1]:.1f} warm / {tc[0][2]:.1f} cold us)  best warm c{bw[0]} {bw[1]:.1f}  best cold c{bc[0]} {bc[2]:.1f}")
    print("    " + "  ".join(f"c{c}:{w:.1f}/{cd:.1f}" for c, w, cd in row), flush=True)
