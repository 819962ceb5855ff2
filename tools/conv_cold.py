#!/usr/bin/env python3
"""Are the low-resolution halo convs (8x8 .. 32x32 latents, split over channel blocks) bound by the latency of their weight stream?
Each forced halo candidate with ONE operand set (weights L2 / Infinity-Cache-warm) and rotating over > 256 MiB of operand sets (cold),
slab split-K as the engine runs them (the finish pass is not timed)."""
import os, sys
os.environ["PDMK_ENV_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k
dev, dt = torch.device("cuda:0"), torch.bfloat16


def gtime(fns, reps=5):
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


for (B, H, Ci, Co, sk) in [(8, 8, 1280, 1280, 6), (8, 8, 2560, 1280, 8), (8, 16, 1280, 1280, 2), (8, 16, 640, 1280, 3), (8, 16, 2560, 1280, 4),
                           (8, 32, 640, 640, 1), (8, 32, 1280, 640, 1), (8, 64, 320, 320, 1)]:
    M = B * H * H
    per = 2 * (M * Ci + Co * 9 * Ci) + 4 * sk * M * Co
    NBUF = max(3, int(300e6 // per) + 1)
    xs = [torch.randn(M, Ci, device=dev).to(dt) for _ in range(NBUF)]
    ws = [(torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5).to(dt) for _ in range(NBUF)]
    ys = [torch.zeros(sk * M * Co if sk > 1 else M * Co, device=dev, dtype=torch.float32 if sk > 1 else dt) for _ in range(NBUF)]

    def call(i):
        if sk > 1:
            return lambda: k.gemm(xs[i], ws[i], ys[i], M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(B, H, H, Ci, H, H, 0, Ci), out_f32=True, splitk=sk, accumulate=2)
        return lambda: k.gemm(xs[i], ws[i], ys[i].view(M, Co), M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(B, H, H, Ci, H, H, 0, Ci))
    row = []
    for c in (15, 16, 13):
        os.environ["PDMK_RING_CFG"] = str(c)
        try:
            call(0)(); torch.cuda.synchronize()
            if k.last_candidate() != c:
                continue
            warm, cold = gtime([call(0)] * 4), gtime([call(i) for i in range(NBUF)], reps=3)
            row.append(f"c{c} warm {warm:6.1f} cold {cold:6.1f} us ({2.0 * M * Co * 9 * Ci / cold / 1e6:5.0f} TF/s)")
        except Exception as e:
            row.append(f"c{c} n/a")
    os.environ.pop("PDMK_RING_CFG", None)
    print(f"conv B{B} {H}x{H} {Ci}->{Co} sk{sk} NBUF={NBUF}: " + " | ".join(row), flush=True)
