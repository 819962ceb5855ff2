import os, sys
os.environ["PDMK_ENV_DYNAMIC"] = "1"
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k
dev, dt = torch.device("cuda:0"), torch.bfloat16
def gtime(fns, reps=6):
    for f in fns: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns: f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3
for (B, H, Ci, Co) in [(8, 64, 320, 320), (8, 64, 640, 320), (8, 64, 960, 320), (8, 32, 640, 640), (8, 32, 1280, 640), (8, 32, 320, 640), (8, 64, 224, 192)]:
    M = B * H * H
    NB = 4
    xs = [torch.randn(M, Ci, device=dev).to(dt) for _ in range(NB)]
    ws = [(torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5).to(dt) for _ in range(NB)]
    ys = [torch.zeros(M, Co, device=dev, dtype=dt) for _ in range(NB)]
    bias = torch.randn(Co, device=dev)
    row = []
    for c in (15, 16, 13):
        os.environ["PDMK_RING_CFG"] = str(c)
        fns = [(lambda i=i: k.gemm(xs[i], ws[i], ys[i], M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(B, H, H, Ci, H, H, 0, Ci), bias=bias)) for i in range(NB)]
        t = gtime(fns)
        row.append(f"c{c} {t:6.1f} us {2.0 * M * Co * 9 * Ci / t / 1e6:6.0f} TF/s")
    print(f"conv B{B} {H}x{H} {Ci}->{Co}: " + " | ".join(row), flush=True)
