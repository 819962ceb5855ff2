import os, sys
os.environ["PDMK_ENV_DYNAMIC"] = "1"
sys.path.insert(0, "/root/repo/unlearn-ft_amd")
import torch
from pdm import _pdmk as k
dev = torch.device("cuda:0"); dt = torch.bfloat16
def gtime(fn, REP=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * REP) * 1e3
names = ["old", "256x128", "256x160", "128x128", "128x160", "64x128", "64x160", "s128x128", "s64x128", "s64x160", "128x192", "64x192", "s128x160"]
for M, N, K in [(8192, 8192, 8192), (4096, 4096, 4096), (32768, 1280, 1280), (32768, 2560, 2560)]:
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt); y = torch.empty(M, N, device=dev, dtype=dt)
    row = []
    for i in (0, 1, 2, 3, 4, 10, 12):
        os.environ["PDMK_RING_CFG"] = str(i)
        t = gtime(lambda: k.gemm(x, w, y, M, N, K, K, K, N))
        row.append(f"{names[i]} {2.0*M*N*K/t/1e6:6.0f}")
    print(f"M{M} N{N} K{K}: " + "  ".join(row) + "  TF/s")
