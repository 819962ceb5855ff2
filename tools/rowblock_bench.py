#!/usr/bin/env python3
"""Row-block Linear kernel (candidates 20 / 21, gemm_rowblock.hip) against every ring candidate on the Linear shapes of the
64x64 / 32x32 latent levels (B = 8: M = 32768 / 8192): forward shapes with bias / residual / GEGLU epilogue, dgrad shapes.
Operands rotate over NBUF buffers (not L2-resident from the previous launch, as in the step); launches are graph-replayed.
PDMK_RB_GRP=<n> forces the number of column groups."""
import os, sys
os.environ["PDMK_ENV_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
from pdm import _pdmk as k

dev, dt, NBUF = torch.device("cuda:0"), torch.bfloat16, 6
RING = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 17, 18, 19]
RB = [20, 21]


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


# (M, N, K, mode): mode in plain / bias / bias+res / geglu / geglu+keep / acc
shapes = [(32768, 320, 320, "plain"), (32768, 320, 320, "bias+res"), (32768, 960, 320, "plain"), (32768, 2560, 320, "geglu"),
          (32768, 1408, 320, "geglu+keep"), (32768, 704, 320, "plain"), (32768, 320, 320, "acc"), (32768, 640, 320, "plain"),
          (8192, 640, 640, "plain"), (8192, 640, 640, "bias+res"), (8192, 1920, 640, "plain"), (8192, 5120, 640, "geglu"),
          (8192, 2816, 640, "geglu+keep"), (8192, 1408, 640, "plain"), (8192, 640, 320, "plain"), (2048, 1280, 640, "plain")]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if str(s[0]) in sys.argv[1:] or s[3] in sys.argv[1:]]
print(f"{'M N K mode':34s} best ring (cand us TF/s) | rowblock c20 c21 (us) | speed-up")
for M, N, K, mode in shapes:
    xs = [torch.randn(M, K, device=dev).to(dt) for _ in range(NBUF)]
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(dt) for _ in range(NBUF)]
    geglu = mode.startswith("geglu")
    ys = [torch.zeros(M, N // 2 if geglu else N, device=dev, dtype=dt) for _ in range(NBUF)]
    pre = [torch.zeros(M, N, device=dev, dtype=dt) for _ in range(NBUF)] if mode == "geglu+keep" else [None] * NBUF
    rs = [torch.randn(M, N, device=dev).to(dt) for _ in range(NBUF)] if "res" in mode else [None] * NBUF
    bias = torch.randn(N, device=dev) if ("bias" in mode or geglu) else None

    def call(i):
        if geglu:
            return lambda: k.gemm_geglu(xs[i], ws[i], ys[i], pre[i], M, N, K, K, K, bias=bias)
        return lambda: k.gemm(xs[i], ws[i], ys[i], M, N, K, K, K, N, R=rs[i], ldr=N if rs[i] is not None else 0, bias=bias,
                              accumulate=(mode == "acc"))
    res = {}
    for c in RING + RB:
        os.environ["PDMK_RING_CFG"] = str(c)
        try:
            fns = [call(i) for i in range(NBUF)]
            fns[0]()
            name = k.candidate_name(k.A_ROWK, k.B_ROWK, k.last_candidate())
            if c in RB and not name.startswith("pdmk_rb::"):
                continue
            if c in RING and not name.startswith("pdmk_ring::"):
                continue
            res[c] = gtime(fns)
        except Exception as e:
            print("   cand", c, "failed:", e)
    ring_best = min((res[c], c) for c in RING if c in res)
    fl = 2.0 * M * N * K
    rb = [res.get(c, float("nan")) for c in RB]
    rb_best = min([t for t in rb if t == t], default=float("nan"))
    print(f"{M:6d}{N:6d}{K:5d} {mode:12s}   c{ring_best[1]:<2d} {ring_best[0]:7.1f} us {fl / ring_best[0] / 1e6:6.0f} TF/s | "
          f"{rb[0]:7.1f} {rb[1]:7.1f}  {fl / rb_best / 1e6:6.0f} TF/s | x{ring_best[0] / rb_best:.2f}")
os.environ.pop("PDMK_RING_CFG", None)
