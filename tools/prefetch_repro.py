"""Repro driver for the teacher-prefetch graph path on the tiny topology: python tools/prefetch_repro.py <mode> [sync]"""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (os.path.join(ROOT, "unlearn-ft_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p_)
faulthandler.enable()
import torch  # noqa: E402
from test_step_parity_gpu import _setup  # noqa: E402
from pdm.training.bilevel import BilevelStepper, GraphedBilevel  # noqa: E402

mode, sync = sys.argv[1], len(sys.argv) > 2 and sys.argv[2] == "sync"
g = torch.Generator().manual_seed(5)
batches = [tuple(x.cuda() for x in (torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 4, 16, 16, generator=g),
                                    torch.randint(0, 1000, (2,), generator=g), torch.randn(2, 13, 64, generator=g)))
           for _ in range(4)]
empty = torch.randn(1, 13, 64, generator=g).expand(2, 13, 64).contiguous().cuda()
ocfg, dense, psd, info, student, teacher = _setup(torch.float32)
st = BilevelStepper(student, teacher, lr=1e-4, upper_lr=1e-4, bilevel=True)
gr = GraphedBilevel(st, 2, 4, 16, 16, 13, 64, segments=3, prefetch=(mode != "in_step"))
gr.capture(bilevel=True)
print("captured", mode, len(gr.g_main), flush=True)
for rep in range(10):
    for i, b in enumerate(batches):
        gr.main(*b, nxt=batches[(i + 1) % 4] if mode == "prefetch" else None)
        if sync:
            torch.cuda.synchronize()
        L = st.losses.clone()
        if i == 1:
            gr.upper(*b, empty)
    print("rep", rep, flush=True)
torch.cuda.synchronize()
print("OK", mode, L.tolist(), flush=True)
