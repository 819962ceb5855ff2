#!/usr/bin/env python3
"""Debug driver: the sequence of test_deferred_wt_refresh_is_complete_before_backward with NaN checks per parameter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import test_step_parity_gpu as T
from pdm.training.bilevel import BilevelStepper
from pdm import _pdmk as k
ocfg, dense, psd, info, student, teacher = T._setup(torch.bfloat16)
lat, noise, t, ehs, empty = T._inputs()
store = student.store
st = BilevelStepper(student, teacher, lr=1e-2, upper_lr=1e-2, bilevel=True)
def report(tag):
    torch.cuda.synchronize()
    g = store.grad
    bad = []
    for e in store.entries if hasattr(store, "entries") else []:
        pass
    print(tag, "grad nan:", bool(torch.isnan(g).any()), "inf:", bool(torch.isinf(g).any()), "master nan:", bool(torch.isnan(store.master).any()),
          "max|g|", float(g[torch.isfinite(g)].abs().max()) if torch.isfinite(g).any() else None, "losses", [float(x) for x in st.losses] if hasattr(st, "losses") and st.losses is not None else None, flush=True)
    if torch.isnan(g).any():
        idx = torch.isnan(g).nonzero().flatten()
        print("  first nan offsets", idx[:5].tolist(), "count", idx.numel())
        for key, e in list(store.by_key.items()):
            n = 1
            for d in e.shape: n *= d
            if torch.isnan(g[e.off:e.off + n]).any():
                print("   nan in", key, e.shape)
st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda()); report("main 1")
st.optimizer_step(); report("opt 1")
for i in range(2):
    st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda()); report(f"main {i+2}")
    st.optimizer_step(upper=False); report(f"opt {i+2}")
    st.upper_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), empty.cuda()); report(f"upper {i+2}")
    st.optimizer_step(upper=False); report(f"opt u{i+2}")
