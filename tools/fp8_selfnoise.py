"""CPU: how far the ORACLE's own gradients move under a 2e-6 relative perturbation of the weights, with and without the e4m3 attention
rounding (tiny topology).  Rounding is discontinuous: an element of Q/K/V that sits within the perturbation of a tie lands one grid
step (6-12 %) away, so the "fp8_e4m3" step is reproducible between two fp32 implementations only up to those flips.  Measured
(round 4): plain max 8e-5 / median 2.5e-5; rounded max 2.4e-2 (to_k / to_q weights) / median 9e-4, loss 3e-6.  This is the envelope
tests/test_step_parity_gpu.py::test_fp8_e4m3_attention_precision_matches_the_oracle uses for its gradient bound."""
import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
from pdm_ref import arch as oarch, weights as oweights, step as ostep, unet as ounet
from pdm_ref.config import UNetConfig as OCfg
ocfg = OCfg.tiny()
dense = oweights.init_dense_state_dict(ocfg, seed=0)
av = oarch.random_arch_vector(ocfg, 0.55, seed=0, drop_depth=(1,5,9,12))
psd, info = oweights.prune_state_dict(dense, ocfg, av)
tinfo = oweights.dense_info(ocfg)
g = torch.Generator().manual_seed(43)
lat = torch.randn(2,4,16,16,generator=g); noise = torch.randn(2,4,16,16,generator=g); t = torch.tensor([10,800]); ehs = torch.randn(2,13,64,generator=g)
def run(dt, fp8):
    ounet.ATTN_FP8 = fp8
    ac = ostep.alphas_cumprod()
    gg = torch.Generator().manual_seed(1)
    P = {k: (v * (1 + dt * torch.randn(v.shape, generator=gg))).requires_grad_(True) for k, v in psd.items()}
    D = dense; dt = torch.float32
    loss = ostep.main_step_loss((P, info), (D, tinfo), ocfg, ac, lat.to(dt), noise.to(dt), t, ehs.to(dt))[0]
    loss.backward(); ounet.ATTN_FP8 = False
    return float(loss.detach()), {k: p.grad.float() for k, p in P.items()}
rel = lambda a,b: (a-b).abs().max().item()/(b.abs().max().item()+1e-12)
for fp8 in (False, True):
    l32, g32 = run(0.0, fp8); l64, g64 = run(2e-6, fp8)
    errs = sorted(((rel(g32[n], g64[n]), n) for n in g32), reverse=True)
    print("fp8" if fp8 else "plain", l32, l64, abs(l32-l64)/l64)
    for e, n in errs[:8]: print("   %.2e %s" % (e, n))
    import statistics; print("   median %.2e" % statistics.median(e for e,_ in errs))
