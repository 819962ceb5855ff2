"""Diagnostic: per-tensor gradient error of the fp32 engine vs the oracle with and without the e4m3 attention rounding (tiny topology)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
import torch
import test_step_parity_gpu as T
from pdm_ref import step as ostep, unet as ounet, weights as oweights
from pdm.training.bilevel import BilevelStepper

def run(fp8):
    ocfg, dense, psd, info, student, teacher = T._setup(torch.float32)
    lat, noise, t, ehs, empty = T._inputs()
    ac = ostep.alphas_cumprod(); tinfo = oweights.dense_info(ocfg)
    ounet.ATTN_FP8 = fp8
    P = {k_: v.clone().requires_grad_(True) for k_, v in psd.items()}
    loss = ostep.main_step_loss((P, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs)[0]
    loss.backward(); ounet.ATTN_FP8 = False
    if fp8:
        for m_ in (student, teacher): m_.set_attention_precision("fp8_e4m3")
    st = BilevelStepper(student, teacher)
    tot = st.total(st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda()))[0]
    grads = student.store.state_dict(arena=student.store.grad)
    errs = sorted(((T._rel(grads[n], p.grad), n) for n, p in P.items()), reverse=True)
    print("fp8" if fp8 else "plain", "loss", tot, float(loss.detach()))
    for e, n in errs[:12]: print("   %.2e  %s" % (e, n))
    import statistics
    print("   median %.2e" % statistics.median(e for e, _ in errs))
run(False); run(True)
