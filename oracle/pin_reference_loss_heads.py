#!/usr/bin/env python3
"""Pin the oracle's loss-head composition with outputs PRODUCED BY THE REFERENCE'S OWN CODE.

Runs ONLY in the build container (reads /root/reference).  `pdm/training/trainer.py` cannot be imported here
(`torchvision`, `diffusers`, `accelerate` are missing), but the loss heads inside `UnetFineTuner.step`
(trainer.py:2451-2488: DDPM min-SNR + block-feature + output-distillation) and `BilevelUnetFineTuner.upper_step`
(trainer.py:2983-3001: negative-guidance distillation + block term) are plain tensor code.  The statements of those two
tails are taken out of the reference's source AT RUN TIME (ast; nothing is copied into this repository) and executed
unmodified on seeded tensors, with `self` an attribute bag carrying what they read (`config` weights, the scheduler's
`alphas_cumprod` / `prediction_type`, the two hooked-activation dicts) and the reference's own `compute_snr`.
Committed: tests/golden/reference_loss_heads.npz = the inputs, the four returned scalars and the gradients w.r.t. the
student prediction and one student block activation, for three weight settings per head.
"""
import ast
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, REF)
from pdm.utils.metric_utils import compute_snr as ref_compute_snr  # noqa: E402  (the reference's own)
from pdm_ref import step as ostep  # noqa: E402

NS = types.SimpleNamespace
KEYS = ("d0", "d1", "d2", "d3", "m", "u0", "u1", "u2", "u3")


def tail_of(cls, fn, first_stmt_pred):
    """Compile the statements of cls.fn from the first one matching `first_stmt_pred` to the end into a function of
    (self, **locals) that returns what the reference returns."""
    path = os.path.join(REF, "pdm/training/trainer.py")
    tree = ast.parse(open(path).read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name == fn:
                    idx = next(i for i, st in enumerate(f.body) if first_stmt_pred(st))
                    return f.body[idx:], path
    raise KeyError((cls, fn))


def run(stmts, path, env):
    fn = ast.FunctionDef(name="_tail", args=ast.arguments(posonlyargs=[], args=[ast.arg(arg=k_) for k_ in env], kwonlyargs=[],
                                                          kw_defaults=[], defaults=[]),
                         body=stmts, decorator_list=[])
    mod = ast.fix_missing_locations(ast.Module(body=[fn], type_ignores=[]))
    ns = {"torch": torch, "F": F, "compute_snr": ref_compute_snr}
    exec(compile(mod, path, "exec"), ns)
    return ns["_tail"](**env)


is_snr_if = lambda st: isinstance(st, ast.If) and "snr_gamma" in ast.unparse(st.test)
is_diff0 = lambda st: isinstance(st, ast.Assign) and ast.unparse(st.targets[0]) == "diff_loss"
main_tail, path = tail_of("UnetFineTuner", "step", is_snr_if)
upper_tail, _ = tail_of("BilevelUnetFineTuner", "upper_step", is_diff0)

ac = ostep.alphas_cumprod()
g = torch.Generator().manual_seed(2024)
B = 3
out = {}


def make(n_keys=9):
    pred = torch.randn(B, 4, 8, 8, generator=g, requires_grad=True)
    acts_s = {k_: torch.randn(B, 6 + i, 4, 4, generator=g, requires_grad=(i == 4)) for i, k_ in enumerate(KEYS[:n_keys])}
    acts_t = {k_: torch.randn(B, 6 + i, 4, 4, generator=g) for i, k_ in enumerate(KEYS[:n_keys])}
    return pred, acts_s, acts_t


def me(cfg_losses, acts_s, acts_t):
    return NS(config=NS(training=NS(losses=cfg_losses)), accelerator=NS(device="cpu"),
              noise_scheduler=NS(alphas_cumprod=ac, config=NS(prediction_type="v_prediction")),
              block_act_student=acts_s, block_act_teacher=acts_t)


for case, (gamma, wd, wb, ws) in enumerate(((5.0, 1.0, 0.1, 2.0), (None, 0.7, 0.0, 1.5), (3.0, 1.0, 0.25, 0.0))):
    pred, acts_s, acts_t = make()
    target, full = torch.randn(B, 4, 8, 8, generator=g), torch.randn(B, 4, 8, 8, generator=g)
    t = torch.tensor([3, 500, 998])
    losses = NS(diffusion_loss=NS(snr_gamma=gamma, weight=wd), block_loss=NS(weight=wb), distillation_loss=NS(weight=ws))
    r = run(main_tail, path, dict(self=me(losses, acts_s, acts_t), model_pred=pred, target=target, timesteps=t,
                                  full_model_pred=full))
    r[0].backward()
    p = f"main{case}_"
    out.update({p + "pred": pred.detach(), p + "target": target, p + "full": full, p + "t": t,
                p + "cfg": torch.tensor([float("nan") if gamma is None else gamma, wd, wb, ws]),
                p + "out": torch.stack([x.detach().float().reshape(()) for x in r]),
                p + "dpred": pred.grad, p + "dact_m": acts_s["m"].grad if acts_s["m"].grad is not None else torch.zeros_like(acts_s["m"])})
    for k_ in KEYS:
        out[p + "as_" + k_], out[p + "at_" + k_] = acts_s[k_].detach(), acts_t[k_]

for case, (ws, wb) in enumerate(((1.0, 0.0), (0.5, 0.3), (2.0, 0.0))):
    pred, acts_s, acts_t = make()
    e_c, e_u = torch.randn(B, 4, 8, 8, generator=g), torch.randn(B, 4, 8, 8, generator=g)
    losses = NS(block_loss=NS(upper_weight=wb), distillation_loss=NS(upper_weight=ws))
    r = run(upper_tail, path, dict(self=me(losses, acts_s, acts_t), model_pred=pred, full_model_pred_cond=e_c,
                                   full_model_pred_uncond=e_u))
    r[0].backward()
    p = f"upper{case}_"
    out.update({p + "pred": pred.detach(), p + "e_c": e_c, p + "e_u": e_u, p + "cfg": torch.tensor([ws, wb]),
                p + "out": torch.stack([torch.as_tensor(x).detach().float().reshape(()) for x in r]),
                p + "dpred": pred.grad, p + "dact_m": acts_s["m"].grad if acts_s["m"].grad is not None else torch.zeros_like(acts_s["m"])})
    for k_ in KEYS:
        out[p + "as_" + k_], out[p + "at_" + k_] = acts_s[k_].detach(), acts_t[k_]

dst = os.path.join(ROOT, "tests", "golden", "reference_loss_heads.npz")
np.savez_compressed(dst, **{k_: v.numpy() for k_, v in out.items()})
print("wrote", dst, {k_: out[k_].tolist() for k_ in out if k_.endswith("_out")})
