#!/usr/bin/env python3
"""Pin the oracle against the importable pieces of the reference and emit golden fixtures.

Runs ONLY in the build container (needs /root/reference); the fixtures it writes under tests/golden/ are plain data
(inputs + expected outputs produced by the REFERENCE's own functions / vendored CompVis twins) and travel to the GPU box.

Checked:
  pdm.utils.metric_utils.compute_snr            vs oracle step.compute_snr / min_snr_weights
  pdm.utils.estimation_utils.hard_concrete      vs oracle arch.hard_concrete
  pdm/models/gates.py (by file path)            WidthGate / LinearWidthGate mask semantics == physical slicing
  ldm util.timestep_embedding, make_beta_schedule
  ldm openaimodel.ResBlock / Downsample / Upsample,  ldm attention.SpatialTransformer (fwd AND grads)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "baselines/erasing/oldcode_erasing_compvis"))

from pdm_ref import arch, step, unet  # noqa: E402
from pdm_ref.config import UNetConfig  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.manual_seed(0)
report = []


def ok(name, a, b, tol):
    err = float((a - b).abs().max())
    scale = float(b.abs().max()) + 1e-12
    status = "OK " if err <= tol * max(1.0, scale) else "FAIL"
    report.append(f"{status} {name}: max|diff|={err:.3e} (scale {scale:.3e}, tol {tol:g})")
    print(report[-1])
    assert status == "OK ", name


# ---------------------------------------------------------------- losses / gates (reference package itself)
from pdm.utils.metric_utils import compute_snr as ref_compute_snr  # noqa: E402
from pdm.utils.estimation_utils import hard_concrete as ref_hard_concrete  # noqa: E402

ac = step.alphas_cumprod()
sched = types.SimpleNamespace(alphas_cumprod=ac)
t = torch.tensor([0, 1, 17, 250, 499, 750, 998, 999])
snr_ref = ref_compute_snr(sched, t)
ok("compute_snr", step.compute_snr(ac, t), snr_ref, 1e-6)
w_ref = torch.stack([snr_ref + 1, 5.0 * torch.ones_like(snr_ref)], dim=1).min(dim=1)[0] / (snr_ref + 1)  # trainer.py:2457-2466
ok("min_snr_weights(v-pred)", step.min_snr_weights(ac, t), w_ref, 1e-6)
x = torch.tensor([[0.2, 0.5, 0.9, 0.49999, 0.0, 1.0]])
ok("hard_concrete", arch.hard_concrete(x), ref_hard_concrete(x).detach(), 0)

spec = importlib.util.spec_from_file_location("ref_gates", os.path.join(REF, "pdm/models/gates.py"))
ref_gates = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_gates)
wg = ref_gates.WidthGate(32)
gate = (torch.rand(1, 32) > 0.4).float()
wg.set_structure_value(gate)
xx = torch.randn(2, 64, 4, 4)
ok("WidthGate == channel-group mask", xx * gate.repeat_interleave(2, dim=1)[:, :, None, None], wg(xx), 0)
lg = ref_gates.LinearWidthGate(32)
lg.set_structure_value(gate)
xl = torch.randn(2, 5, 128)
ok("LinearWidthGate == last-dim mask", xl * gate.repeat_interleave(4, dim=1)[:, None, :], lg(xl), 0)

# ---------------------------------------------------------------- vendored CompVis twins
from ldm.modules.diffusionmodules import util as ldm_util  # noqa: E402
from ldm.modules.diffusionmodules.openaimodel import ResBlock, Downsample, Upsample  # noqa: E402
from ldm.modules.attention import SpatialTransformer  # noqa: E402

tt = torch.tensor([0, 1, 500, 999])
ok("timestep_embedding(320)", unet.timestep_embedding(tt, 320), ldm_util.timestep_embedding(tt, 320), 1e-6)
betas64 = ldm_util.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
ac64 = torch.from_numpy(np.cumprod(1.0 - np.asarray(betas64)))
ok("alphas_cumprod vs ldm float64 schedule", ac.double(), ac64, 1e-5)


def sd_from_resblock(rb, p):
    sd = {p + ".norm1.weight": rb.in_layers[0].weight, p + ".norm1.bias": rb.in_layers[0].bias,
          p + ".conv1.weight": rb.in_layers[2].weight, p + ".conv1.bias": rb.in_layers[2].bias,
          p + ".time_emb_proj.weight": rb.emb_layers[1].weight, p + ".time_emb_proj.bias": rb.emb_layers[1].bias,
          p + ".norm2.weight": rb.out_layers[0].weight, p + ".norm2.bias": rb.out_layers[0].bias,
          p + ".conv2.weight": rb.out_layers[3].weight, p + ".conv2.bias": rb.out_layers[3].bias}
    if not isinstance(rb.skip_connection, torch.nn.Identity):
        sd[p + ".conv_shortcut.weight"] = rb.skip_connection.weight
        sd[p + ".conv_shortcut.bias"] = rb.skip_connection.bias
    return sd


golden = {}
for (cin, cout) in ((64, 64), (96, 64)):
    rb = ResBlock(cin, 256, 0.0, out_channels=cout)
    for prm in rb.parameters():                      # un-zero the zero_module conv, jitter the norms
        torch.nn.init.normal_(prm, std=0.05) if prm.dim() > 1 else torch.nn.init.normal_(prm, mean=0.3, std=0.2)
    xin = torch.randn(2, cin, 8, 8, requires_grad=True)
    emb = torch.randn(2, 256, requires_grad=True)
    y_ref = rb(xin, emb)
    sd = sd_from_resblock(rb, "r")
    y = unet.resblock(sd, "r", xin, emb, 32, 32)
    ok(f"ResBlock {cin}->{cout} fwd", y, y_ref.detach(), 1e-5)
    gref = torch.autograd.grad(y_ref.square().sum(), [xin, emb, rb.in_layers[2].weight])
    gmine = torch.autograd.grad(y.square().sum(), [xin, emb, sd["r.conv1.weight"]])
    for nm, a, b in zip(("dx", "demb", "dconv1"), gmine, gref):
        ok(f"ResBlock {cin}->{cout} {nm}", a, b, 1e-4)
    tag = f"resblock_{cin}_{cout}"
    golden[tag + "_x"] = xin.detach().numpy()
    golden[tag + "_emb"] = emb.detach().numpy()
    golden[tag + "_y"] = y_ref.detach().numpy()
    for k, v in sd.items():
        golden[tag + "_w_" + k] = v.detach().numpy()

ds = Downsample(64, True)
us = Upsample(64, True)
xin = torch.randn(2, 64, 8, 8)
ok("Downsample conv s2", torch.nn.functional.conv2d(xin, ds.op.weight, ds.op.bias, stride=2, padding=1), ds(xin).detach(), 1e-6)
yu = torch.nn.functional.conv2d(torch.nn.functional.interpolate(xin, scale_factor=2.0, mode="nearest"),
                                us.conv.weight, us.conv.bias, padding=1)
ok("Upsample nearest+conv", yu, us(xin).detach(), 1e-6)

C, H, ctxd = 128, 2, 48
st = SpatialTransformer(C, H, 64, depth=1, context_dim=ctxd)
for prm in st.parameters():
    torch.nn.init.normal_(prm, std=0.05) if prm.dim() > 1 else torch.nn.init.normal_(prm, mean=0.3, std=0.2)
tb = st.transformer_blocks[0]
p, tp = "a", "a.transformer_blocks.0"
sd = {p + ".norm.weight": st.norm.weight, p + ".norm.bias": st.norm.bias,
      p + ".proj_in.weight": st.proj_in.weight[:, :, 0, 0], p + ".proj_in.bias": st.proj_in.bias,
      p + ".proj_out.weight": st.proj_out.weight[:, :, 0, 0], p + ".proj_out.bias": st.proj_out.bias}
for i, nrm in ((1, tb.norm1), (2, tb.norm2), (3, tb.norm3)):
    sd[f"{tp}.norm{i}.weight"], sd[f"{tp}.norm{i}.bias"] = nrm.weight, nrm.bias
for an, at in (("attn1", tb.attn1), ("attn2", tb.attn2)):
    sd[f"{tp}.{an}.to_q.weight"], sd[f"{tp}.{an}.to_k.weight"], sd[f"{tp}.{an}.to_v.weight"] = \
        at.to_q.weight, at.to_k.weight, at.to_v.weight
    sd[f"{tp}.{an}.to_out.0.weight"], sd[f"{tp}.{an}.to_out.0.bias"] = at.to_out[0].weight, at.to_out[0].bias
sd[f"{tp}.ff.net.0.proj.weight"], sd[f"{tp}.ff.net.0.proj.bias"] = tb.ff.net[0].proj.weight, tb.ff.net[0].proj.bias
sd[f"{tp}.ff.net.2.weight"], sd[f"{tp}.ff.net.2.bias"] = tb.ff.net[2].weight, tb.ff.net[2].bias
xin = torch.randn(2, C, 4, 4, requires_grad=True)
ctx = torch.randn(2, 13, ctxd)
y_ref = st(xin, ctx)
y = unet.transformer2d(sd, p, xin, ctx, H, H, 64, 32)
ok("SpatialTransformer fwd", y, y_ref.detach(), 1e-5)
gref = torch.autograd.grad(y_ref.square().sum(), [xin, tb.attn2.to_k.weight, tb.ff.net[0].proj.weight])
gmine = torch.autograd.grad(y.square().sum(), [xin, sd[f"{tp}.attn2.to_k.weight"], sd[f"{tp}.ff.net.0.proj.weight"]])
for nm, a, b in zip(("dx", "dWk2", "dWff"), gmine, gref):
    ok(f"SpatialTransformer {nm}", a, b, 1e-4)
golden["st_x"], golden["st_ctx"], golden["st_y"] = xin.detach().numpy(), ctx.numpy(), y_ref.detach().numpy()
for k, v in sd.items():
    golden["st_w_" + k] = v.detach().numpy()

# ---------------------------------------------------------------- known answers (scalars)
golden["snr_t"] = t.numpy()
golden["snr_ref"] = snr_ref.numpy()
golden["minsnr_w_ref"] = w_ref.numpy()
golden["temb_t"] = tt.numpy()
golden["temb_ref"] = ldm_util.timestep_embedding(tt, 320).numpy()
golden["alphas_cumprod_f64"] = ac64.numpy()
np.savez_compressed(os.path.join(GOLD, "reference_twins.npz"), **golden)

# arch-vector layout facts: 1606 width + 14 depth = 1620
cfg = UNetConfig.sd21()
s = arch.structure(cfg)
nw = sum(sum(w) for w in s["width"])
nd = sum(sum(d) for d in s["depth"])
assert (nw, nd) == (1606, 14), (nw, nd)
print(f"OK  arch vector layout: {nw} width + {nd} depth = {nw + nd}")
report.append(f"OK  arch vector layout: {nw}+{nd}")
with open(os.path.join(GOLD, "reference_twins.report.txt"), "w") as f:
    f.write("\n".join(report) + "\n")
print("all reference-twin checks passed; fixtures ->", GOLD)
