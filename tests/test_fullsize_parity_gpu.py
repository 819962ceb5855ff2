"""Full-size parity (-m gpu): the REAL SD-2.1 topology against the CPU oracle, gradients included.

Round-1 stopped at full-size *losses*; this file closes the remaining BASELINE.json configurations:

  configs[0]/[1]/[2]  MAC-budget-0.55 student, B=1, 64x64 latent: main step AND upper (concept-suppression) step -
                      losses + parameter gradients (every tensor; a named set spread over down/mid/up is asserted at the
                      tight tolerance) for the fp32 engine (2e-3 of each tensor's scale) and the bf16 engine (cosine >= 0.98)
  configs[3]          "82 %-pruned" in its second reading: keep ~ 0.18 of the MACs - main-step losses AND gradients
  configs[4]          dense student, 96x96 latents (768^2 images), B=1 (N = 9216 self-attention) - main-step losses AND
                      gradients, plus one upper step (loss + gradients)

Reference arithmetic: pdm/training/trainer.py:2403-2488 (step), :2904-3001 (upper_step); attention blocks.py:257-277.
The oracle runs ONCE per (budget, step kind) (module-scoped cache) and both engine dtypes are compared with it.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

# gradients asserted at the tight tolerance: conv1/conv2, shortcut, samplers, q/k/v/out of both attentions (the K/V rows
# are column slices of the engine's ONE batched cross-attention projection), GEGLU + FF out, every norm kind, the
# time-embedding MLP and time_emb_proj rows (row blocks of the engine's ONE batched projection), first and last conv
NAMED = [
    "conv_in.weight", "conv_out.weight", "conv_out.bias", "conv_norm_out.weight", "conv_norm_out.bias",
    "time_embedding.linear_1.weight", "time_embedding.linear_2.weight", "time_embedding.linear_2.bias",
    "down_blocks.0.resnets.0.norm1.weight", "down_blocks.0.resnets.0.conv1.weight", "down_blocks.0.resnets.0.conv1.bias",
    "down_blocks.0.resnets.0.time_emb_proj.weight", "down_blocks.0.resnets.0.norm2.bias",
    "down_blocks.0.resnets.0.conv2.weight",
    "down_blocks.0.attentions.0.norm.weight", "down_blocks.0.attentions.0.proj_in.weight",
    "down_blocks.0.attentions.0.transformer_blocks.0.norm1.weight",
    "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight",
    "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_k.weight",
    "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_v.weight",
    "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.weight",
    "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_q.weight",
    "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight",
    "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_v.weight",
    "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_out.0.bias",
    "down_blocks.0.attentions.0.transformer_blocks.0.norm3.bias",
    "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
    "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.bias",
    "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.2.weight",
    "down_blocks.0.attentions.0.proj_out.weight",
    "down_blocks.0.downsamplers.0.conv.weight",
    "down_blocks.1.resnets.0.conv_shortcut.weight", "down_blocks.1.resnets.0.conv1.weight",
    "down_blocks.1.attentions.1.transformer_blocks.0.attn1.to_k.weight",
    "down_blocks.2.resnets.1.conv2.weight", "down_blocks.2.attentions.0.transformer_blocks.0.ff.net.2.weight",
    "down_blocks.2.attentions.1.transformer_blocks.0.attn2.to_v.weight",
    "down_blocks.3.resnets.0.conv1.weight", "down_blocks.3.resnets.1.time_emb_proj.weight",
    "mid_block.resnets.0.conv2.weight", "mid_block.attentions.0.transformer_blocks.0.attn1.to_v.weight",
    "mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.weight", "mid_block.resnets.1.norm1.weight",
    "up_blocks.0.resnets.0.conv1.weight", "up_blocks.0.resnets.2.conv_shortcut.weight", "up_blocks.0.upsamplers.0.conv.weight",
    "up_blocks.1.resnets.2.conv1.weight", "up_blocks.1.attentions.2.transformer_blocks.0.attn2.to_k.weight",
    "up_blocks.1.attentions.0.transformer_blocks.0.ff.net.2.weight", "up_blocks.1.upsamplers.0.conv.weight",
    "up_blocks.2.resnets.0.conv_shortcut.weight", "up_blocks.2.attentions.1.transformer_blocks.0.attn1.to_q.weight",
    "up_blocks.2.resnets.2.conv2.weight",
    "up_blocks.3.resnets.0.time_emb_proj.weight", "up_blocks.3.resnets.2.conv1.weight", "up_blocks.3.resnets.2.norm2.weight",
    "up_blocks.3.attentions.2.transformer_blocks.0.attn1.to_out.0.weight",
    "up_blocks.3.attentions.2.transformer_blocks.0.ff.net.0.proj.weight", "up_blocks.3.attentions.2.proj_out.weight",
]

_CACHE = {}


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def _inputs(hw=64):
    g = torch.Generator().manual_seed(43)
    lat, noise = torch.randn(1, 4, hw, hw, generator=g), torch.randn(1, 4, hw, hw, generator=g)
    t, ehs = torch.tensor([431]), torch.randn(1, 77, 1024, generator=g)
    empty = torch.randn(1, 77, 1024, generator=g)
    return lat, noise, t, ehs, empty


def _dense(seed=0):
    """The dense SD-2.1 state dict every model of this module shares (engine random init, fp32 master)."""
    if ("dense", seed) not in _CACHE:
        from pdm.models.unet.spec import UNetConfig
        from pdm.models.unet.unet_2d_conditional import UNet2DConditionModelPruned
        m = UNet2DConditionModelPruned(UNetConfig.sd21(), None, "cuda:0", torch.float32, train=False, seed=seed)
        _CACHE[("dense", seed)] = m.state_dict()
        del m
        torch.cuda.empty_cache()
    return _CACHE[("dense", seed)]


def _oracle(kind, budget, hw=64, grads=True):
    """(losses, {name: grad}) of the CPU oracle for one step kind; computed once per module run."""
    key = (kind, budget, hw, grads)
    if key in _CACHE:
        return _CACHE[key]
    from pdm_ref import step as ostep, weights as oweights
    from pdm_ref.config import UNetConfig as OCfg
    from pdm.models.unet.spec import UNetConfig, arch_vector_for_budget
    ocfg, cfg = OCfg.sd21(), UNetConfig.sd21()
    dense = _dense()
    if budget >= 1.0:      # dense student with its OWN weights (seed 1), so that the distillation / block terms are not 0
        psd, info, av = _dense(1), oweights.dense_info(ocfg), None
    else:
        av = arch_vector_for_budget(cfg, budget, hw=hw)[0]
        psd, info = oweights.prune_state_dict(dense, ocfg, av)
    lat, noise, t, ehs, empty = _inputs(hw)
    P = {k_: (v.clone().requires_grad_(True) if grads else v) for k_, v in psd.items()}
    tch = (dense, oweights.dense_info(ocfg))
    with torch.set_grad_enabled(grads):
        if kind == "main":
            out = ostep.main_step_loss((P, info), tch, ocfg, ostep.alphas_cumprod(), lat, noise, t, ehs)
        else:
            out = ostep.upper_step_loss((P, info), tch, ocfg, ostep.alphas_cumprod(), lat, noise, t, ehs, empty)
        if grads:
            out[0].backward()
    res = (tuple(float(x.detach()) for x in out[:4]), {n: p.grad for n, p in P.items()} if grads else None, av)
    _CACHE[key] = res
    return res


def _models(dtype, av, train=True, student_sd=None):
    from pdm.models.unet.spec import UNetConfig
    from pdm.models.unet.unet_2d_conditional import UNet2DConditionModelPruned
    cfg = UNetConfig.sd21()
    dense = _dense()
    teacher = UNet2DConditionModelPruned(cfg, None, "cuda:0", dtype, train=False, init=False)
    teacher.load_dense_or_pruned(dense)
    student = UNet2DConditionModelPruned(cfg, av, "cuda:0", dtype, train=train, init=False)
    student.load_dense_or_pruned(dense if student_sd is None else student_sd)
    return student, teacher


def _check_grads(student, ref, dn):
    got = student.store.state_dict(arena=student.store.grad)
    assert set(got) == set(ref)
    missing = [n for n in NAMED if n not in ref]
    assert not missing, missing
    bad = []
    if dn == "f32":
        for n, g in ref.items():
            tol = 2e-3 if n in NAMED else 1e-2
            r = _rel(got[n], g)
            if not r <= tol:
                bad.append((n, r))
    else:
        for n, g in ref.items():
            if g.numel() < 4096 and n not in NAMED:
                continue
            cos = torch.nn.functional.cosine_similarity(got[n].flatten().double(), g.flatten().double(), dim=0).item()
            if not cos >= (0.98 if g.numel() >= 1024 else 0.95):
                bad.append((n, cos))
    assert not bad, (len(bad), bad[:10])
    # the packed arena's padding carries exactly-zero gradients at full size too
    packed = float(student.store.grad.double().abs().sum())
    logical = float(sum(v.double().abs().sum() for v in got.values()))
    assert math.isfinite(packed) and abs(packed - logical) <= 1e-5 * logical


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_full_size_main_step_gradients_match_oracle(dev, dn):
    from pdm.training.bilevel import BilevelStepper
    (loss, diff, dist_, block), gref, av = _oracle("main", 0.55)
    dtype = torch.float32 if dn == "f32" else torch.bfloat16
    student, teacher = _models(dtype, av)
    lat, noise, t, ehs, _ = _inputs()
    st = BilevelStepper(student, teacher)
    tot, d, s, b = st.total(st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda()))
    tol = 1e-3 if dn == "f32" else 3e-2
    for name, got, ref in (("diff", d, diff), ("dist", s, dist_), ("block", b, block), ("total", tot, loss)):
        assert abs(got - ref) <= tol * max(abs(ref), 1e-3), (name, got, ref)
    _check_grads(student, gref, dn)


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_full_size_upper_step_loss_and_gradients_match_oracle(dev, dn):
    from pdm.training.bilevel import BilevelStepper
    (loss, _, dist_, _), gref, av = _oracle("upper", 0.55)
    dtype = torch.float32 if dn == "f32" else torch.bfloat16
    student, teacher = _models(dtype, av)
    lat, noise, t, ehs, empty = _inputs()
    st = BilevelStepper(student, teacher)
    tot, _, s, _ = st.total(st.upper_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), empty.cuda()), upper=True)
    tol = 1e-3 if dn == "f32" else 3e-2
    assert abs(tot - loss) <= tol * abs(loss), (tot, loss)
    assert abs(s - dist_) <= tol * abs(dist_), (s, dist_)
    _check_grads(student, gref, dn)


def test_full_size_bf16_error_against_both_oracle_precisions(dev):
    """One full-size data point for the bf16 engine's distance from fp32 (real SD-2.1, MAC budget 0.55, B = 1, 64x64 latent, fixed
    weights): the student's nine hooked block activations and prediction and the teacher's prediction as relative L2 errors
    against the fp32 oracle, next to the same quantities of the oracle evaluated under the reference's mixed-precision policy
    (pdm_ref/step.py mixed="cuda": CUDA autocast, trainer.py:516-527).  The HIP engine must be no further from fp32 than 1.15 x
    that evaluation on every tensor (tiny topology: 0.82-0.95 x, tests/test_step_parity_gpu.py), and the three main-step loss
    totals (HIP bf16 / oracle mixed / oracle fp32) are printed and held to the 3e-2 the other bf16 tests use."""
    from pdm_ref import step as ostep, unet as ounet, weights as oweights
    from pdm_ref.config import UNetConfig as OCfg
    from pdm.training.bilevel import BilevelStepper
    (loss32, _, _, _), _, av = _oracle("main", 0.55)
    ocfg = OCfg.sd21()
    dense = _dense()
    psd, info = oweights.prune_state_dict(dense, ocfg, av)
    tinfo = oweights.dense_info(ocfg)
    lat, noise, t, ehs, _ = _inputs()
    ac = ostep.alphas_cumprod()
    noisy = ostep.add_noise(ac, lat, noise, t)
    ref, pol = {}, {}
    for store, mixed in ((ref, False), (pol, "cuda")):
        mp = ostep._mixed(mixed)
        acts = {}
        with torch.no_grad(), mp.ctx():
            store["pred"] = ounet.unet_forward(psd, ocfg, info, noisy, t, ehs, acts).float()
            store["teacher"] = ounet.unet_forward(mp.teacher_sd(dense), ocfg, tinfo, noisy, t, ehs).float()
        store.update({k_: v.float() for k_, v in acts.items()})
    with torch.no_grad():
        loss_mix = float(ostep.main_step_loss((psd, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs, mixed="cuda")[0])
    student, teacher = _models(torch.bfloat16, av)
    acts = {}
    for i, h in enumerate(student.down_blocks):
        h.register_forward_hook(lambda m, inp, out, i=i: acts.__setitem__(f"d{i}", out[0]))
    student.mid_block.register_forward_hook(lambda m, inp, out: acts.__setitem__("m", out))
    for i, h in enumerate(student.up_blocks):
        h.register_forward_hook(lambda m, inp, out, i=i: acts.__setitem__(f"u{i}", out))
    hip = {"pred": student.eval()(noisy, t, ehs).sample.float().cpu(), "teacher": teacher(noisy, t, ehs).sample.float().cpu()}
    hip.update({k_: v.float().cpu() for k_, v in acts.items()})
    student.train()
    l2 = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    report, bad = {}, []
    for key in list(ostep.BLOCK_KEYS) + ["pred", "teacher"]:
        e_hip, e_pol = l2(hip[key], ref[key]), l2(pol[key], ref[key])
        report[key] = (round(e_hip, 5), round(e_pol, 5))
        assert e_pol > 1e-3, (key, e_pol)
        if e_hip > 1.15 * e_pol:
            bad.append(key)
    print("full-size relative L2 error vs fp32 (HIP bf16, oracle mixed):", report)
    assert not bad, (bad, report)
    st = BilevelStepper(student, teacher)
    tot = st.total(st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), backward=False))[0]
    print(f"full-size main-step loss: HIP bf16 {tot:.6f}  oracle mixed {loss_mix:.6f}  oracle fp32 {loss32:.6f}")
    assert abs(tot - loss32) <= 3e-2 * abs(loss32) and abs(loss_mix - loss32) <= 3e-2 * abs(loss32), (tot, loss_mix, loss32)


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_keep_018_student_main_step_losses_and_gradients(dev, dn):
    """BASELINE configs[3] read as "82 % pruned" = 18 % of the MACs kept: the four loss heads and every parameter gradient
    (NAMED set at the tight tolerance, as at budget 0.55) against the oracle.  trainer.py:2403-2488."""
    from pdm.training.bilevel import BilevelStepper
    (loss, diff, dist_, block), gref, av = _oracle("main", 0.18)
    dtype = torch.float32 if dn == "f32" else torch.bfloat16
    student, teacher = _models(dtype, av)
    lat, noise, t, ehs, _ = _inputs()
    st = BilevelStepper(student, teacher)
    tot, d, s, b = st.total(st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda()))
    tol = 1e-3 if dn == "f32" else 3e-2
    for name, got, ref in (("diff", d, diff), ("dist", s, dist_), ("block", b, block), ("total", tot, loss)):
        assert abs(got - ref) <= tol * max(abs(ref), 1e-3), (name, got, ref)
    _check_grads(student, gref, dn)


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_dense_student_96x96_latents_main_step_losses_and_gradients(dev, dn):
    """BASELINE configs[4] shape: dense (unpruned) student, 768^2 images = 96x96 latents, N = 9216 self-attention tokens: loss
    heads and every parameter gradient against the oracle.  (The config's fp8 attention is a precision option that this build
    does not have - DESIGN.md 10; parity is checked on the fp32 engine at 1e-3 / 2e-3 and on the bf16 engine.)"""
    from pdm.training.bilevel import BilevelStepper
    (loss, diff, dist_, block), gref, av = _oracle("main", 1.0, hw=96)
    dtype = torch.float32 if dn == "f32" else torch.bfloat16
    student, teacher = _models(dtype, None, student_sd=_dense(1))
    lat, noise, t, ehs, _ = _inputs(96)
    st = BilevelStepper(student, teacher)
    tot, d, s, b = st.total(st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda()))
    tol = 1e-3 if dn == "f32" else 3e-2
    for name, got, ref in (("diff", d, diff), ("dist", s, dist_), ("block", b, block), ("total", tot, loss)):
        assert abs(got - ref) <= tol * max(abs(ref), 1e-3), (name, got, ref)
    _check_grads(student, gref, dn)


def test_dense_student_96x96_latents_upper_step(dev):
    """The upper (concept-suppression) step at configs[4]'s shape, bf16 engine: teacher on the conditional and the empty
    prompt (one 2B pass at N = 9216), student forward, negative-guidance target, backward - loss and every gradient against
    the oracle.  trainer.py:2904-3001."""
    from pdm.training.bilevel import BilevelStepper
    (loss, _, dist_, _), gref, av = _oracle("upper", 1.0, hw=96)
    student, teacher = _models(torch.bfloat16, None, student_sd=_dense(1))
    lat, noise, t, ehs, empty = _inputs(96)
    st = BilevelStepper(student, teacher)
    tot, _, s, _ = st.total(st.upper_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), empty.cuda()), upper=True)
    assert abs(tot - loss) <= 3e-2 * abs(loss) and abs(s - dist_) <= 3e-2 * abs(dist_), (tot, loss, s, dist_)
    _check_grads(student, gref, "bf16")


def _oracle_batch(kind, budget, B):
    """Oracle losses and gradients of a batch of B samples as B runs of one sample each (every head is a mean over the
    batch and GroupNorm statistics are per sample, so loss = mean of the per-sample losses and likewise the gradients):
    B x ~5 s at 16 threads instead of one B-sample autograd graph of ~50 GB."""
    key = ("batch", kind, budget, B)
    if key in _CACHE:
        return _CACHE[key]
    from pdm_ref import step as ostep, weights as oweights
    from pdm_ref.config import UNetConfig as OCfg
    from pdm.models.unet.spec import UNetConfig, arch_vector_for_budget
    ocfg, cfg = OCfg.sd21(), UNetConfig.sd21()
    dense = _dense()
    av = arch_vector_for_budget(cfg, budget, hw=64)[0]
    psd, info = oweights.prune_state_dict(dense, ocfg, av)
    tch = (dense, oweights.dense_info(ocfg))
    lat, noise, t, ehs, empty = _batch_inputs(B)
    ac = ostep.alphas_cumprod()
    P = {k_: v.clone().requires_grad_(True) for k_, v in psd.items()}
    tot = [0.0] * 4
    for b in range(B):
        sl = slice(b, b + 1)
        if kind == "main":
            out = ostep.main_step_loss((P, info), tch, ocfg, ac, lat[sl], noise[sl], t[sl], ehs[sl])
        else:
            out = ostep.upper_step_loss((P, info), tch, ocfg, ac, lat[sl], noise[sl], t[sl], ehs[sl], empty[sl])
        (out[0] / B).backward()                # gradients accumulate into P[...].grad
        tot = [a + float(x.detach()) / B for a, x in zip(tot, out[:4])]
    res = (tuple(tot), {n: p.grad for n, p in P.items()}, av)
    _CACHE[key] = res
    return res


def _batch_inputs(B):
    g = torch.Generator().manual_seed(4343)
    lat, noise = torch.randn(B, 4, 64, 64, generator=g), torch.randn(B, 4, 64, 64, generator=g)
    t, ehs = torch.randint(0, 1000, (B,), generator=g), torch.randn(B, 77, 1024, generator=g)
    empty = torch.randn(1, 77, 1024, generator=g).expand(B, -1, -1).contiguous()
    return lat, noise, t, ehs, empty


def test_benchmarked_configuration_matches_oracle(dev, tmp_path):
    """The configuration bench.py measures - real SD-2.1, budget 0.55, **B = 8, bf16, hipGraph replay with the tuned plans**
    (256-row halo tiles, M = 32 768 ring shapes, slab-mode weight gradients) - against the oracle: main-step losses and
    gradients (the 59 NAMED tensors and every tensor of >= 4096 elements by cosine, as the B = 1 test), plus one upper
    step.  Reference: trainer.py:2403-2488, 2904-3001.  The plan cache of this process is then read back: the step must have
    planned at least one 256-row halo-conv tile and one slab-mode weight gradient; the row-block Linear kernel (picked by
    the tuner for a handful of shapes, box-dependent) is FORCED for a second eager pass over the same batch, which must
    reproduce the graph's losses and gradients - so that all three kernel families of the benchmarked path meet the oracle
    at full size."""
    import os
    from pdm import _pdmk as k
    from pdm.training.bilevel import BilevelStepper, GraphedBilevel
    B = 8
    (loss, diff, dist_, block), gref, av = _oracle_batch("main", 0.55, B)
    (uloss, _, udist, _), ugref, _ = _oracle_batch("upper", 0.55, B)
    student, teacher = _models(torch.bfloat16, av)
    lat, noise, t, ehs, empty = (x.cuda() for x in _batch_inputs(B))
    st = BilevelStepper(student, teacher)
    gr = GraphedBilevel(st, B, 4, 64, 64, 77, 1024, prefetch=True)      # as bench.py builds it: cross-step teacher prefetch, 12 shares
    gr.capture(bilevel=True)
    assert len(gr.g_main.bwd) >= 2 and (gr.g_main.teacher is None) == st.lockstep
    assert gr.prefetch != st.lockstep and (gr.g_main.loss is not None) == gr.prefetch
    # GroupNorm statistics from the producing GEMM's epilogue (pdmk_gemm_args.colstat): on unless PDMK_GN_EPI=0, and then the
    # path the captured step runs for most of its GroupNorms (the rest: split-K producers, copied concat halves)
    gc = student.engine.gn_count
    assert gc[0] >= 40 and (not student.engine.gn_epi or gc[1] * 2 >= gc[0]), gc
    print("groupnorms per forward / with statistics from a GEMM epilogue:", gc)
    store = student.store
    # ---- main step: gradients only (no optimiser), replayed twice (the second replay must not see stale state)
    # (prefetch mode: the first replay runs its teacher pass in line and queues the SAME batch's pass - under the token "again" -
    # and the upper step's behind its loss heads; the second replay and the upper step below find them done)
    for i in range(2):
        k.zero_(store.grad)
        gr._load(lat, noise, t, ehs)
        have = gr._claim(False, "again" if i else None, (lat, noise, t, ehs))
        assert have == (bool(i) and gr.prefetch)
        gr._replay_step(gr.g_main, None, have_teacher=have,
                        ahead=gr._ahead_fn((lat, noise, t, ehs), "again", (lat, noise, t, ehs, empty), "up") if i == 0 else None)
    torch.cuda.synchronize()
    tot, d, s, b = st.total(st.losses)
    for name, got, ref in (("diff", d, diff), ("dist", s, dist_), ("block", b, block), ("total", tot, loss)):
        assert abs(got - ref) <= 3e-2 * max(abs(ref), 1e-3), (name, got, ref)
    _check_grads(student, gref, "bf16")
    g_graph = store.grad.clone()
    # ---- upper step
    k.zero_(store.grad)
    gr._load(lat, noise, t, ehs, empty)
    have = gr._claim(True, "up", (lat, noise, t, ehs, empty))
    assert have == gr.prefetch
    gr._replay_step(gr.g_upper, None, have_teacher=have)
    torch.cuda.synchronize()
    utot, _, us, _ = st.total(st.losses, upper=True)
    assert abs(utot - uloss) <= 3e-2 * abs(uloss) and abs(us - udist) <= 3e-2 * abs(udist), (utot, uloss, us, udist)
    _check_grads(student, ugref, "bf16")
    # ---- which plans the step made
    path = str(tmp_path / "plans.txt")
    k.plan_export(path)
    names, slab_wgrads = [], 0
    for line in open(path).read().splitlines()[1:]:
        f = line.split()
        if f[0] != "c":
            continue
        v, cand = [int(x) for x in f[1:11]], int(f[11])
        if v[0] < 8192 and v[2] < 8192:
            continue                                       # not a B = 8, 64^2 / 32^2 shape of this step
        names.append(k.candidate_name(v[3], v[4], cand))
        slab_wgrads += int(v[3] == k.A_COLK and v[5] >= 1000 and v[9] > 1)
    assert any("conv_halo_kernel<256" in n for n in names), sorted(set(names))
    assert slab_wgrads >= 1, sorted(set(names))
    gr.close()
    # ---- the row-block Linear kernel forced on every shape it takes (eager, same batch): same losses and gradients
    rb = [i for i in range(1, 64) if "rowblock_kernel<128" in _safe_name(k, i)]
    assert rb, "no row-block candidate in this build"
    os.environ["PDMK_RING_CFG"] = str(rb[0])
    try:
        k.zero_(store.grad)
        st.defer_reduce = False
        L = st.main_step(lat, noise, t, ehs)
        torch.cuda.synchronize()
        assert k.last_candidate() >= 0
    finally:
        del os.environ["PDMK_RING_CFG"]
    tot2 = st.total(L)[0]
    assert abs(tot2 - tot) <= 1e-2 * abs(tot), (tot2, tot)
    cos = torch.nn.functional.cosine_similarity(store.grad.double(), g_graph.double(), dim=0).item()
    assert cos >= 0.999, cos
    _check_grads(student, gref, "bf16")


def _safe_name(k, i):
    try:
        return k.candidate_name(k.A_ROWK, k.B_ROWK, i)
    except Exception:
        return ""
