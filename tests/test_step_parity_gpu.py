"""Step-level parity (-m gpu): the HIP engine (through the C ABI) against the CPU oracle on identical seeded
(weights, arch vector, latent, noise, timestep, prompt-embed) inputs, tiny U-Net topology (same code paths as SD-2.1:
channel padding, dropped blocks, 13-token cross attention, N=4..256 self attention).

Tolerances (north_star: "loss curves matching the CPU reference to 1e-3"):
  fp32 engine : losses 2e-4 relative, U-Net output 1e-3 of scale, every parameter gradient 2e-3 of that tensor's scale
  bf16 engine : losses 3e-2 relative (bf16 activations/weights, fp32 statistics and accumulation), output 4e-2 of
                scale, gradients compared by cosine similarity >= 0.98 per large tensor
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dtype, drop_depth=(1, 5, 9, 12)):
    from pdm_ref import arch as oarch, weights as oweights
    from pdm_ref.config import UNetConfig as OCfg
    from pdm.models.unet.spec import UNetConfig
    from pdm.models.unet.unet_2d_conditional import UNet2DConditionModelPruned
    ocfg, cfg = OCfg.tiny(), UNetConfig.tiny()
    dense = oweights.init_dense_state_dict(ocfg, seed=0)
    av = oarch.random_arch_vector(ocfg, 0.55, seed=0, drop_depth=drop_depth)
    psd, info = oweights.prune_state_dict(dense, ocfg, av)
    student = UNet2DConditionModelPruned(cfg, av, "cuda:0", dtype, train=True, init=False)
    student.load_dense_or_pruned(dense)
    teacher = UNet2DConditionModelPruned(cfg, None, "cuda:0", dtype, train=False, init=False)
    teacher.load_dense_or_pruned(dense)
    return ocfg, dense, psd, info, student, teacher


def _inputs(B=2, hw=16, T=13, ctx=64, seed=43):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(B, 4, hw, hw, generator=g)
    noise = torch.randn(B, 4, hw, hw, generator=g)
    t = torch.tensor([10, 800][:B])
    ehs = torch.randn(B, T, ctx, generator=g)
    empty = torch.randn(1, T, ctx, generator=g).expand(B, T, ctx).contiguous()
    return lat, noise, t, ehs, empty


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def test_state_dict_roundtrip_matches_reference_pruning(dev):
    ocfg, dense, psd, info, student, teacher = _setup(torch.float32)
    mine = student.state_dict()
    assert set(mine) == set(psd), (sorted(set(mine) ^ set(psd))[:6])
    for kname, v in psd.items():
        assert tuple(mine[kname].shape) == tuple(v.shape), kname
        assert torch.equal(mine[kname], v), kname
    assert set(teacher.state_dict()) == set(dense)


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_forward_matches_oracle(dev, dn):
    from pdm_ref import unet as ounet, weights as oweights
    dtype = torch.float32 if dn == "f32" else torch.bfloat16
    ocfg, dense, psd, info, student, teacher = _setup(dtype)
    lat, noise, t, ehs, _ = _inputs()
    acts_ref = {}
    ref = ounet.unet_forward(psd, ocfg, info, lat, t, ehs, acts_ref)
    acts = {}
    for i, h in enumerate(student.down_blocks):
        h.register_forward_hook(lambda m, inp, out, i=i: acts.__setitem__(f"d{i}", out[0]))
    student.mid_block.register_forward_hook(lambda m, inp, out: acts.__setitem__("m", out))
    for i, h in enumerate(student.up_blocks):
        h.register_forward_hook(lambda m, inp, out, i=i: acts.__setitem__(f"u{i}", out))
    out = student.eval()(lat, t, ehs).sample.cpu()
    tol = 1e-3 if dn == "f32" else 4e-2
    assert _rel(out, ref) < tol, _rel(out, ref)
    for key, r in acts_ref.items():
        assert _rel(acts[key].float().cpu(), r) < tol * 2, (key, _rel(acts[key].float().cpu(), r))
    ref_t = ounet.unet_forward(dense, ocfg, oweights.dense_info(ocfg), lat, t, ehs)
    out_t = teacher(lat, t, ehs).sample.cpu()
    assert _rel(out_t, ref_t) < tol, _rel(out_t, ref_t)


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_main_step_losses_grads_and_adamw(dev, dn):
    from pdm_ref import step as ostep, weights as oweights
    from pdm.training.bilevel import BilevelStepper
    dtype = torch.float32 if dn == "f32" else torch.bfloat16
    ocfg, dense, psd, info, student, teacher = _setup(dtype)
    lat, noise, t, ehs, _ = _inputs()
    ac = ostep.alphas_cumprod()
    P = {k_: v.clone().requires_grad_(True) for k_, v in psd.items()}
    loss, diff, dist_, block, _ = ostep.main_step_loss((P, info), (dense, oweights.dense_info(ocfg)), ocfg, ac, lat,
                                                       noise, t, ehs)
    loss.backward()
    lr = 1e-3
    st = BilevelStepper(student, teacher, lr=lr, upper_lr=lr)
    L = st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda())
    tot, d, s, b = st.total(L)
    ltol = 2e-4 if dn == "f32" else 3e-2
    for name, got, ref in (("diff", d, diff.item()), ("dist", s, dist_.item()), ("block", b, block.item()),
                           ("total", tot, loss.item())):
        assert abs(got - ref) <= ltol * max(abs(ref), 1e-3), (name, got, ref)
    grads = student.store.state_dict(arena=student.store.grad)
    bad = []
    for name, p in P.items():
        gref, g = p.grad, grads[name]
        if dn == "f32":
            if _rel(g, gref) > 2e-3:
                bad.append((name, _rel(g, gref)))
        elif gref.numel() >= 1024:
            cos = torch.nn.functional.cosine_similarity(g.flatten(), gref.flatten(), dim=0).item()
            if cos < 0.98:
                bad.append((name, cos))
    assert not bad, bad[:8]
    # padding of the packed arena must carry exactly-zero gradients (fixed point of the step)
    packed_total = float(student.store.grad.abs().sum())
    logical_total = float(sum(v.abs().sum() for v in grads.values()))
    assert abs(packed_total - logical_total) <= 1e-5 * logical_total
    # AdamW (step 1) against the oracle's restatement of torch.optim.AdamW
    if dn == "f32":
        params = {k_: v.detach().clone() for k_, v in P.items()}
        g_ = {k_: v.grad for k_, v in P.items()}
        m = {k_: torch.zeros_like(v) for k_, v in params.items()}
        v_ = {k_: torch.zeros_like(v) for k_, v in params.items()}
        ostep.adamw_step(params, g_, m, v_, 1, lr)
        st.optimizer_step()
        new = student.state_dict()
        # first Adam step ~ lr*g/(|g|+eps): compare the applied deltas where the gradient is not in the eps regime
        worst = 0.0
        for k_ in params:
            mask = g_[k_].abs() > 1e-3 * g_[k_].abs().max()
            d_ref, d_got = (params[k_] - psd[k_])[mask], (new[k_] - psd[k_])[mask]
            worst = max(worst, (d_got - d_ref).abs().max().item() / lr)
        assert worst < 2e-2, worst
        assert float(student.store.grad.abs().max()) == 0.0


@pytest.mark.parametrize("dn,wb", [("f32", 0.0), ("bf16", 0.0), ("f32", 0.3)])
def test_upper_step_matches_oracle(dev, dn, wb):
    """wb > 0: the upper step's block term (disabled in the shipped configs) against the teacher's LAST call - the
    unconditional half of the 2B teacher batch (trainer.py:2951-2954, 2986-2992)."""
    from pdm_ref import step as ostep, weights as oweights
    from pdm.training.bilevel import BilevelStepper
    dtype = torch.float32 if dn == "f32" else torch.bfloat16
    ocfg, dense, psd, info, student, teacher = _setup(dtype, drop_depth=())
    lat, noise, t, ehs, empty = _inputs()
    ac = ostep.alphas_cumprod()
    P = {k_: v.clone().requires_grad_(True) for k_, v in psd.items()}
    loss, _, dist_, blk, _ = ostep.upper_step_loss((P, info), (dense, oweights.dense_info(ocfg)), ocfg, ac, lat, noise, t,
                                                   ehs, empty, w_block=wb)
    loss.backward()
    st = BilevelStepper(student, teacher, up_w_block=wb)
    L = st.upper_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), empty.cuda())
    tot, _, s, b = st.total(L, upper=True)
    ltol = 2e-4 if dn == "f32" else 3e-2
    assert abs(tot - loss.item()) <= ltol * abs(loss.item()), (tot, loss.item())
    if wb > 0:
        assert blk.item() > 0 and abs(b - blk.item()) <= ltol * blk.item(), (b, blk.item())
    grads = student.store.state_dict(arena=student.store.grad)
    if dn == "f32":
        worst = max((_rel(grads[n], p.grad), n) for n, p in P.items())
        assert worst[0] < 2e-3, worst
    else:
        big = [n for n, p in P.items() if p.numel() >= 4096]
        cos = min(torch.nn.functional.cosine_similarity(grads[n].flatten(), P[n].grad.flatten(), dim=0).item() for n in big)
        assert cos > 0.98, cos


@pytest.mark.parametrize("dn,budget", [("bf16", 0.82)])    # budget 0.55 (+ gradients, upper step, 96^2): test_fullsize_parity_gpu.py
def test_full_size_sd21_main_step_matches_oracle(dev, dn, budget):
    """BASELINE.json configs[0] (and the 82 %-budget student of configs[3]): the REAL SD-2.1 topology (865.9 M-parameter
    dense teacher, MAC-budget-0.55 / 0.82 student),
    B=1, 64x64 latent, 77x1024 text states: main-step losses of the HIP engine vs the CPU oracle (fp32 engine 1e-3,
    bf16 engine 3e-2 relative - north_star: "loss curves matching the CPU reference to 1e-3" for the fp32 path)."""
    from pdm_ref import step as ostep, weights as oweights
    from pdm_ref.config import UNetConfig as OCfg
    from pdm.models.unet.spec import UNetConfig, arch_vector_for_budget
    from pdm.models.unet.unet_2d_conditional import UNet2DConditionModelPruned
    from pdm.training.bilevel import BilevelStepper
    dtype = torch.float32 if dn == "f32" else torch.bfloat16
    ocfg, cfg = OCfg.sd21(), UNetConfig.sd21()
    av, ratio, _ = arch_vector_for_budget(cfg, budget)
    teacher = UNet2DConditionModelPruned(cfg, None, "cuda:0", dtype, train=False, seed=0)
    dense = teacher.state_dict()                      # the oracle gets exactly the weights the engine holds
    student = UNet2DConditionModelPruned(cfg, av, "cuda:0", dtype, train=True, init=False)
    student.load_dense_or_pruned(dense)
    psd, info = oweights.prune_state_dict(dense, ocfg, av)
    mine = student.state_dict()
    assert set(mine) == set(psd) and all(mine[k_].shape == psd[k_].shape for k_ in psd)
    g = torch.Generator().manual_seed(43)
    lat, noise = torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 4, 64, 64, generator=g)
    t, ehs = torch.tensor([431]), torch.randn(1, 77, 1024, generator=g)
    with torch.no_grad():
        loss, diff, dist_, block, _ = ostep.main_step_loss((psd, info), (dense, oweights.dense_info(ocfg)), ocfg,
                                                           ostep.alphas_cumprod(), lat, noise, t, ehs)
    st = BilevelStepper(student, teacher)
    tot, d, s, b = st.total(st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), backward=(dn == "bf16")))
    tol = 1e-3 if dn == "f32" else 3e-2
    for name, got, ref in (("diff", d, diff.item()), ("dist", s, dist_.item()), ("block", b, block.item()),
                           ("total", tot, loss.item())):
        assert abs(got - ref) <= tol * max(abs(ref), 1e-3), (name, got, ref, ratio)
    if dn == "bf16":        # the backward ran: gradients are finite and the packed padding carries none
        gsum = float(student.store.grad.double().abs().sum())
        assert math.isfinite(gsum) and gsum > 0


def test_side_stream_wgrad_matches_in_stream(dev):
    """engine.wgrad_async moves the weight-gradient GEMMs to a second HIP stream (eager multi-GPU mode); the gradients
    must not depend on it (regression test for a read-after-alias race on residual layers)."""
    from pdm.training.bilevel import BilevelStepper
    ocfg, dense, psd, info, student, teacher = _setup(torch.float32)
    lat, noise, t, ehs, _ = _inputs()
    st = BilevelStepper(student, teacher)
    grads = []
    for mode in (False, True, True):
        student.engine.wgrad_async = mode
        student.store.grad.zero_()
        st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda())
        torch.cuda.synchronize()
        grads.append(student.store.state_dict(arena=student.store.grad))
    for other in grads[1:]:
        worst = max((_rel(other[n], grads[0][n]), n) for n in grads[0])
        assert worst[0] < 1e-4, worst


def test_segmented_graph_replay_matches_eager(dev):
    """GraphedBilevel cuts the backward into several hipGraphs (where the multi-GPU all-reduce buckets are issued); the
    replayed gradients must equal the eager ones, with one graph and with four."""
    from pdm.training.bilevel import BilevelStepper, GraphedBilevel
    ocfg, dense, psd, info, student, teacher = _setup(torch.bfloat16)
    lat, noise, t, ehs, _ = _inputs()
    st = BilevelStepper(student, teacher)
    lat, noise, t, ehs = lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda()
    st.main_step(lat, noise, t, ehs)
    torch.cuda.synchronize()
    ref = student.store.grad.clone()
    ref_loss = st.losses.clone()
    for nseg in (1, 4):
        g = GraphedBilevel(st, 2, 4, 16, 16, 13, 64, segments=nseg, stream_opt=False)
        g.force_segments = nseg > 1
        g.capture(bilevel=False)
        assert len(g.g_main.bwd) == nseg and (g.g_main.teacher is None) == st.lockstep, (len(g.g_main.bwd), g.g_main.offs)
        student.store.grad.zero_()
        g._load(lat, noise, t, ehs)
        g._replay_step(g.g_main)
        torch.cuda.synchronize()
        assert torch.allclose(st.losses, ref_loss, rtol=5e-3, atol=1e-6)       # split-K atomics: not bit-reproducible
        rel = (student.store.grad - ref).abs().max().item() / ref.abs().max().item()
        assert rel < 2e-2, (nseg, rel)       # split-K atomics order differs between launches; bf16 path


@pytest.mark.parametrize("forced", [False, True])
def test_streamed_adamw_matches_step_then_optimizer(dev, forced):
    """Graph mode applies AdamW to every finished sixth of the gradient arena while the backward is still running (and,
    with several ranks, behind that share's all-reduce); parameters and moments after two iterations must equal
    backward-then-optimiser.  forced=True exercises the multi-graph (N > 1) replay path on one rank."""
    from pdm.training.bilevel import BilevelStepper, GraphedBilevel
    lat, noise, t, ehs, _ = _inputs()
    lat, noise, t, ehs = lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda()
    results = []
    for mode in ("eager", "graph"):
        ocfg, dense, psd, info, student, teacher = _setup(torch.bfloat16)
        st = BilevelStepper(student, teacher, lr=1e-3, upper_lr=1e-3)
        if mode == "graph":
            g = GraphedBilevel(st, 2, 4, 16, 16, 13, 64, segments=4, stream_opt=True)
            g.force_segments = forced
            g.capture(bilevel=False)
        for _ in range(2):
            if mode == "eager":
                st.main_step(lat, noise, t, ehs)
                st.optimizer_step()
            else:
                g.main(lat, noise, t, ehs)
        torch.cuda.synchronize()
        assert float(student.store.grad.abs().max()) == 0.0
        results.append((student.store.master.clone(), st.opt.m.clone(), student.store.w.float().clone(),
                        student.store.wt.float().clone()))
    # Adam's first steps move every weight by ~lr whatever the gradient size, so a near-zero gradient whose sign flips
    # with the split-K atomics order moves a weight by 2 lr per step: bound the worst element by that, and the mean tightly
    for a, b, what in zip(results[0], results[1], ("master", "exp_avg", "bf16 copy", "transposed copy")):
        d = (a - b).abs()
        if what == "exp_avg":
            assert d.max().item() <= 2e-2 * a.abs().max().item(), (what, d.max().item())
        else:
            assert d.max().item() <= 4.2e-3 + 1e-2 * a.abs().max().item() * (what != "master"), (what, d.max().item())
        assert d.mean().item() <= 2e-3 * a.abs().mean().item() + 1e-6, (what, d.mean().item(), a.abs().mean().item())


def test_graph_replay_with_teacher_graph_matches_eager_and_survives_recapture(dev):
    """The graph mode of round 3: every captured graph is single-stream (teacher, student forward, loss heads + backward
    segments), the teacher graph replays on the teacher stream beside the student forward, the AdamW of every finished
    share on the opt stream.  Four bilevel iterations on four different batches (one upper step in between) must give the
    eager mode's losses and parameters.  Then the sequence a training job with a ragged last batch walks - capture B = 2,
    replay, capture B = 1, replay, replay B = 2 again, close B = 1, capture a THIRD executor set, replay - all in this one
    process: the pattern that crashed hipGraphLaunch (hip::Graph::UpdateStreams) while graphs had parallel branches."""
    from pdm.training.bilevel import BilevelStepper, GraphedBilevel
    g = torch.Generator().manual_seed(5)
    batches = [tuple(x.cuda() for x in (torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 4, 16, 16, generator=g),
                                        torch.randint(0, 1000, (2,), generator=g), torch.randn(2, 13, 64, generator=g)))
               for _ in range(4)]
    empty = torch.randn(1, 13, 64, generator=g).expand(2, 13, 64).contiguous().cuda()
    results = []
    for mode in ("eager", "graph"):
        ocfg, dense, psd, info, student, teacher = _setup(torch.float32)
        st = BilevelStepper(student, teacher, lr=1e-4, upper_lr=1e-4, bilevel=True)
        if mode == "graph":
            gr = GraphedBilevel(st, 2, 4, 16, 16, 13, 64, segments=3)
            gr.capture(bilevel=True)
            assert not st.lockstep and gr.g_main.teacher is not None and gr.g_upper.teacher is not None and len(gr.g_main.bwd) >= 2
        losses = []
        for i, b in enumerate(batches):
            if mode == "graph":
                gr.main(*b)
            else:
                st.main_step(*b)
                st.optimizer_step()
            losses.append(st.losses.clone())
            if i == 1:
                if mode == "graph":
                    gr.upper(*b, empty)
                else:
                    st.upper_step(*b, empty)
                    st.optimizer_step(upper=True)
        torch.cuda.synchronize()
        results.append((torch.stack(losses).cpu(), student.store.master.clone()))
    assert torch.allclose(results[1][0], results[0][0], rtol=1e-5, atol=1e-9), (results[1][0], results[0][0])
    d, dr = (results[1][1] - results[0][1]).abs().max().item(), results[0][1].abs().max().item()
    assert d <= 1e-5 * dr + 8e-4, d          # Adam turns round-off-sized gradients into +-lr steps (see the DP test)
    # ---- shape change -> second executor set -> back -> eviction -> third set, in this process
    one = tuple(x[:1].contiguous() for x in batches[0])
    g1 = GraphedBilevel(st, 1, 4, 16, 16, 13, 64, segments=3)
    g1.capture(bilevel=True)
    g1.main(*one)
    g1.upper(*one, empty[:1].contiguous())
    gr.main(*batches[1])
    g1.main(*one)
    g1.close()
    g3 = GraphedBilevel(st, 2, 4, 16, 16, 13, 64, segments=2)
    g3.capture(bilevel=True)
    g3.main(*batches[2])
    gr.main(*batches[3])
    g3.upper(*batches[2], empty)
    torch.cuda.synchronize()
    assert torch.isfinite(st.losses).all() and torch.isfinite(student.store.master).all()
    for x in (gr, g3):
        x.close()


def test_teacher_prefetch_gives_the_plain_graph_replay(dev):
    """Cross-step teacher prefetch (GraphedBilevel(prefetch=True)): the frozen teacher's pass over batch t+1 is replayed on the
    teacher stream behind the loss heads of step t (a graph of their own in this mode), from the teacher graph's OWN static inputs;
    an upper step's 2B teacher pass is announced to the main step in front of it.  Six bilevel iterations over four batches (upper
    step after the second and the fifth; the fourth main step is NOT announced, the sixth is announced under a token the caller then
    does not present: both run their teacher in line) must give the losses and parameters of the plain replay; the hand-overs are
    counted."""
    from pdm.training.bilevel import BilevelStepper, GraphedBilevel
    g = torch.Generator().manual_seed(11)
    batches = [tuple(x.cuda() for x in (torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 4, 16, 16, generator=g),
                                        torch.randint(0, 1000, (2,), generator=g), torch.randn(2, 13, 64, generator=g)))
               for _ in range(4)]
    empty = torch.randn(1, 13, 64, generator=g).expand(2, 13, 64).contiguous().cuda()
    results = []
    for pre in (False, True):
        ocfg, dense, psd, info, student, teacher = _setup(torch.float32)
        st = BilevelStepper(student, teacher, lr=1e-4, upper_lr=1e-4, bilevel=True)
        gr = GraphedBilevel(st, 2, 4, 16, 16, 13, 64, segments=3, prefetch=pre)
        gr.capture(bilevel=True)
        assert gr.prefetch == pre
        losses = []
        for i in range(6):
            b, nb_ = batches[i % 4], batches[(i + 1) % 4]
            token = i if i != 5 else "someone else's"          # step 5 finds a pass queued under another name ...
            nxt = dict(next_batch=nb_, next_id=i + 1) if i != 2 else {}        # ... and step 3 nothing queued for it
            if i in (1, 4):          # an upper step follows: ITS teacher pass is announced to the main step, the next main batch to it
                gr.main(*b, batch_id=token, next_upper=b + (empty,), upper_id=("u", i))
                losses.append(st.losses.clone())
                gr.upper(*b, empty, batch_id=("u", i), **nxt)
            else:
                gr.main(*b, batch_id=token, **nxt)
            losses.append(st.losses.clone())
        torch.cuda.synchronize()
        assert gr.prefetch_hits == (5 if pre else 0), gr.prefetch_hits      # main steps 1, 2 and 4, both upper steps
        assert (gr.g_main.loss is not None) == pre and (gr.g_upper.loss is not None) == pre
        results.append((torch.stack(losses).cpu(), student.store.master.clone()))
        gr.close()
    assert torch.allclose(results[1][0], results[0][0], rtol=1e-5, atol=1e-9), (results[1][0], results[0][0])
    d, dr = (results[1][1] - results[0][1]).abs().max().item(), results[0][1].abs().max().item()
    assert d <= 1e-5 * dr + 8e-4, d          # Adam turns round-off-sized gradients into +-lr steps (see the DP test)


@pytest.mark.parametrize("dn,tol", [("f32", 1e-3), ("bf16", 2e-2)])
def test_bilevel_loss_curve_matches_oracle(dev, dn, tol):
    """north_star: "loss curves matching the CPU reference to 1e-3".  Nine bilevel iterations on the tiny topology - main
    step + AdamW every iteration, upper (concept-suppression) step + its own AdamW every third, fresh seeded (latent,
    noise, timestep, prompt) inputs per iteration - against the same loop on the fp32 CPU oracle (autograd + the oracle's
    AdamW restatement).  fp32 engine: every loss of both curves agrees to 1e-3 relative.  bf16 engine (the benchmarked
    dtype: bf16 weights / activations, fp32 accumulation, statistics, master weights and optimiser): 2e-2 - one bf16
    rounding is 2^-9 = 2e-3 relative and a loss is a mean of squared O(1) differences behind ~100 rounded layers, so the
    curve cannot track an fp32 reference to 1e-3; what it must not do is DRIFT: the bound is the same at iteration 9 as
    at iteration 1 (fp32 master weights, so rounding does not accumulate in the parameters)."""
    from pdm.training.bilevel import BilevelStepper
    ocfg, dense, psd, info, student, teacher = _setup(torch.float32 if dn == "f32" else torch.bfloat16, drop_depth=(1, 9))
    lr, ulr = 2e-5, 5e-5
    ref_curve = _oracle_curve(ocfg, dense, psd, info, lr, ulr)
    st = BilevelStepper(student, teacher, lr=lr, upper_lr=ulr, bilevel=True)
    curve = []
    for it, (lat, noise, t, ehs, empty) in enumerate(_curve_inputs()):
        for name in ["main"] + (["upper"] if (it + 1) % 3 == 0 else []):
            if name == "main":
                L = st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda())
            else:
                L = st.upper_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), empty.cuda())
            st.optimizer_step(upper=name == "upper")
            curve.append((name, st.total(L, upper=name == "upper")[0]))
    assert len(curve) == 12 == len(ref_curve)
    worst = max(abs(got - ref) / max(abs(ref), 1e-6) for (_, got), ref in zip(curve, ref_curve))
    assert worst <= tol, [(n, round(got, 6), round(ref, 6)) for (n, got), ref in zip(curve, ref_curve)]
    # the curves must actually move (the optimisers are live): last main loss differs from a frozen-weights evaluation
    new = student.state_dict()
    drift = max((new[n] - psd[n]).abs().max().item() for n in psd)
    assert drift > 5e-5


def _curve_inputs():
    g = torch.Generator().manual_seed(7)
    out = []
    for _ in range(9):
        lat, noise = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 4, 16, 16, generator=g)
        t = torch.randint(0, 1000, (2,), generator=g)
        ehs = torch.randn(2, 13, 64, generator=g)
        empty = torch.randn(1, 13, 64, generator=g).expand(2, 13, 64).contiguous()
        out.append((lat, noise, t, ehs, empty))
    return out


_CURVE = {}


def _oracle_curve(ocfg, dense, psd, info, lr, ulr, mixed=False):
    """The same 9-iteration bilevel loop on the CPU oracle (autograd + its AdamW restatement); computed once per session
    (it does not depend on the engine dtype under test).  mixed: the oracle in the reference's `--mixed_precision bf16`
    numerics (bf16-cast teacher, student forward under autocast, fp32 master weights and optimiser): True = torch's CPU
    autocast, "cuda" = CUDA autocast's op policy (fp32 norms, fused-SDPA numerics; pdm_ref/step.py _mixed)."""
    key = ("mixed-" + str(mixed)) if mixed else "ref"
    if key in _CURVE:
        return _CURVE[key]
    from pdm_ref import step as ostep, weights as oweights
    ac = ostep.alphas_cumprod()
    P = {k_: v.clone() for k_, v in psd.items()}
    mom = [({k_: torch.zeros_like(v) for k_, v in P.items()}, {k_: torch.zeros_like(v) for k_, v in P.items()}) for _ in range(2)]
    tinfo = oweights.dense_info(ocfg)
    steps, ref = [0, 0], []
    for it, (lat, noise, t, ehs, empty) in enumerate(_curve_inputs()):
        for name, oi in [("main", 0)] + ([("upper", 1)] if (it + 1) % 3 == 0 else []):
            Pg = {k_: v.clone().requires_grad_(True) for k_, v in P.items()}
            if name == "main":
                loss = ostep.main_step_loss((Pg, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs, mixed=mixed)[0]
            else:
                loss = ostep.upper_step_loss((Pg, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs, empty, mixed=mixed)[0]
            loss.float().backward()
            steps[oi] += 1
            ostep.adamw_step(P, {k_: v.grad for k_, v in Pg.items()}, mom[oi][0], mom[oi][1], steps[oi], lr if oi == 0 else ulr)
            ref.append(loss.item())
    _CURVE[key] = ref
    return ref


def _l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_bf16_error_at_fixed_weights_is_below_the_reference_policys(dev):
    """Where the bf16 engine's distance from the fp32 oracle comes from, measured where it can be measured without chaos: at
    FIXED weights (no optimiser in the loop), as relative L2 errors over whole tensors.  The comparator is the oracle evaluated
    the way the reference trains under `--mixed_precision bf16` (CUDA autocast's op policy: bf16 conv / linear / attention
    products, fp32 norms and softmax, fp32 master weights - pdm_ref/step.py mixed="cuda", trainer.py:516-527).  Every hooked
    block activation, the prediction, the teacher's prediction and the whole main-step gradient of the HIP engine must be no
    further from fp32 than 1.15 x that evaluation (measured round 4, tools/bf16_bisect.py: 0.91-0.95 x on every activation,
    0.82 x on the teacher, 0.885 x on the gradient, every layer kind below 1) - i.e. no op class of the engine keeps less
    precision than autocast does (epilogues add bias / residual in fp32 before the one rounding, weight gradients stay fp32)."""
    from pdm_ref import step as ostep, unet as ounet, weights as oweights
    from pdm.training.bilevel import BilevelStepper
    ocfg, dense, psd, info, student, teacher = _setup(torch.bfloat16, drop_depth=(1, 9))
    lat, noise, t, ehs, _ = _inputs()
    ac = ostep.alphas_cumprod()
    tinfo = oweights.dense_info(ocfg)
    noisy = ostep.add_noise(ac, lat, noise, t)
    ref, pol = {}, {}
    for store, mixed in ((ref, False), (pol, "cuda")):
        mp = ostep._mixed(mixed)
        acts = {}
        with torch.no_grad(), mp.ctx():
            store["pred"] = ounet.unet_forward(psd, ocfg, info, noisy, t, ehs, acts).float()
            store["teacher"] = ounet.unet_forward(mp.teacher_sd(dense), ocfg, tinfo, noisy, t, ehs).float()
        store.update({k_: v.float() for k_, v in acts.items()})
        P = {k_: v.clone().requires_grad_(True) for k_, v in psd.items()}
        ostep.main_step_loss((P, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs, mixed=mixed)[0].float().backward()
        store["grad"] = torch.cat([P[k_].grad.float().flatten() for k_ in sorted(P)])
    acts = {}
    for i, h in enumerate(student.down_blocks):
        h.register_forward_hook(lambda m, inp, out, i=i: acts.__setitem__(f"d{i}", out[0]))
    student.mid_block.register_forward_hook(lambda m, inp, out: acts.__setitem__("m", out))
    for i, h in enumerate(student.up_blocks):
        h.register_forward_hook(lambda m, inp, out, i=i: acts.__setitem__(f"u{i}", out))
    hip = {"pred": student.eval()(noisy, t, ehs).sample.float().cpu(), "teacher": teacher(noisy, t, ehs).sample.float().cpu()}
    hip.update({k_: v.float().cpu() for k_, v in acts.items()})
    student.train()
    st = BilevelStepper(student, teacher)
    st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda())
    torch.cuda.synchronize()
    g = student.store.state_dict(arena=student.store.grad)
    hip["grad"] = torch.cat([g[k_].float().cpu().flatten() for k_ in sorted(g)])
    report, bad = {}, []
    for key in list(ostep.BLOCK_KEYS) + ["pred", "teacher", "grad"]:
        e_hip, e_pol = _l2(hip[key], ref[key]), _l2(pol[key], ref[key])
        report[key] = (round(e_hip, 5), round(e_pol, 5))
        assert e_pol > 1e-3, (key, e_pol)                       # the comparator really computes in bf16
        if e_hip > 1.15 * e_pol:
            bad.append(key)
    assert not bad, (bad, report)


def test_bf16_curve_error_is_the_dtype_not_the_kernels(dev):
    """north_star asks for loss curves within 1e-3 of the CPU reference; the fp32 engine meets that (test above), the bf16
    engine cannot - and neither can the reference itself under `--mixed_precision bf16`.  What a bf16 curve's distance from the
    fp32 curve IS, is measured on the oracle: evaluated under the two op policies torch has for bf16 autocast (CPU autocast;
    CUDA autocast = the one the reference trains under, pdm_ref/step.py _mixed) the SAME 12-point loop ends max 1.8e-3 / RMS
    0.7e-3 and max 5.3e-3 / RMS 1.6e-3 away from fp32, and the same policy on two hosts differs point by point by factors of
    0.3-8 (round 4, tools/bf16_bisect.py).  The distance is chaotic, not systematic: the first Adam steps move every weight by
    +-lr according to the SIGN of its gradient, so rounding noise on small gradients becomes full-size weight steps, and the three
    upper-step points (loss ~0.07, a difference of O(1) terms computed without the .float() cast, trainer.py:2994-2999) carry most
    of it.  The systematic part is pinned by test_bf16_error_at_fixed_weights_is_below_the_reference_policys; here the HIP
    curve must stay within 1.5 x the envelope of the two oracle evaluations, in the maximum and in the RMS over the 12 points.
    Round 3 compared with the CPU-autocast evaluation alone, measured 1.6 x on one box and widened the bound to 2.0 x; the same
    comparison measured 0.70 x in round 4 (HIP max 1.3e-3 / RMS 0.55e-3) - the ratio to ONE evaluation is not a property of the engine."""
    from pdm.training.bilevel import BilevelStepper
    ocfg, dense, psd, info, student, teacher = _setup(torch.bfloat16, drop_depth=(1, 9))
    lr, ulr = 2e-5, 5e-5
    ref = _oracle_curve(ocfg, dense, psd, info, lr, ulr)
    mixes = [_oracle_curve(ocfg, dense, psd, info, lr, ulr, mixed=m) for m in (True, "cuda")]
    st = BilevelStepper(student, teacher, lr=lr, upper_lr=ulr, bilevel=True)
    hip = []
    for it, (lat, noise, t, ehs, empty) in enumerate(_curve_inputs()):
        for name in ["main"] + (["upper"] if (it + 1) % 3 == 0 else []):
            if name == "main":
                L = st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda())
            else:
                L = st.upper_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), empty.cuda())
            st.optimizer_step(upper=name == "upper")
            hip.append(st.total(L, upper=name == "upper")[0])
    rel = lambda a: [abs(x - r) / max(abs(r), 1e-6) for x, r in zip(a, ref)]
    rms = lambda d: (sum(x * x for x in d) / len(d)) ** 0.5
    d_mix, d_hip = [rel(m) for m in mixes], rel(hip)
    assert all(max(d) > 1e-4 for d in d_mix), d_mix            # the mixed oracles really compute in bf16
    env_max, env_rms = max(max(d) for d in d_mix), max(rms(d) for d in d_mix)
    report = [tuple(round(x, 5) for x in p) for p in zip(d_hip, *d_mix)]
    assert max(d_hip) <= 1.5 * env_max, (max(d_hip), env_max, report)
    assert rms(d_hip) <= 1.5 * env_rms, (rms(d_hip), env_rms, report)
    assert max(d_hip) <= 1e-2                                    # and in absolute terms: a few times the fp32 engine's 1e-3


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_fp8_e4m3_attention_precision_matches_the_oracle(dev, dn):
    """BASELINE.json configs[4] names an "fp8 MFMA attention path"; the reference has no fp8 code (its attention runs in the
    activation dtype, blocks.py:257-277), so the option is defined here and restated in the oracle the same way: with
    `attention_precision = "fp8_e4m3"` Q, K and V of every self- and cross-attention are rounded to the nearest e4m3fn value
    (pdmk_quantize_e4m3; gradients straight through) before QK^T and PV, forward and backward alike.  Main and upper step of the tiny
    topology, losses and every gradient against the oracle evaluated with the same rounding - and the option must be live: the
    losses move away from the un-rounded oracle by more than the comparison tolerance."""
    from pdm_ref import step as ostep, unet as ounet, weights as oweights
    from pdm.training.bilevel import BilevelStepper
    dtype = torch.float32 if dn == "f32" else torch.bfloat16
    ocfg, dense, psd, info, student, teacher = _setup(dtype)
    lat, noise, t, ehs, empty = _inputs()
    ac = ostep.alphas_cumprod()
    tinfo = oweights.dense_info(ocfg)
    plain = float(ostep.main_step_loss((psd, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs)[0])
    ounet.ATTN_FP8 = True
    try:
        P = {k_: v.clone().requires_grad_(True) for k_, v in psd.items()}
        loss, diff, dist_, block, _ = ostep.main_step_loss((P, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs)
        loss.backward()
        loss, diff, dist_, block = loss.detach(), diff.detach(), dist_.detach(), block.detach()
        uloss = float(ostep.upper_step_loss((psd, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs, empty)[0])
    finally:
        ounet.ATTN_FP8 = False
    for m_ in (student, teacher):
        m_.set_attention_precision("fp8_e4m3")
    st = BilevelStepper(student, teacher)
    tot, d, s, b = st.total(st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda()))
    ltol = 5e-4 if dn == "f32" else 3e-2
    for name, got, ref in (("diff", d, float(diff)), ("dist", s, float(dist_)), ("block", b, float(block)), ("total", tot, float(loss))):
        assert abs(got - ref) <= ltol * max(abs(ref), 1e-3), (name, got, ref)
    if dn == "f32":          # the option is live: the rounding moves the loss, and the engine lands on the rounded value, not the plain one
        assert abs(plain - float(loss)) > 1e-4 * abs(plain), (plain, float(loss))
        assert abs(tot - float(loss)) < 0.1 * abs(plain - float(loss)), (tot, float(loss), plain)
    grads = student.store.state_dict(arena=student.store.grad)
    if dn == "f32":
        # Rounding is discontinuous: an element of Q/K/V within the two implementations' fp32 difference (5e-6, the plain test) of
        # a tie lands one grid step (6-12 %) away.  The oracle ALONE, under a 2e-6 relative perturbation of its weights, moves its
        # to_k / to_q gradients by up to 2.4e-2 (median over tensors 9e-4; 8e-5 / 2.5e-5 without the rounding) -
        # tools/fp8_selfnoise.py.  So: per tensor inside that envelope, the median at the flip-free level, and - the part that
        # shows the rounding is applied where the oracle applies it - the q/k projections nearer the rounded oracle than the plain one.
        errs = {n: _rel(grads[n], p.grad) for n, p in P.items()}
        worst = max(errs.values())
        med = sorted(errs.values())[len(errs) // 2]
        assert worst < 8e-2 and med < 3e-3, (worst, med, max(errs, key=errs.get))
        Pp = {k_: v.clone().requires_grad_(True) for k_, v in psd.items()}
        ostep.main_step_loss((Pp, info), (dense, tinfo), ocfg, ac, lat, noise, t, ehs)[0].backward()
        qk = [n for n in P if n.endswith(("to_q.weight", "to_k.weight"))]
        to_rounded = sum(errs[n] for n in qk)
        to_plain = sum(_rel(grads[n], Pp[n].grad) for n in qk)
        assert to_rounded < 0.25 * to_plain, (to_rounded, to_plain)
    else:
        bad = []
        for name, p in P.items():
            if p.numel() >= 1024:
                cos = torch.nn.functional.cosine_similarity(grads[name].flatten(), p.grad.flatten(), dim=0).item()
                if cos < 0.98:
                    bad.append((name, cos))
        assert not bad, bad[:8]
    k_ = __import__("pdm._pdmk", fromlist=["x"])
    k_.zero_(student.store.grad)
    ut = st.total(st.upper_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), empty.cuda()), upper=True)[0]
    assert abs(ut - uloss) <= ltol * abs(uloss), (ut, uloss)


def test_deferred_wt_refresh_is_complete_before_backward(dev):
    """The transposed (dgrad) weight copies are refreshed on a side stream at the start of the NEXT training step, not by
    the optimiser: after an optimiser step they are stale, and by the end of the next step they equal a fresh transpose of
    the updated weights (the backward waited for the side stream)."""
    from pdm.training.bilevel import BilevelStepper
    ocfg, dense, psd, info, student, teacher = _setup(torch.bfloat16)
    lat, noise, t, ehs, empty = _inputs()
    store = student.store
    st = BilevelStepper(student, teacher, lr=1e-2, upper_lr=1e-2, bilevel=True)
    assert store.defer_wt
    st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda())
    st.optimizer_step()
    torch.cuda.synchronize()
    stale = store.wt.clone()
    for step_fn in (lambda: st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda()),
                    lambda: st.upper_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda(), empty.cuda())):
        step_fn()
        torch.cuda.synchronize()
        got = store.wt.clone()
        store.refresh_wt()
        torch.cuda.synchronize()
        assert torch.equal(got, store.wt)
        st.optimizer_step(upper=False)
        torch.cuda.synchronize()
    assert not torch.equal(stale, store.wt)
    # and the gradients of a deferred step equal those of a step whose copies were refreshed with the optimiser
    g_def = store.grad.clone()
    store.defer_wt = False
    store.refresh_wt()
    k_ = __import__("pdm._pdmk", fromlist=["x"])
    k_.zero_(store.grad)
    st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda())
    torch.cuda.synchronize()
    g_now = store.grad.clone()
    store.defer_wt = True
    k_.zero_(store.grad)
    st.main_step(lat.cuda(), noise.cuda(), t.cuda(), ehs.cuda())
    torch.cuda.synchronize()
    assert torch.allclose(store.grad, g_now, rtol=0, atol=1e-6 * float(g_now.abs().max()) + 1e-12) or \
        torch.nn.functional.cosine_similarity(store.grad, g_now, dim=0).item() > 0.99999


def test_lockstep_forward_matches_two_stream_forward(dev):
    """PDMK_LOCKSTEP=1: teacher pass and student forward recorded and issued side by side on one stream, layer pairs through
    pdmk_gemm_group (bit-identical to separate launches by the kernel tests); losses, gradients and the parameter update of a
    main and an upper step equal the two-stream mode's, eager and as graph replay."""
    from pdm import _pdmk as k
    from pdm.training.bilevel import BilevelStepper, GraphedBilevel
    lat, noise, t, ehs, empty = (x.cuda() for x in _inputs())
    res = []
    for mode in ("streams", "lockstep", "lockstep_graph"):
        ocfg, dense, psd, info, student, teacher = _setup(torch.bfloat16, drop_depth=(1, 9))
        st = BilevelStepper(student, teacher, lr=1e-3, upper_lr=1e-3)
        st.lockstep = mode != "streams"
        if mode == "lockstep_graph":
            g = GraphedBilevel(st, 2, 4, 16, 16, 13, 64, segments=2)
            g.capture(bilevel=True)
            assert g.g_main.teacher is None and g.g_upper.teacher is None          # one forward graph for both models
            g._load(lat, noise, t, ehs)
            g._replay_step(g.g_main, None)
        else:
            k.STATS.update(launches=0, grouped=0)
            st.main_step(lat, noise, t, ehs)
            if mode == "lockstep":
                assert k.STATS["launches"] > 50, k.STATS                            # the recorded path really ran
        torch.cuda.synchronize()
        grad, losses = student.store.grad.clone(), st.losses.clone()
        k.zero_(student.store.grad)
        if mode == "lockstep_graph":
            g._load(lat, noise, t, ehs, empty)
            g._replay_step(g.g_upper, None)
        else:
            st.upper_step(lat, noise, t, ehs, empty)
        torch.cuda.synchronize()
        res.append((losses, grad, st.losses.clone(), student.store.grad.clone()))
    for other in res[1:]:
        for a, b in zip(other, res[0]):
            assert torch.allclose(a.double(), b.double(), rtol=2e-2, atol=2e-2 * float(b.abs().max())), (a - b).abs().max()
