"""CPU: the oracle against the committed golden vectors, which were produced by the REFERENCE's own importable pieces
(oracle/validate_against_reference.py: pdm.utils.metric_utils.compute_snr, the vendored CompVis ResBlock /
SpatialTransformer / timestep_embedding / beta schedule)."""
import os

import numpy as np
import torch

from pdm_ref import arch, step, unet, weights
from pdm_ref.config import UNetConfig

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_twins.npz"))


def T(name):
    return torch.from_numpy(GOLD[name])


def test_snr_and_min_snr_weights_match_reference():
    ac = step.alphas_cumprod()
    t = T("snr_t")
    assert torch.allclose(step.compute_snr(ac, t), T("snr_ref"), rtol=1e-6, atol=0)
    assert torch.allclose(step.min_snr_weights(ac, t), T("minsnr_w_ref"), rtol=1e-6, atol=0)
    # known answers recorded in SURVEY.md 8c
    snr = step.compute_snr(ac, torch.tensor([0, 499, 999]))
    assert torch.allclose(snr, torch.tensor([1175.4, 0.38441, 0.0046819]), rtol=2e-4)


def test_schedule_and_timestep_embedding_match_ldm_twins():
    assert torch.allclose(step.alphas_cumprod().double(), T("alphas_cumprod_f64"), atol=1e-5)
    assert torch.allclose(unet.timestep_embedding(T("temb_t"), 320), T("temb_ref"), atol=1e-6)


def _sd(prefix):
    return {k[len(prefix):]: T(k) for k in GOLD.files if k.startswith(prefix)}


def test_resblock_matches_ldm_resblock():
    for cin, cout in ((64, 64), (96, 64)):
        tag = f"resblock_{cin}_{cout}"
        y = unet.resblock(_sd(tag + "_w_"), "r", T(tag + "_x"), T(tag + "_emb"), 32, 32)
        assert torch.allclose(y, T(tag + "_y"), atol=2e-5, rtol=1e-5)


def test_transformer_matches_ldm_spatial_transformer():
    y = unet.transformer2d(_sd("st_w_"), "a", T("st_x"), T("st_ctx"), 2, 2, 64, 32)
    assert torch.allclose(y, T("st_y"), atol=2e-5, rtol=1e-5)


def test_arch_vector_layout_and_param_count():
    cfg = UNetConfig.sd21()
    s = arch.structure(cfg)
    assert sum(map(sum, s["width"])) == 1606 and sum(map(sum, s["depth"])) == 14
    assert arch.hard_concrete(torch.tensor([[0.2, 0.5, 0.9]])).tolist() == [[0.0, 1.0, 1.0]]
    # SD-2.1 U-Net parameter count, without materialising the weights: shapes only
    n = 0
    tiny = UNetConfig.tiny()
    sd = weights.init_dense_state_dict(tiny, 0)
    psd, info = weights.prune_state_dict(sd, tiny, arch.random_arch_vector(tiny, 0.5, 0, drop_depth=(0, 13)))
    assert info["down_blocks.0.resnets.1"]["dropped"] and info["up_blocks.3.attentions.2"]["dropped"]
    assert not any(k.startswith("down_blocks.0.resnets.1.") for k in psd)
    lat, t, ehs = torch.randn(1, 4, 8, 8), torch.tensor([3]), torch.randn(1, 5, 64)
    out = unet.unet_forward(psd, tiny, info, lat, t, ehs)
    assert out.shape == lat.shape and torch.isfinite(out).all()


# ---------------------------------------------------------------- VAE encoder (oracle/validate_vae_against_reference.py)
VGOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_twin.npz"))


def test_vae_encoder_matches_compvis_encoder():
    from pdm_ref import vae
    for tag, cfg in (("tiny", vae.VAEConfig.tiny()),
                     ("mid", vae.VAEConfig(block_out_channels=(32, 64, 128, 128), layers_per_block=2))):
        sd = vae.init_state_dict(cfg, seed=7)
        x = torch.from_numpy(VGOLD[f"{tag}_x"])
        mom = vae.encode_moments(sd, cfg, x)
        ref = torch.from_numpy(VGOLD[f"{tag}_moments"])
        assert mom.shape == ref.shape and torch.allclose(mom, ref, atol=2e-5), tag
        z = vae.sample_latents(ref * 4.0, torch.from_numpy(VGOLD[f"{tag}_eps"]), 1.0)
        assert torch.allclose(z, torch.from_numpy(VGOLD[f"{tag}_z4"]), atol=1e-6), tag


def test_vae_downsample_is_bottom_right_padded_stride2():
    x, w, b = (torch.from_numpy(VGOLD[k]) for k in ("ds_x", "ds_w", "ds_b"))
    y = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    assert torch.allclose(y, torch.from_numpy(VGOLD["ds_y"]), atol=1e-6)


# ---------------------------------------------------------------- CLIP text encoder (oracle/validate_clip_against_transformers.py)
def test_clip_text_oracle_matches_transformers_outputs():
    from pdm_ref import clip_text
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_text_hf.npz"))
    for tag, cfg in (("tiny", clip_text.CLIPTextConfig.tiny()),
                     ("wide", clip_text.CLIPTextConfig(vocab_size=2000, hidden_size=1024, intermediate_size=4096,
                                                       num_hidden_layers=2, num_attention_heads=16))):
        sd = clip_text.init_state_dict(cfg, seed=5)
        out = clip_text.encode(sd, cfg, torch.from_numpy(gold[f"{tag}_ids"]))
        assert torch.allclose(out, torch.from_numpy(gold[f"{tag}_out"]), atol=2e-5), tag
    # causality: a token's state does not depend on later tokens
    cfg = clip_text.CLIPTextConfig.tiny()
    sd = clip_text.init_state_dict(cfg, seed=5)
    ids = torch.from_numpy(gold["tiny_ids"])[:1].clone()
    a = clip_text.encode(sd, cfg, ids)
    ids[0, 40:] = 7
    b = clip_text.encode(sd, cfg, ids)
    assert torch.allclose(a[0, :40], b[0, :40], atol=1e-6) and not torch.allclose(a[0, 40:], b[0, 40:], atol=1e-3)


# ---------------------------------------------------------------- PNDM restatement (parity unpinned: closed forms only)
def test_pndm_reduces_to_ddim_and_reproduces_constant_predictions():
    from pdm_ref.sampler import PNDM
    for pt in ("epsilon", "v_prediction"):
        s = PNDM(prediction_type=pt)
        s.set_timesteps(50)
        assert len(s.timesteps) == 51 and s.timesteps[0] == 981 and s.timesteps[1] == 961 and s.timesteps[2] == 961
        assert s.timesteps[-1] == 1
        g = torch.Generator().manual_seed(0)
        x, out = torch.randn(2, 4, 8, 8, generator=g).double(), torch.randn(2, 4, 8, 8, generator=g).double()
        t, pt_ = 601, 581
        a_t, a_p = s.ac[t], s.ac[pt_]
        eps = out if pt == "epsilon" else a_t.sqrt() * out + (1 - a_t).sqrt() * x
        x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        ddim = a_p.sqrt() * x0 + (1 - a_p).sqrt() * eps                   # deterministic DDIM step (eta = 0)
        assert torch.allclose(s.get_prev_sample(x, t, pt_, out), ddim, atol=1e-10)
        # constant prediction: every Adams-Bashforth combination returns it (weights sum to one), so the walk equals DDIM
        c = torch.randn(1, 4, 4, 4, generator=g)
        s.set_timesteps(10)
        lat, ref = torch.randn(1, 4, 4, 4, generator=g), None
        ref = lat.double()
        seen = []
        for i, tt in enumerate(s.timesteps):
            lat = s.step(c, tt, lat)
            seen.append(tt)
        # DDIM walk over the distinct timesteps 901, 801, ..., 1 -> -99 (final alpha)
        for tt in sorted(set(seen), reverse=True):
            ref = PNDM(prediction_type=pt).get_prev_sample(ref, tt, tt - 100, c.double())
        assert torch.allclose(lat.double(), ref, atol=1e-5), pt


# ------------------------------------------------------------------------------------------------------------------
# tests/golden/reference_pruning.json: outputs of the REFERENCE'S OWN method bodies (hypernet.py:100-150 classmethods and
# the four prune() methods of blocks.py), executed in the build container by oracle/pin_reference_pruning.py.
def _digest(t):
    import hashlib
    t = t.detach().to(torch.float32).contiguous()
    return [list(t.shape), hashlib.sha256(t.numpy().tobytes()).hexdigest()]


def _pruning_fixture():
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_pruning.json")))


def test_arch_vector_classmethods_match_reference_outputs():
    """HyperStructure.get_random_arch_vector / transform_arch_vector (incl. force_width_non_zero): the oracle's and the
    product's restatements reproduce the reference's outputs bit for bit on the same seeds."""
    from pdm.models.unet import spec
    fx = _pruning_fixture()["arch"]
    assert len(fx) == 8
    for rec in fx:
        ocfg = getattr(UNetConfig, rec["cfg"])()
        pcfg = getattr(spec.UNetConfig, rec["cfg"])()
        for structure in (arch.structure(ocfg), spec.gate_structure(pcfg)):
            torch.manual_seed(rec["seed"])                       # the reference draws from the global generator
            av = spec.get_random_arch_vector(rec["ratio"], structure)
            assert _digest(av) == rec["random_arch_vector"], (rec["cfg"], rec["seed"])
            assert int((av >= 0.5).sum()) == rec["kept"]
        n = rec["n"]
        g = torch.Generator().manual_seed(rec["transform_input_seed"])
        x = torch.rand(1, n, generator=g)
        structure = spec.gate_structure(pcfg)
        w1 = structure["width"][0][0] + (structure["width"][0][1] if len(structure["width"][0]) > 1 else structure["width"][1][0])
        x[0, :w1] = 0.3 * x[0, :w1]                              # first two gates entirely below the threshold
        for force in (False, True):
            want = rec[f"transform_force{int(force)}"]
            xin = spec.force_width_non_zero(x, structure) if force else x
            for name, tv in (("product", spec.transform_arch_vector(xin, structure)),
                             ("oracle", arch.transform_arch_vector(xin, ocfg))):
                assert len(tv["width"]) == want["nwidth"] and len(tv["depth"]) == want["ndepth"], name
                assert [_digest(w) for w in tv["width"]] == want["width"], (name, force)
                assert [_digest(d) for d in tv["depth"]] == want["depth"], (name, force)
        assert rec["transform_force1"]["first_elems"][0] != rec["transform_force0"]["first_elems"][0]    # the branch fired


def test_physical_pruning_matches_reference_prune_methods():
    """ResnetBlock2DWidth[Depth]Gated.prune, GatedAttention.prune, GEGLUGated.prune_gate + FeedForwardWidthGated.prune run
    on the tiny topology's seeded dense weights: the oracle's prune_state_dict yields bit-identical tensors (sha256), the
    same kept-head / kept-group counts and the same dropped blocks - and so does the product's slice_dense_state_dict."""
    from pdm.models.unet import spec
    from pdm.models.unet.unet_2d_conditional import slice_dense_state_dict
    cases = _pruning_fixture()["pruning"]
    assert len(cases) == 3
    ocfg, pcfg = UNetConfig.tiny(), spec.UNetConfig.tiny()
    for c in cases:
        dense = weights.init_dense_state_dict(ocfg, seed=c["dense_seed"])
        av = arch.random_arch_vector(ocfg, c["ratio"], seed=c["arch_seed"], drop_depth=tuple(c["drop_depth"]))
        assert _digest(av) == c["arch_vector"]
        psd, info = weights.prune_state_dict(dense, ocfg, av)
        prod = slice_dense_state_dict(dense, pcfg, spec.apply_arch_vector(pcfg, av))
        ref = c["tensors"]
        dropped = {k for k, v in ref.items() if v == "dropped"}
        assert dropped == {p for p, i in info.items() if i["dropped"]}
        assert len(dropped) == len(c["drop_depth"])
        n = 0
        for name, want in ref.items():
            if want == "dropped":
                assert not any(k.startswith(name + ".") for k in psd) and not any(k.startswith(name + ".") for k in prod)
            elif name.endswith(".heads"):
                blk, an = name.rsplit(".transformer_blocks.0.", 1)[0], name.split(".")[-2]
                assert info[blk]["heads1" if an == "attn1" else "heads2"] == want
            elif name.endswith(".num_groups"):
                assert info[name[:-len(".norm2.num_groups")]]["groups2"] == want
            else:
                assert _digest(psd[name]) == want, name
                assert _digest(prod[name]) == want, name
                n += 1
        assert n > 300


def test_loss_heads_match_reference_statements():
    """tests/golden/reference_loss_heads.npz: the loss-head statements of UnetFineTuner.step (trainer.py:2451-2488) and
    BilevelUnetFineTuner.upper_step (:2983-3001), executed from the reference's source on seeded tensors
    (oracle/pin_reference_loss_heads.py).  The oracle's heads give the same four scalars and the same gradients w.r.t. the
    student prediction and a hooked block activation, for three weight settings each (incl. snr_gamma = None, zero block /
    distillation weights, a non-zero upper block weight)."""
    G_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_loss_heads.npz"))
    tt = lambda n: torch.from_numpy(G_[n])
    keys = step.BLOCK_KEYS
    ac = step.alphas_cumprod()
    for case in range(3):
        p = f"main{case}_"
        gamma, wd, wb, ws = tt(p + "cfg").tolist()
        gamma = None if gamma != gamma else gamma
        pred = tt(p + "pred").requires_grad_(True)
        acts_s = {k_: tt(p + "as_" + k_).requires_grad_(k_ == "m") for k_ in keys}
        acts_t = {k_: tt(p + "at_" + k_) for k_ in keys}
        out = step.main_loss_heads(pred, tt(p + "target"), tt(p + "full"), acts_s, acts_t, ac, tt(p + "t"), wd, wb, ws, gamma)
        out[0].backward()
        assert torch.allclose(torch.stack([x.detach().reshape(()) for x in out]), tt(p + "out"), rtol=1e-6, atol=1e-7), case
        assert torch.allclose(pred.grad, tt(p + "dpred"), rtol=1e-5, atol=1e-8)
        got = acts_s["m"].grad if acts_s["m"].grad is not None else torch.zeros_like(acts_s["m"])
        assert torch.allclose(got, tt(p + "dact_m"), rtol=1e-5, atol=1e-9)
        p = f"upper{case}_"
        ws, wb = tt(p + "cfg").tolist()
        pred = tt(p + "pred").requires_grad_(True)
        acts_s = {k_: tt(p + "as_" + k_).requires_grad_(k_ == "m") for k_ in keys}
        acts_t = {k_: tt(p + "at_" + k_) for k_ in keys}
        out = step.upper_loss_heads(pred, tt(p + "e_c"), tt(p + "e_u"), acts_s, acts_t, ws, wb)
        out[0].backward()
        assert torch.allclose(torch.stack([x.detach().reshape(()) for x in out]), tt(p + "out"), rtol=1e-6, atol=1e-7), case
        assert torch.allclose(pred.grad, tt(p + "dpred"), rtol=1e-5, atol=1e-8)
        got = acts_s["m"].grad if acts_s["m"].grad is not None else torch.zeros_like(acts_s["m"])
        assert torch.allclose(got, tt(p + "dact_m"), rtol=1e-5, atol=1e-9)


def test_oracle_e4m3_quantiser_is_torchs_float8_cast():
    """pdm_ref.unet.quant_e4m3 (the value grid of the "fp8_e4m3" attention precision, BASELINE.json configs[4]) against torch's own
    float8_e4m3fn cast: random values over five decades, exact ties (round-half-even), subnormals, the saturation edge, signed zeros;
    NaN stays NaN."""
    import torch
    from pdm_ref import unet as U
    g = torch.Generator().manual_seed(5)
    x = torch.cat([torch.randn(50000, generator=g) * 3, torch.randn(5000, generator=g) * 100, torch.randn(5000, generator=g) * 0.01,
                   torch.tensor([0.0, -0.0, 448.0, 447.9, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 2.5 * 2.0 ** -9, 0.0625, 15.5, 17.0, 18.0,
                                 19.0, 20.0, -17.0, -19.0, 2.0 ** -6, 0.9 * 2.0 ** -6])])
    q = U.quant_e4m3(x)
    assert torch.equal(q, x.to(torch.float8_e4m3fn).float())
    assert torch.equal(U.quant_e4m3(torch.tensor([449.0, 1e6, -1e6])), torch.tensor([448.0, 448.0, -448.0]))       # saturating
    assert torch.isnan(U.quant_e4m3(torch.tensor([float("nan")]))).all()
    xb = (torch.randn(4000, generator=g) * 4).bfloat16()
    qb = U.quant_e4m3(xb)
    assert qb.dtype == torch.bfloat16 and torch.equal(qb.float(), xb.float().to(torch.float8_e4m3fn).float())
    # straight-through gradient, and off by default
    U.ATTN_FP8 = True
    try:
        t = torch.randn(8, requires_grad=True)
        U._fq(t).sum().backward()
        assert torch.equal(t.grad, torch.ones(8))
    finally:
        U.ATTN_FP8 = False
    assert U._fq(x) is x
