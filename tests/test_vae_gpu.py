"""VAE encode in front of the step (-m gpu): libpdmk through the C ABI against the oracle
(oracle/pdm_ref/vae.py, pinned to the reference's vendored CompVis encoder) and against that twin's own outputs in
tests/golden/vae_twin.npz.  Tolerances as in test_kernels_gpu.py: fp32 2e-4, bf16 2e-2 of the output scale for single
kernels; whole-encoder bf16 5e-2 (about thirty bf16-rounded layers deep)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "bf16": torch.bfloat16}
TOL = {"f32": 2e-4, "bf16": 2e-2}
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_twin.npz"))


def close(got, ref, tol, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert math.isfinite(err) and err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def conv_w_pack(w):
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


@pytest.mark.parametrize("dn", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 32, 40, 10, 10), (1, 64, 160, 32, 32), (3, 32, 32, 4, 6)])
def test_conv_mode4_bottom_right_padded_stride2(dev, dn, shape):
    from pdm import _pdmk as k
    torch.manual_seed(1)
    dt = DT[dn]
    Bn, Ci, Co, Hs, Ws = shape
    x = torch.randn(Bn, Hs, Ws, Ci, device=dev).to(dt)
    w = (torch.randn(Co, Ci, 3, 3, device=dev) * (9 * Ci) ** -0.5).to(dt)
    bias = torch.randn(Co, device=dev)
    ref = F.conv2d(F.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1)), w.float(), bias, stride=2)
    Ho, Wo = Hs // 2, Ws // 2
    assert ref.shape[2:] == (Ho, Wo)
    C = torch.zeros(Bn * Ho * Wo, Co, device=dev, dtype=dt)
    k.gemm(x, conv_w_pack(w), C, Bn * Ho * Wo, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV,
           conv=(Bn, Hs, Ws, Ci, Ho, Wo, 4, Ci), bias=bias)
    close(C.view(Bn, Ho, Wo, Co), ref.permute(0, 2, 3, 1), TOL[dn], "conv mode 4")
    if dn == "bf16":          # every tile shape of the LDS-DMA ring kernels must implement the same gather
        for cand in range(12):
            os.environ["PDMK_RING_CFG"] = str(cand)
            try:
                C.zero_()
                k.gemm(x, conv_w_pack(w), C, Bn * Ho * Wo, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV,
                       conv=(Bn, Hs, Ws, Ci, Ho, Wo, 4, Ci), bias=bias)
            finally:
                del os.environ["PDMK_RING_CFG"]
            close(C.view(Bn, Ho, Wo, Co), ref.permute(0, 2, 3, 1), TOL[dn], f"conv mode 4, ring shape {cand}")


def test_conv_mode4_rejects_odd_sides_and_wgrad(dev):
    from pdm import _pdmk as k
    x = torch.zeros(1, 5, 6, 32, device=dev, dtype=torch.bfloat16)
    w = torch.zeros(32, 9 * 32, device=dev, dtype=torch.bfloat16)
    C = torch.zeros(2 * 3, 32, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        k.gemm(x, w, C, 6, 32, 288, 0, 288, 32, a_mode=k.A_CONV, conv=(1, 5, 6, 32, 3, 3, 4, 32))
    # forward only: the encoder is frozen, a mode-4 weight gradient is refused rather than silently computed as mode 1
    x2 = torch.zeros(1, 8, 8, 32, device=dev, dtype=torch.bfloat16)
    dy = torch.zeros(16, 32, device=dev, dtype=torch.bfloat16)
    dW = torch.zeros(32, 288, device=dev)
    with pytest.raises(RuntimeError):
        k.gemm(dy, x2, dW, 32, 288, 16, 32, 0, 288, a_mode=k.A_COLK, b_mode=k.B_COLK_CONV, out_f32=True,
               conv=(1, 8, 8, 32, 4, 4, 4, 32))


@pytest.mark.parametrize("dn", ["f32", "bf16"])
@pytest.mark.parametrize("rows,cols,lds", [(7, 64, 64), (130, 4096, 4096), (33, 1000, 1024), (5, 9000, 9000)])
def test_softmax_rows(dev, dn, rows, cols, lds):
    from pdm import _pdmk as k
    torch.manual_seed(2)
    s = torch.randn(rows, lds, device=dev) * 6.0
    s[0, 0] = 80.0                                    # a dominant logit: the row max must be subtracted
    p = torch.full((rows, cols + 8), 7.0, device=dev, dtype=DT[dn])
    k.softmax_rows(s, p, rows, cols, lds, cols + 8)
    ref = torch.softmax(s[:, :cols], dim=-1)
    close(p[:, :cols], ref, 1e-5 if dn == "f32" else 1e-2, "softmax rows")
    assert (p[:, cols:] == 7.0).all(), "columns beyond `cols` must stay untouched"
    if dn == "f32":
        assert (p[:, :cols].sum(-1) - 1).abs().max().item() < 1e-5


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_latent_sample_matches_oracle(dev, dn):
    from pdm import _pdmk as k
    from pdm_ref import vae as ovae
    torch.manual_seed(3)
    B, C, H, W, ld = 2, 4, 6, 5, 32
    mom = torch.zeros(B * H * W, ld, device=dev)
    mom[:, :2 * C] = torch.randn(B * H * W, 2 * C, device=dev) * 3.0
    mom[0, C:2 * C] = torch.tensor([-100.0, -30.0, 20.0, 100.0], device=dev)       # logvar clamp to [-30, 20]
    mom = mom.to(DT[dn])
    eps = torch.randn(B, C, H, W, device=dev)
    z = torch.empty(B, C, H, W, device=dev)
    k.latent_sample(mom, eps, z, B, C, H * W, ld, 0.18215)
    mom_nchw = mom.float()[:, :2 * C].view(B, H, W, 2 * C).permute(0, 3, 1, 2).cpu()
    ref = ovae.sample_latents(mom_nchw, eps.cpu(), 0.18215)
    close(z, ref, 1e-5, "latent sample")


def _run(cfg_o, dn, x, dev, seed=7):
    from pdm.models.vae.autoencoder_kl import AutoencoderKL, VAEConfig
    from pdm_ref import vae as ovae
    sd = ovae.init_state_dict(cfg_o, seed=seed)
    cfg = VAEConfig(block_out_channels=cfg_o.block_out_channels, layers_per_block=cfg_o.layers_per_block)
    m = AutoencoderKL(cfg, dev, DT[dn], init=False)
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize("dn,tol", [("f32", 3e-4), ("bf16", 5e-2)])
@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_encoder_matches_reference_twin_outputs(dev, dn, tol, tag):
    """HIP path vs the outputs of the reference's vendored CompVis Encoder (golden fixture), same seeded weights."""
    from pdm_ref import vae as ovae
    cfg_o = ovae.VAEConfig.tiny() if tag == "tiny" else ovae.VAEConfig(block_out_channels=(32, 64, 128, 128))
    x = torch.from_numpy(GOLD[f"{tag}_x"])
    m, sd = _run(cfg_o, dn, x, dev)
    dist = m.encode(x.to(dev)).latent_dist
    mom = dist.parameters_nchw()
    ref = torch.from_numpy(GOLD[f"{tag}_moments"])
    assert mom.shape == ref.shape
    close(mom, ref, tol, f"moments[{tag}]")
    close(mom, ovae.encode_moments(sd, cfg_o, x), tol, f"moments[{tag}] vs oracle")
    eps = torch.from_numpy(GOLD[f"{tag}_eps"])
    z = dist.sample(noise=eps.to(dev), scale=0.18215)
    close(z, ovae.sample_latents(ref, eps, 0.18215), tol, f"latents[{tag}]")
    close(dist.mode(), ref[:, :4], tol, "mode")


@pytest.mark.parametrize("dn,tol", [("f32", 3e-4), ("bf16", 5e-2)])
@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_decoder_matches_reference_twin_outputs(dev, dn, tol, tag):
    """vae.decode vs the outputs of the reference's vendored CompVis Decoder (golden fixture) and the oracle."""
    from pdm_ref import vae as ovae
    cfg_o = ovae.VAEConfig.tiny() if tag == "tiny" else ovae.VAEConfig(block_out_channels=(32, 64, 128, 128))
    z = torch.from_numpy(GOLD[f"{tag}_zin"])
    m, sd = _run(cfg_o, dn, z, dev)
    img = m.decode(z.to(dev)).sample
    ref = torch.from_numpy(GOLD[f"{tag}_img"])
    assert img.shape == ref.shape and img.dtype == torch.float32
    close(img, ref, tol, f"decoded image[{tag}]")
    close(img, ovae.decode(sd, cfg_o, z), tol, f"decoded image[{tag}] vs oracle")


def test_sd21_width_decoder_matches_oracle_and_full_size_runs(dev):
    from pdm_ref import vae as ovae
    cfg_o = ovae.VAEConfig.sd21()
    g = torch.Generator().manual_seed(6)
    z = torch.randn(1, 4, 16, 16, generator=g)
    m, sd = _run(cfg_o, "bf16", z, dev, seed=3)
    img = m.decode(z.to(dev)).sample
    with torch.no_grad():
        ref = ovae.decode(sd, cfg_o, z)
    assert img.shape == (1, 3, 128, 128)
    close(img, ref, 5e-2, "sd21-width decoded image")
    big = m.decode(torch.randn(2, 4, 64, 64, device=dev)).sample            # 512 x 512 output
    assert big.shape == (2, 3, 512, 512) and torch.isfinite(big).all()


def test_state_dict_round_trip_and_legacy_attention_names(dev):
    from pdm.models.vae.autoencoder_kl import AutoencoderKL, VAEConfig
    from pdm_ref import vae as ovae
    cfg_o = ovae.VAEConfig.tiny()
    sd = ovae.init_state_dict(cfg_o, seed=1)
    m = AutoencoderKL(VAEConfig(block_out_channels=cfg_o.block_out_channels, layers_per_block=1), dev, torch.float32, init=False)
    legacy = {}
    for key, v in sd.items():       # the key names the reference's converter writes (convertModels.py:120-140) + decoder keys
        for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            key = key.replace(f"attentions.0.{new}.", f"attentions.0.{old}.")
        legacy[key] = v
    legacy["loss.logvar"] = torch.zeros(1)                   # keys of neither half are ignored
    m.load_state_dict(legacy)
    out = m.state_dict()
    assert set(out) == set(sd)
    for key in sd:
        assert out[key].shape == sd[key].shape and torch.equal(out[key], sd[key]), key


@pytest.mark.parametrize("dn,tol", [("f32", 3e-4), ("bf16", 5e-2)])
def test_sd21_width_encoder_matches_oracle(dev, dn, tol):
    """The real SD-2.1 VAE widths (128/256/512/512, 2 ResBlocks per level, 512-wide single-head attention) at 128x128."""
    from pdm_ref import vae as ovae
    cfg_o = ovae.VAEConfig.sd21()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 128, 128, generator=g) * 2 - 1
    m, sd = _run(cfg_o, dn, x, dev, seed=3)
    mom = m.encode(x.to(dev)).latent_dist.parameters_nchw()
    with torch.no_grad():
        ref = ovae.encode_moments(sd, cfg_o, x)
    assert mom.shape == (2, 8, 16, 16)
    close(mom, ref, tol, "sd21-width moments")


def test_full_size_batch_properties(dev):
    """512x512, B=4, bf16 (the BASELINE.json resolution): finite, deterministic, images independent of their batch mates,
    and latents consistent with the moments (size-independent properties; the oracle would take minutes here)."""
    from pdm.models.vae.autoencoder_kl import AutoencoderKL
    m = AutoencoderKL(None, dev, torch.bfloat16, seed=2)
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.rand(4, 3, 512, 512, device=dev, generator=g) * 2 - 1
    d1 = m.encode(x).latent_dist
    p1 = d1.parameters_nchw()
    assert p1.shape == (4, 8, 64, 64) and torch.isfinite(p1).all()
    p2 = m.encode(x).latent_dist.parameters_nchw()
    assert torch.equal(p1, p2), "encode must be deterministic"
    p3 = m.encode(x[2:3]).latent_dist.parameters_nchw()
    close(p3, p1[2:3], 2e-2, "image 2 alone vs in the batch")      # tile shapes may differ with M: not bit-equal
    noise = torch.randn(4, 4, 64, 64, device=dev, generator=g)
    z = d1.sample(noise=noise, scale=0.18215)
    mean, logvar = p1[:, :4], p1[:, 4:].clamp(-30, 20)
    close(z, (mean + torch.exp(0.5 * logvar) * noise) * 0.18215, 1e-5, "latents vs moments")
    z2 = m.encode_latents(x, generator=torch.Generator(device=dev).manual_seed(1))
    z3 = m.encode_latents(x, generator=torch.Generator(device=dev).manual_seed(1))
    assert torch.equal(z2, z3) and z2.shape == (4, 4, 64, 64)
