"""Sampler for image logging / FID (-m gpu): StableDiffusionPruningPipeline.generate_samples on
libpdmk against the oracle pipeline (oracle/pdm_ref/sampler.py: pinned U-Net and VAE decoder, PNDM restated - scheduler
parity is "unpinned", its closed-form checks live in tests/test_oracle_golden.py).  fp32 engine path, tolerance 2e-3 of
scale after 5-7 chained U-Net calls; bf16 is checked for agreement in direction only (chaotic amplification of rounding)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def close(got, ref, tol, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert math.isfinite(err) and err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def _setup(dev, dtype, prediction_type):
    from pdm_ref import arch as oarch, weights as oweights, vae as ovae
    from pdm_ref.config import UNetConfig as OCfg
    from pdm.models.unet.spec import UNetConfig
    from pdm.models.unet.unet_2d_conditional import UNet2DConditionModelPruned
    from pdm.models.vae.autoencoder_kl import AutoencoderKL, VAEConfig
    from pdm.pipelines.pruning_pipelines import StableDiffusionPruningPipeline, PNDMScheduler
    ocfg, cfg = OCfg.tiny(), UNetConfig.tiny()
    dense = oweights.init_dense_state_dict(ocfg, seed=0)
    av = oarch.random_arch_vector(ocfg, 0.6, seed=1, drop_depth=(1,))
    psd, info = oweights.prune_state_dict(dense, ocfg, av)
    unet = UNet2DConditionModelPruned(cfg, av, dev, dtype, train=False, init=False)
    unet.load_dense_or_pruned(dense)
    vcfg = ovae.VAEConfig.tiny()
    vsd = ovae.init_state_dict(vcfg, seed=7)
    vae = AutoencoderKL(VAEConfig(block_out_channels=vcfg.block_out_channels, layers_per_block=1), dev, dtype, init=False)
    vae.load_state_dict(vsd)
    pipe = StableDiffusionPruningPipeline(vae, None, unet, PNDMScheduler(prediction_type=prediction_type))
    g = torch.Generator().manual_seed(4)
    pe, ne = torch.randn(2, 13, 64, generator=g), torch.randn(1, 13, 64, generator=g).expand(2, -1, -1).contiguous()
    lat = torch.randn(2, 4, 16, 16, generator=g)
    return pipe, (psd, info), ocfg, vsd, vcfg, pe, ne, lat


@pytest.mark.parametrize("prediction_type", ["epsilon", "v_prediction"])
@pytest.mark.parametrize("steps", [1, 2, 6])
def test_sampler_matches_oracle_fp32(dev, prediction_type, steps):
    from pdm_ref import sampler as osampler
    pipe, ounet, ocfg, vsd, vcfg, pe, ne, lat = _setup(dev, torch.float32, prediction_type)
    lat_ref, img_ref = osampler.generate(ounet, ocfg, vsd, vcfg, pe, ne, lat, steps, 7.5, prediction_type)
    out = pipe.generate_samples(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=steps,
                                guidance_scale=7.5, output_type="latent")
    close(out.images, lat_ref, 2e-3, f"latents after {steps} PNDM steps")
    img = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=steps, output_type="pt").images
    assert img.shape == (2, 3, 64, 64) and img.min() >= 0 and img.max() <= 1
    close(img, img_ref, 5e-3, "decoded image in [0, 1]")


def test_sampler_no_guidance_bf16_and_outputs(dev):
    from pdm_ref import sampler as osampler
    pipe, ounet, ocfg, vsd, vcfg, pe, ne, lat = _setup(dev, torch.bfloat16, "epsilon")
    lat_ref, _ = osampler.generate(ounet, ocfg, vsd, vcfg, pe, ne, lat, 3, 7.5, "epsilon")
    got = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=3, output_type="latent").images
    cos = torch.nn.functional.cosine_similarity(got.float().cpu().flatten(), lat_ref.flatten(), dim=0).item()
    assert cos > 0.99, cos
    # guidance_scale <= 1: single-batch U-Net calls, no negative prompt needed
    one = pipe(prompt_embeds=pe, latents=lat, num_inference_steps=2, guidance_scale=1.0, output_type="np").images
    assert one.shape == (2, 64, 64, 3) and one.dtype.name == "float32"
    gen = torch.Generator(device=dev).manual_seed(0)
    a = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=2, generator=gen, height=64, width=64,
             output_type="latent").images
    gen = torch.Generator(device=dev).manual_seed(0)
    b = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=2, generator=gen, height=64, width=64,
             output_type="latent").images
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        pipe(prompt_embeds=pe, num_inference_steps=2)                      # guidance without a negative prompt
    with pytest.raises(ValueError):
        pipe(prompt_embeds=pe, negative_prompt_embeds=ne, height=66, width=64, num_inference_steps=1)
