"""`StableDiffusionPruningPipeline.generate_samples` on libpdmk - image logging / FID sampling.

Mirror of pdm/pipelines/pruning_pipelines.py:867-1010 as scripts/metrics/generate_fid_images.py:113-153 drives it:
PNDM (PLMS, skip_prk_steps) scheduler, classifier-free guidance on a doubled batch, `vae.decode(latents /
scaling_factor)`, `image / 2 + 0.5` clamped to [0, 1].  Models are this package's `UNet2DConditionModelPruned`,
`AutoencoderKL`, `CLIPTextModel`; the scheduler's per-step latent arithmetic runs in `pdmk_axpby` (fp32), its scalar
coefficients on the host in float64 like diffusers.  Prompts come as token ids or embeddings (tokenisation is host-side
data preparation).  The safety checker of the diffusers base class is not reproduced (the reference's FID script keeps
every image).  Scheduler parity is "unpinned" (diffusers absent, no vendored twin): see oracle/pdm_ref/sampler.py.
"""
from types import SimpleNamespace

import torch

from .. import _pdmk as k


class PNDMScheduler:
    """diffusers PNDMScheduler as configured by SD-2.1's scheduler_config.json (skip_prk_steps, steps_offset 1,
    set_alpha_to_one False, scaled_linear betas)."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 prediction_type="epsilon"):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).double()
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, steps_offset=steps_offset,
                                      prediction_type=prediction_type, skip_prk_steps=True)
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        base = (torch.arange(0, num_inference_steps) * ratio).round().long() + self.config.steps_offset
        self.timesteps = torch.cat([base[:-1], base[-2:-1], base[-1:]]).flip(0)
        self.ets, self.counter, self.cur_sample = [], 0, None

    def scale_model_input(self, sample, t=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict=True):
        """model_output / sample: contiguous fp32 device tensors of one shape; returns the previous sample (new tensor)."""
        t = int(timestep)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        prev_t = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:] + [model_output]
        else:
            prev_t, t = t, t + ratio
        e = self.ets
        mo = model_output.clone()
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            k.axpby(e[-1], mo, 0.5, 0.5)                                   # (model_output + ets[-1]) / 2
            sample, self.cur_sample = self.cur_sample, None
        else:
            coef = {2: (3 / 2, -1 / 2), 3: (23 / 12, -16 / 12, 5 / 12), 4: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}[len(e)]
            mo.copy_(e[-1])
            k.axpby(e[-2], mo, coef[1], coef[0])
            for i in range(2, len(e)):
                k.axpby(e[-1 - i], mo, coef[i], 1.0)
        self.counter += 1
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if self.config.prediction_type == "v_prediction":
            k.axpby(sample, mo, float(b_t.sqrt()), float(a_t.sqrt()))     # eps = sqrt(a) v + sqrt(1 - a) x
        elif self.config.prediction_type != "epsilon":
            raise ValueError(f"prediction_type {self.config.prediction_type!r} must be epsilon or v_prediction")
        coeff = float((a_prev / a_t).sqrt())
        denom = float(a_t * b_prev.sqrt() + (a_t * b_t * a_prev).sqrt())
        prev = sample.clone()
        k.axpby(mo, prev, -float(a_prev - a_t) / denom, coeff)            # coeff * sample - (a_prev - a_t) eps / denom
        return SimpleNamespace(prev_sample=prev) if return_dict else (prev,)


class StableDiffusionPruningPipeline:
    def __init__(self, vae, text_encoder, unet, scheduler=None, tokenizer=None):
        self.vae, self.text_encoder, self.unet, self.tokenizer = vae, text_encoder, unet, tokenizer
        self.scheduler = scheduler or PNDMScheduler()
        self.vae_scale_factor = 2 ** (len(vae.cfg.block_out_channels) - 1)
        self.device = unet.device

    def encode_prompt(self, prompt_ids=None, negative_prompt_ids=None, prompt_embeds=None, negative_prompt_embeds=None,
                      do_classifier_free_guidance=True):
        if prompt_embeds is None:
            if prompt_ids is None:
                raise ValueError("pass prompt_embeds or prompt_ids (token ids; tokenisation is host-side)")
            prompt_embeds = self.text_encoder(prompt_ids)[0]
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            if negative_prompt_ids is None:
                raise ValueError("classifier-free guidance needs negative_prompt_embeds or negative_prompt_ids "
                                 "(the tokenised empty prompt)")
            negative_prompt_embeds = self.text_encoder(negative_prompt_ids)[0]
        return prompt_embeds, negative_prompt_embeds

    @torch.no_grad()
    def generate_samples(self, prompt_ids=None, height=None, width=None, num_inference_steps=50, guidance_scale=7.5,
                         negative_prompt_ids=None, generator=None, latents=None, prompt_embeds=None,
                         negative_prompt_embeds=None, output_type="np", return_dict=True, callback=None, callback_steps=1):
        cfg_on = guidance_scale > 1.0
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(prompt_ids, negative_prompt_ids, prompt_embeds,
                                                                   negative_prompt_embeds, cfg_on)
        B = prompt_embeds.shape[0]
        f = self.vae_scale_factor
        if latents is not None:            # caller-provided latents fix the size (prepare_latents would reject a mismatch)
            height, width = height or latents.shape[2] * f, width or latents.shape[3] * f
        height, width = height or 64 * f, width or 64 * f     # unet.config.sample_size * vae_scale_factor at 512 px
        if height % f or width % f:
            raise ValueError(f"`height` and `width` have to be divisible by {f} but are {height} and {width}.")
        dev = self.device
        ehs = torch.cat([negative_prompt_embeds.to(dev), prompt_embeds.to(dev)]) if cfg_on else prompt_embeds.to(dev)
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps, device=dev)
        C = self.unet.cfg.in_channels
        shape = (B, C, height // f, width // f)
        if latents is None:
            latents = torch.randn(shape, device=dev, dtype=torch.float32, generator=generator)
        latents = (latents.to(dev, torch.float32) * sch.init_noise_sigma).contiguous()
        was_training = self.unet.training
        self.unet.eval()
        try:
            x2 = torch.empty((2 * B if cfg_on else B,) + shape[1:], device=dev, dtype=torch.float32)
            for i, t in enumerate(sch.timesteps.tolist()):
                x2[:B].copy_(latents)
                if cfg_on:
                    x2[B:].copy_(latents)
                tt = torch.full((x2.shape[0],), t, device=dev, dtype=torch.int64)
                out = self.unet(sch.scale_model_input(x2, t), tt, ehs, return_dict=False)[0]
                if cfg_on:        # uncond + g (text - uncond) = (1 - g) uncond + g text, in place on the text half
                    noise = out[B:]
                    k.axpby(out[:B], noise, 1.0 - guidance_scale, guidance_scale)
                else:
                    noise = out
                latents = sch.step(noise.contiguous(), t, latents, return_dict=False)[0]
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, latents)
        finally:
            self.unet.train(was_training)
        if output_type == "latent":
            image = latents
        else:
            image = self.vae.decode(latents / self.vae.cfg.scaling_factor, return_dict=False)[0]
            image = (image / 2 + 0.5).clamp(0, 1)                         # VaeImageProcessor.postprocess (denormalize)
            if output_type == "np":
                image = image.permute(0, 2, 3, 1).cpu().numpy()
            elif output_type != "pt":
                raise ValueError("output_type must be 'latent', 'pt' or 'np' (PIL conversion is left to the caller)")
        return SimpleNamespace(images=image, nsfw_content_detected=None) if return_dict else (image, None)

    __call__ = generate_samples
