"""PNDM (PLMS) sampler with classifier-free guidance + VAE decode (oracle side, fp32 CPU) - SURVEY 8f row N3.

What the reference runs for image logging / FID (pdm/pipelines/pruning_pipelines.py:867-1010 `generate_samples`;
scripts/metrics/generate_fid_images.py:113-153: PNDMScheduler.from_pretrained(subfolder="scheduler"), 50 steps, guidance
7.5, 512 x 512): cat([negative, prompt]) embeddings, per step one U-Net call on cat([latents]*2), guidance
`uncond + g (text - uncond)`, `scheduler.step`, finally `vae.decode(latents / scaling_factor)` and `image/2 + 0.5` clamped.

PARITY UNPINNED for the scheduler: diffusers (0.30.3, env.yaml:52) is not importable here and the reference vendors no
PLMS twin.  `PNDM` below restates diffusers' PNDMScheduler as SD-2.1's scheduler_config uses it (skip_prk_steps=True,
steps_offset=1, set_alpha_to_one=False, scaled_linear betas 0.00085..0.012, prediction_type epsilon or v_prediction),
i.e. the pseudo linear multi-step method of Liu et al. 2022 (eq. 9-12): Adams-Bashforth combinations of the last four
noise predictions and the transfer formula `_get_prev_sample`.  tests/test_oracle_golden.py checks it against the two
closed forms it must reproduce: with one stored prediction the transfer is the deterministic DDIM step, and a constant
noise prediction makes every multi-step combination that same prediction.  The U-Net and the VAE decoder used here ARE
pinned (unet.py, vae.py).
"""
import torch

from . import vae as ovae
from .step import alphas_cumprod
from .unet import unet_forward


class PNDM:
    def __init__(self, num_train_timesteps=1000, steps_offset=1, prediction_type="epsilon"):
        self.ac = alphas_cumprod(num_train_timesteps).double()
        self.final_alpha_cumprod = self.ac[0]                    # set_alpha_to_one = False
        self.n_train, self.steps_offset, self.prediction_type = num_train_timesteps, steps_offset, prediction_type
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.n_train // n
        base = (torch.arange(0, n) * ratio).round().long() + self.steps_offset
        # skip_prk_steps: no Runge-Kutta warm-up; the second step is repeated once instead (diffusers set_timesteps)
        self.timesteps = torch.cat([base[:-1], base[-2:-1], base[-1:]]).flip(0).tolist()
        self.ets, self.counter, self.cur_sample = [], 0, None

    def get_prev_sample(self, sample, t, prev_t, eps):
        a_t = self.ac[t]
        a_prev = self.ac[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if self.prediction_type == "v_prediction":
            eps = a_t.sqrt() * eps + b_t.sqrt() * sample
        coeff = (a_prev / a_t).sqrt()
        denom = a_t * b_prev.sqrt() + (a_t * b_t * a_prev).sqrt()
        return (coeff * sample - (a_prev - a_t) * eps / denom)

    def step(self, model_output, t, sample):
        ratio = self.n_train // self.num_inference_steps
        prev_t = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:] + [model_output]
        else:
            prev_t, t = t, t + ratio
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            model_output = (model_output + e[-1]) / 2
            sample, self.cur_sample = self.cur_sample, None
        elif len(e) == 2:
            model_output = (3 * e[-1] - e[-2]) / 2
        elif len(e) == 3:
            model_output = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
        else:
            model_output = (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4]) / 24
        self.counter += 1
        return self.get_prev_sample(sample.double(), t, prev_t, model_output.double()).to(sample.dtype)


def generate(unet, cfg, vae_sd, vae_cfg, prompt_embeds, negative_prompt_embeds, latents, num_inference_steps=50,
             guidance_scale=7.5, prediction_type="epsilon"):
    """unet = (sd, info).  latents [B,4,h,w] ~ N(0,1).  Returns (final latents, image in [0,1] NCHW)."""
    sch = PNDM(prediction_type=prediction_type)
    sch.set_timesteps(num_inference_steps)
    ehs = torch.cat([negative_prompt_embeds, prompt_embeds])
    latents = latents * sch.init_noise_sigma
    B = latents.shape[0]
    with torch.no_grad():
        for t in sch.timesteps:
            tt = torch.full((2 * B,), t, dtype=torch.long)
            out = unet_forward(unet[0], cfg, unet[1], torch.cat([latents] * 2), tt, ehs, {})
            un, tx = out.chunk(2)
            latents = sch.step(un + guidance_scale * (tx - un), t, latents)
        img = ovae.decode(vae_sd, vae_cfg, latents / vae_cfg.scaling_factor)
    return latents, (img / 2 + 0.5).clamp(0, 1)
