"""Scheduler arithmetic, loss heads and the main / upper training steps (oracle side, fp32 CPU, torch autograd).

  DDIM schedule / add_noise / get_velocity    SURVEY Appendix B.9 (diffusers DDIMScheduler semantics);
                                               twin: ldm/modules/diffusionmodules/util.py:21-44 (float64)
  compute_snr                                 pdm/utils/metric_utils.py:3-26
  step()   (DDPM min-SNR + block + distill)   pdm/training/trainer.py:2403-2488
  upper_step() (negative-guidance target)     pdm/training/trainer.py:2904-3001
  AdamW / constant_with_warmup                trainer.py:265-284, 436-443
"""
import torch
import torch.nn.functional as F

import contextlib

from . import unet as _unet
from .unet import unet_forward

BLOCK_KEYS = ("d0", "d1", "d2", "d3", "m", "u0", "u1", "u2", "u3")


def alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(ac, x0, noise, t):
    a = ac[t].sqrt().view(-1, 1, 1, 1)
    s = (1.0 - ac[t]).sqrt().view(-1, 1, 1, 1)
    return a * x0 + s * noise


def get_velocity(ac, x0, noise, t):
    a = ac[t].sqrt().view(-1, 1, 1, 1)
    s = (1.0 - ac[t]).sqrt().view(-1, 1, 1, 1)
    return a * noise - s * x0


def compute_snr(ac, t):
    alpha = ac.sqrt()[t].float()
    sigma = (1.0 - ac).sqrt()[t].float()
    return (alpha / sigma) ** 2


def min_snr_weights(ac, t, gamma=5.0, v_prediction=True):
    snr = compute_snr(ac, t)
    if v_prediction:
        snr = snr + 1
    return torch.minimum(snr, gamma * torch.ones_like(snr)) / snr


def main_loss_heads(pred, target, full, acts_s, acts_t, ac, t, w_diff=1.0, w_block=0.1, w_dist=2.0, gamma=5.0):
    """The three loss heads of UnetFineTuner.step (trainer.py:2451-2488) on given tensors: DDPM min-SNR(gamma) (plain MSE
    when gamma is None), block-feature MSE averaged over the hooked blocks, output distillation; returns
    (loss, diff, dist, block) in the reference's return order.  Pinned by tests/golden/reference_loss_heads.npz (the
    reference's own statements executed on the same tensors)."""
    if gamma is None:
        l = F.mse_loss(pred.float(), target.float(), reduction="mean")
    else:
        w = min_snr_weights(ac, t, gamma)
        l = F.mse_loss(pred.float(), target.float(), reduction="none")
        l = (l.mean(dim=(1, 2, 3)) * w).mean()
    diff = l.detach().clone()
    loss = l * w_diff
    block = torch.zeros(())
    if w_block > 0:
        for k in acts_s:
            block = block + F.mse_loss(acts_s[k], acts_t[k].detach())
        block = block / len(acts_s)
        loss = loss + w_block * block
    dist = torch.zeros(())
    if w_dist > 0:
        dist = F.mse_loss(pred.float(), full.float())
        loss = loss + w_dist * dist
    return loss, diff, dist.detach(), block.detach()


def upper_loss_heads(pred, e_c, e_u, acts_s, acts_t, w_dist=1.0, w_block=0.0):
    """The heads of BilevelUnetFineTuner.upper_step (trainer.py:2983-3001): negative-guidance distillation target
    e_u - (e_c - e_u) (no .float() cast) + the block term against the teacher's LAST call; diff_loss is 0."""
    loss = torch.zeros(())
    block = torch.zeros(())
    if w_block > 0:
        for k in acts_s:
            block = block + F.mse_loss(acts_s[k], acts_t[k].detach())
        block = block / len(acts_s)
        loss = loss + w_block * block
    dist = torch.zeros(())
    if w_dist > 0:
        dist = F.mse_loss(pred, e_u - (e_c - e_u))
        loss = loss + w_dist * dist
    return loss, torch.zeros(()), dist.detach(), block.detach()


class _mixed:
    """`--mixed_precision bf16` as accelerate applies it to the reference trainer (trainer.py:516-527, 2730-2733; SURVEY
    Appendix B.11): the frozen teacher's weights are CAST to bf16, the student keeps fp32 master weights and its forward runs
    under torch.autocast(bfloat16) (here the CPU autocast: conv / linear / matmul in bf16, the loss heads on what comes out
    of them - H1/H2 upcast with .float(), H3/H4 do not).  mixed=False: plain fp32.
    mixed="cuda": the same, with CUDA autocast's op policy where it differs from the CPU autocast's (unet.FP32_NORMS:
    group_norm / layer_norm upcast to fp32, attention as the fused SDPA computes it) - a second, equally valid bf16
    evaluation of the same step; the two differ from each other by as much as either differs from fp32, which is what the
    curve test uses to size its bound."""

    def __init__(self, on):
        self.on = bool(on)
        self.cuda_policy = on == "cuda"

    def teacher_sd(self, sd):
        if not self.on:
            return sd
        key = id(sd)
        if _mixed._cache.get("key") != key:
            _mixed._cache = {"key": key, "sd": {k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in sd.items()}}
        return _mixed._cache["sd"]

    @contextlib.contextmanager
    def ctx(self):
        prev = _unet.FP32_NORMS
        _unet.FP32_NORMS = self.cuda_policy
        try:
            with (torch.autocast("cpu", dtype=torch.bfloat16) if self.on else torch.autocast("cpu", enabled=False)):
                yield
        finally:
            _unet.FP32_NORMS = prev

    _cache = {}


def main_step_loss(student, teacher, cfg, ac, latents, noise, t, ehs, w_diff=1.0, w_block=0.1, w_dist=2.0, gamma=5.0,
                   mixed=False):
    """student/teacher = (sd, info).  Returns (loss, diff, dist, block, pred) like trainer.py:2488.
    mixed=True: the reference's `--mixed_precision bf16` numerics (see _mixed) instead of fp32."""
    noisy = add_noise(ac, latents, noise, t)
    target = get_velocity(ac, latents, noise, t)
    acts_t, acts_s = {}, {}
    mp = _mixed(mixed)
    with mp.ctx():
        with torch.no_grad():
            full = unet_forward(mp.teacher_sd(teacher[0]), cfg, teacher[1], noisy, t, ehs, acts_t)
        pred = unet_forward(student[0], cfg, student[1], noisy, t, ehs, acts_s)
    acts_s = {k: acts_s[k] for k in BLOCK_KEYS}
    loss, diff, dist, block = main_loss_heads(pred, target, full, acts_s, acts_t, ac, t, w_diff, w_block, w_dist, gamma)
    return loss, diff, dist, block, pred


def upper_step_loss(student, teacher, cfg, ac, latents, noise, t, ehs, empty_ehs, w_dist=1.0, w_block=0.0, mixed=False):
    noisy = add_noise(ac, latents, noise, t)
    acts_t, acts_s = {}, {}
    mp = _mixed(mixed)
    with mp.ctx():
        tsd = mp.teacher_sd(teacher[0])
        with torch.no_grad():
            e_c = unet_forward(tsd, cfg, teacher[1], noisy, t, ehs, {})
            e_u = unet_forward(tsd, cfg, teacher[1], noisy, t, empty_ehs, acts_t)   # the hooks hold the LAST call (uncond)
        pred = unet_forward(student[0], cfg, student[1], noisy, t, ehs, acts_s)
    acts_s = {k: acts_s[k] for k in BLOCK_KEYS}
    loss, diff, dist, block = upper_loss_heads(pred, e_c, e_u, acts_s, acts_t, w_dist, w_block)
    return loss, diff, dist, block, pred


def adamw_step(params, grads, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.AdamW semantics (decoupled weight decay, bias-corrected), in place; step is 1-based."""
    for k in params:
        g = grads[k]
        p = params[k]
        p.mul_(1 - lr * wd)
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (v[k].sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m[k], denom, value=-lr / bc1)


def constant_with_warmup(base_lr, k, warmup):
    """lr after k scheduler steps (k counts scheduler.step() calls; accelerate steps it W times per optimiser step)."""
    return base_lr * min(1.0, k / max(1, warmup)) if warmup > 0 else base_lr
