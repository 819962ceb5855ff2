"""Scheduler scalars of the loss heads (host side, computed once; O(1000) floats).

compute_snr mirrors pdm/utils/metric_utils.py:3-26; the DDIM/SD-2.1 schedule is SURVEY Appendix B.9
(scaled_linear betas: fp32 linspace of sqrt(beta) in [sqrt(0.00085), sqrt(0.012)], squared; alphas_cumprod = cumprod).
"""
import torch


def alphas_cumprod_sd(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def compute_snr(noise_scheduler, timesteps):
    """snr_t = (sqrt(acp_t) / sqrt(1 - acp_t))^2; `noise_scheduler` needs an `.alphas_cumprod` tensor."""
    acp = noise_scheduler.alphas_cumprod
    alpha = (acp ** 0.5).to(timesteps.device)[timesteps].float()
    sigma = ((1.0 - acp) ** 0.5).to(timesteps.device)[timesteps].float()
    return (alpha / sigma) ** 2


def min_snr_weight_table(acp, gamma, v_prediction=True):
    """Per-timestep loss weight min(snr', gamma) / snr' with snr' = snr + 1 for v-prediction (trainer.py:2457-2466)."""
    from types import SimpleNamespace
    snr = compute_snr(SimpleNamespace(alphas_cumprod=acp), torch.arange(acp.numel()))
    if v_prediction:
        snr = snr + 1
    return torch.minimum(snr, gamma * torch.ones_like(snr)) / snr
