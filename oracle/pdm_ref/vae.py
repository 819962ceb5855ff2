"""Functional fp32 AutoencoderKL encoder and decoder (row N3) on a state dict (oracle side, NCHW).

What the reference runs right before the U-Net (pdm/training/trainer.py:2405-2406):
    latents = vae.encode(pixel_values).latent_dist.sample() * vae.config.scaling_factor
with the SD-2.1 VAE (diffusers AutoencoderKL: block_out_channels (128,256,512,512), 2 ResBlocks per level, GroupNorm(32,
eps 1e-6), one single-head attention in the middle, latent_channels 4, scaling_factor 0.18215).  diffusers is not in the
container; the twin that IS importable is the CompVis encoder the diffusers class was converted from:
    baselines/erasing/oldcode_erasing_compvis/ldm/modules/diffusionmodules/model.py
        Encoder :368-460, ResnetBlock :82-143, AttnBlock :150-204, Downsample (pad (0,1,0,1) + stride-2 conv) :60-81
    baselines/.../ldm/modules/distributions/distributions.py:24-37   DiagonalGaussianDistribution (clamp, std, sample)
and the key-name correspondence CompVis -> diffusers is the reference's own converter
    baselines/.../train-scripts/convertModels.py:481-600 (convert_ldm_vae_checkpoint).
`oracle/validate_vae_against_reference.py` pins this file against those twins.  State-dict keys are the diffusers ones.
Decoder (image logging / FID sampling, pdm/pipelines/pruning_pipelines.py:993-995: vae.decode(latents / scaling_factor)):
twin Decoder model.py:462-569, Upsample (nearest x2 + conv) :42-58; diffusers order up_blocks[i] = CompVis up[n-1-i]
(convertModels.py:555-580).
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215
    eps: float = 1e-6

    @staticmethod
    def sd21():
        return VAEConfig()

    @staticmethod
    def tiny():
        return VAEConfig(block_out_channels=(32, 64, 64), layers_per_block=1)


def resnet(sd, p, x, groups, eps):
    h = F.silu(F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def mid_attention(sd, p, x, groups, eps):
    """Single head over all channels, scale C^-1/2, biases on q/k/v/out, residual (model.py:150-204)."""
    B, C, H, W = x.shape
    h = F.group_norm(x, groups, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], eps)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    s = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (C ** -0.5), dim=-1)
    o = F.linear(torch.bmm(s, v), sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return x + o.reshape(B, H, W, C).permute(0, 3, 1, 2)


def encode_moments(sd, cfg: VAEConfig, x):
    """pixels [B,3,R,R] -> moments [B, 2*latent, R/8.., ..] = quant_conv(encoder(x))  (mean | logvar)."""
    G, eps = cfg.norm_num_groups, cfg.eps
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, G, eps)
        if i != n - 1:
            p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[p + ".weight"], sd[p + ".bias"], stride=2)
    h = resnet(sd, "encoder.mid_block.resnets.0", h, G, eps)
    h = mid_attention(sd, "encoder.mid_block.attentions.0", h, G, eps)
    h = resnet(sd, "encoder.mid_block.resnets.1", h, G, eps)
    h = F.silu(F.group_norm(h, G, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], eps))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def decode(sd, cfg: VAEConfig, z):
    """latents [B, latent, h, w] (already divided by scaling_factor) -> image [B, 3, 8h.., ..] = decoder(post_quant_conv(z))."""
    G, eps = cfg.norm_num_groups, cfg.eps
    h = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = resnet(sd, "decoder.mid_block.resnets.0", h, G, eps)
    h = mid_attention(sd, "decoder.mid_block.attentions.0", h, G, eps)
    h = resnet(sd, "decoder.mid_block.resnets.1", h, G, eps)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            h = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, G, eps)
        if i != n - 1:
            p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd[p + ".weight"], sd[p + ".bias"], padding=1)
    h = F.silu(F.group_norm(h, G, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], eps))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def sample_latents(moments, eps_noise, scaling_factor):
    """latent_dist.sample() * scaling_factor with the Gaussian draw `eps_noise` supplied (distributions.py:24-37)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return (mean + std * eps_noise) * scaling_factor


def init_state_dict(cfg: VAEConfig, seed=0, jitter=True):
    """Random encoder weights under the diffusers key names (PyTorch-default uniform fan-in init; norm affine jittered so
    parity tests see non-trivial scales)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, k):
        b = (ci * k * k) ** -0.5
        sd[name + ".weight"] = (torch.rand(co, ci, k, k, generator=g) * 2 - 1) * b
        sd[name + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * b

    def lin(name, co, ci):
        b = ci ** -0.5
        sd[name + ".weight"] = (torch.rand(co, ci, generator=g) * 2 - 1) * b
        sd[name + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * b

    def norm(name, c):
        sd[name + ".weight"] = 1.0 + (0.2 * torch.randn(c, generator=g) if jitter else torch.zeros(c))
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g) if jitter else torch.zeros(c)

    def res(p, ci, co):
        norm(p + ".norm1", ci)
        conv(p + ".conv1", co, ci, 3)
        norm(p + ".norm2", co)
        conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    ch = cfg.block_out_channels
    conv("encoder.conv_in", ch[0], cfg.in_channels, 3)
    cin = ch[0]
    for i, co in enumerate(ch):
        for j in range(cfg.layers_per_block):
            res(f"encoder.down_blocks.{i}.resnets.{j}", cin, co)
            cin = co
        if i != len(ch) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    res("encoder.mid_block.resnets.0", cin, cin)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", cin)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        lin(f"{a}.{nm}", cin, cin)
    res("encoder.mid_block.resnets.1", cin, cin)
    norm("encoder.conv_norm_out", cin)
    conv("encoder.conv_out", 2 * cfg.latent_channels, cin, 3)
    conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    # decoder half (drawn after the encoder so the encoder's weights do not depend on it)
    conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    rev = tuple(reversed(ch))
    conv("decoder.conv_in", rev[0], cfg.latent_channels, 3)
    res("decoder.mid_block.resnets.0", rev[0], rev[0])
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", rev[0])
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        lin(f"{a}.{nm}", rev[0], rev[0])
    res("decoder.mid_block.resnets.1", rev[0], rev[0])
    cin = rev[0]
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", cin, co)
            cin = co
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm("decoder.conv_norm_out", cin)
    conv("decoder.conv_out", cfg.in_channels, cin, 3)
    return sd
