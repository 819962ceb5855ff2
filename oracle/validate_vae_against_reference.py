#!/usr/bin/env python3
"""Pin oracle/pdm_ref/vae.py against the CompVis VAE encoder vendored in the reference and emit golden fixtures.

Runs ONLY in the build container (needs /root/reference).  The oracle's random diffusers-named state dict is loaded into
the twin modules through the CompVis<->diffusers key correspondence of the reference's own converter
(baselines/erasing/oldcode_erasing_compvis/train-scripts/convertModels.py:481-600); the twin's outputs on seeded inputs
are written to tests/golden/vae_twin.npz (plain data: inputs + expected outputs; weights are regenerated from the seed).

Checked:
  ldm/modules/diffusionmodules/model.py  Encoder (:368-460) whole forward, AttnBlock (:150-204), Downsample (:60-81)
  ldm/modules/distributions/distributions.py  DiagonalGaussianDistribution (:24-37): clamp / std / sample
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REF, "baselines/erasing/oldcode_erasing_compvis"))

from pdm_ref import vae  # noqa: E402
from ldm.modules.diffusionmodules.model import Encoder, Decoder, AttnBlock, Downsample  # noqa: E402
from ldm.modules.distributions.distributions import DiagonalGaussianDistribution  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
report = []


def ok(name, a, b, tol):
    err = float((a - b).abs().max())
    scale = float(b.abs().max()) + 1e-12
    status = "OK " if err <= tol * max(1.0, scale) else "FAIL"
    report.append(f"{status} {name}: max|diff|={err:.3e} (scale {scale:.3e}, tol {tol:g})")
    print(report[-1])
    assert status == "OK ", name


def twin_key(k):
    """diffusers encoder key -> CompVis Encoder key (inverse of convertModels.py:481-600)."""
    k = k[len("encoder."):]
    k = k.replace("conv_norm_out", "norm_out").replace("conv_shortcut", "nin_shortcut")
    k = k.replace("mid_block.resnets.0", "mid.block_1").replace("mid_block.resnets.1", "mid.block_2")
    k = k.replace("mid_block.attentions.0.group_norm", "mid.attn_1.norm")
    for a, b in (("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("to_out.0", "proj_out")):
        k = k.replace(f"mid_block.attentions.0.{a}.", f"mid.attn_1.{b}.")
    if k.startswith("down_blocks."):
        _, i, kind, j, rest = k.split(".", 4)
        k = f"down.{i}.block.{j}.{rest}" if kind == "resnets" else f"down.{i}.downsample.{rest}"
    return k


def twin_key_dec(k, nlev):
    """diffusers decoder key -> CompVis Decoder key: up_blocks[i] = up[nlev-1-i] (convertModels.py:555-580)."""
    k = "encoder." + k[len("decoder."):]                  # reuse the shared renames, then fix the up path
    if k.startswith("encoder.up_blocks."):
        _, _, i, kind, j, rest = k.split(".", 5)
        lvl = nlev - 1 - int(i)
        rest = rest.replace("conv_shortcut", "nin_shortcut")
        return f"up.{lvl}.block.{j}.{rest}" if kind == "resnets" else f"up.{lvl}.upsample.{rest}"
    return twin_key(k)


def load_twin(enc, sd, prefix="encoder.", nlev=0):
    tsd = {}
    for k, v in sd.items():
        if not k.startswith(prefix):
            continue
        tk = twin_key(k) if prefix == "encoder." else twin_key_dec(k, nlev)
        if ".attn_1." in tk and tk.endswith(".weight") and v.dim() == 2:
            v = v[:, :, None, None]                      # the twin's q/k/v/proj_out are 1x1 convs
        tsd[tk] = v
    missing, unexpected = enc.load_state_dict(tsd, strict=True), None
    return missing


golden = {}
for tag, cfg, res, B in (("tiny", vae.VAEConfig.tiny(), 32, 2),
                         ("mid", vae.VAEConfig(block_out_channels=(32, 64, 128, 128), layers_per_block=2), 64, 1)):
    sd = vae.init_state_dict(cfg, seed=7)
    ch = cfg.block_out_channels
    enc = Encoder(ch=ch[0], out_ch=3, ch_mult=tuple(c // ch[0] for c in ch), num_res_blocks=cfg.layers_per_block,
                  attn_resolutions=[], in_channels=3, resolution=res, z_channels=cfg.latent_channels, double_z=True)
    load_twin(enc, sd)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 3, res, res, generator=g) * 2 - 1
    with torch.no_grad():
        h_ref = enc(x)
        mom_ref = torch.nn.functional.conv2d(h_ref, sd["quant_conv.weight"], sd["quant_conv.bias"])
        mom = vae.encode_moments(sd, cfg, x)
    ok(f"Encoder[{tag}] + quant_conv moments", mom, mom_ref, 2e-5)
    dist = DiagonalGaussianDistribution(mom_ref * 4.0)          # x4: exercises a wider logvar range
    torch.manual_seed(5)
    z_ref = dist.sample()
    torch.manual_seed(5)
    eps = torch.randn(dist.mean.shape)
    ok(f"DiagonalGaussian[{tag}] sample", vae.sample_latents(mom_ref * 4.0, eps, 1.0), z_ref, 1e-6)
    # decoder half: post_quant_conv + Decoder on seeded latents
    dec = Decoder(ch=ch[0], out_ch=3, ch_mult=tuple(c // ch[0] for c in ch), num_res_blocks=cfg.layers_per_block,
                  attn_resolutions=[], in_channels=3, resolution=res, z_channels=cfg.latent_channels)
    load_twin(dec, sd, prefix="decoder.", nlev=len(ch))
    zin = torch.randn(B, cfg.latent_channels, *mom_ref.shape[2:], generator=g)
    with torch.no_grad():
        img_ref = dec(torch.nn.functional.conv2d(zin, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"]))
        img = vae.decode(sd, cfg, zin)
    ok(f"post_quant_conv + Decoder[{tag}] image", img, img_ref, 2e-5)
    golden[f"{tag}_zin"], golden[f"{tag}_img"] = zin.numpy(), img_ref.numpy()
    golden[f"{tag}_x"], golden[f"{tag}_moments"] = x.numpy(), mom_ref.numpy()
    golden[f"{tag}_eps"], golden[f"{tag}_z4"] = eps.numpy(), z_ref.numpy()

# logvar clamp: beyond +-30/20 the twin saturates
mom = torch.zeros(1, 8, 2, 2)
mom[:, 4:] = torch.tensor([-100.0, -30.0, 20.0, 100.0]).view(1, 1, 2, 2)
dist = DiagonalGaussianDistribution(mom)
ok("DiagonalGaussian clamp", vae.sample_latents(mom, torch.ones(1, 4, 2, 2), 1.0), dist.mean + dist.std, 1e-6)

# single sub-blocks at the real width (512 channels, one head): attention scale C^-1/2 and asymmetric-pad downsample
torch.manual_seed(3)
ab = AttnBlock(512)
for prm in ab.parameters():
    torch.nn.init.normal_(prm, std=0.04) if prm.dim() > 1 else torch.nn.init.normal_(prm, mean=0.3, std=0.2)
sd = {"a.group_norm.weight": ab.norm.weight, "a.group_norm.bias": ab.norm.bias}
for a, b in (("to_q", ab.q), ("to_k", ab.k), ("to_v", ab.v), ("to_out.0", ab.proj_out)):
    sd[f"a.{a}.weight"], sd[f"a.{a}.bias"] = b.weight[:, :, 0, 0], b.bias
xa = torch.randn(1, 512, 8, 8)
with torch.no_grad():
    ok("AttnBlock(512)", vae.mid_attention(sd, "a", xa, 32, 1e-6), ab(xa), 1e-5)
ds = Downsample(64, True)
xd = torch.randn(2, 64, 10, 10)
with torch.no_grad():
    mine = torch.nn.functional.conv2d(torch.nn.functional.pad(xd, (0, 1, 0, 1)), ds.conv.weight, ds.conv.bias, stride=2)
    ok("Downsample pad(0,1,0,1)+s2", mine, ds(xd), 1e-6)
golden["ds_x"], golden["ds_w"], golden["ds_b"] = xd.numpy(), ds.conv.weight.detach().numpy(), ds.conv.bias.detach().numpy()
golden["ds_y"] = ds(xd).detach().numpy()

np.savez_compressed(os.path.join(GOLD, "vae_twin.npz"), **golden)
with open(os.path.join(GOLD, "vae_twin.report.txt"), "w") as f:
    f.write("\n".join(report) + "\n")
print("all VAE twin checks passed; fixtures ->", GOLD)
