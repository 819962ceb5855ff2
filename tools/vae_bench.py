#!/usr/bin/env python3
"""Time the VAE encode in front of the step on one MI355X: B x 3 x R x R pixels -> scaled latents.
Prints ms per batch, images/s, achieved TFLOP/s on the encoder's algorithmic MACs, and the GEMM launches by class."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
from pdm import _pdmk as k  # noqa: E402
from pdm.models.vae.autoencoder_kl import AutoencoderKL  # noqa: E402


def encoder_macs(cfg, R):
    """Algorithmic multiply-accumulates of one image (convs, 1x1 shortcuts, attention projections and contractions)."""
    ch, L = cfg.block_out_channels, cfg.layers_per_block
    hw, macs, cin = R * R, 0, ch[0]
    macs += hw * 9 * cfg.in_channels * ch[0]
    for i, co in enumerate(ch):
        for _ in range(L):
            macs += hw * 9 * (cin * co + co * co) + (hw * cin * co if cin != co else 0)
            cin = co
        if i != len(ch) - 1:
            hw //= 4
            macs += hw * 9 * co * co
    macs += 2 * hw * 9 * 2 * cin * cin                       # two mid ResBlocks
    macs += 4 * hw * cin * cin + 2 * hw * hw * cin           # q, k, v, out projections + QK^T + PV
    macs += hw * 9 * cin * 2 * cfg.latent_channels + hw * (2 * cfg.latent_channels) ** 2
    return macs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = AutoencoderKL(None, dev, torch.bfloat16, seed=0)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand(a.batch, 3, a.res, a.res, device=dev, generator=g) * 2 - 1
    for _ in range(2):                                       # first call tunes the GEMM plans
        z = m.encode_latents(x, generator=g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        z = m.encode_latents(x, generator=g)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    macs = encoder_macs(m.cfg, a.res) * a.batch
    print(f"vae encode B={a.batch} {a.res}x{a.res} bf16: {ms:.2f} ms/batch, {a.batch / ms * 1e3:.1f} img/s, "
          f"{2 * macs / ms / 1e9:.1f} TFLOP/s on {macs / a.batch / 1e9:.1f} GMAC/img; peak mem "
          f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; latents std {z.std().item():.3f}")
    if a.profile:
        k.PROFILE = []
        m.encode_latents(x, generator=g)
        torch.cuda.synchronize()
        agg = {}
        for kind, fl, e0, e1, shape in k.PROFILE:
            name = k.candidate_name(kind[1], kind[2], kind[3])
            t = e0.elapsed_time(e1)
            d = agg.setdefault((name, shape), [0.0, 0.0, 0])
            d[0] += t; d[1] += fl; d[2] += 1
        k.PROFILE = None
        tot = sum(v[0] for v in agg.values())
        print(f"GEMM launches: {tot:.2f} ms of {ms:.2f}")
        for (name, shape), (t, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
            print(f"  {t:7.3f} ms x{n:2d} {fl / t / 1e9:7.1f} TF/s  M,N,K,sk={shape}  {name}")


if __name__ == "__main__":
    main()
