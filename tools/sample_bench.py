#!/usr/bin/env python3
"""Time the image-logging / FID sampler on one MI355X: B prompts, PNDM, CFG 7.5, 512 x 512, bf16, pruned
student (MAC budget 0.55) + VAE decode.  Reports s / batch, images/s and the U-Net forward rate."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
from pdm.models.unet.spec import UNetConfig, arch_vector_for_budget, plan_macs  # noqa: E402
from pdm.models.unet.unet_2d_conditional import UNet2DConditionModelPruned  # noqa: E402
from pdm.models.vae.autoencoder_kl import AutoencoderKL  # noqa: E402
from pdm.models.clip.text_encoder import CLIPTextModel  # noqa: E402
from pdm.pipelines.pruning_pipelines import StableDiffusionPruningPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--budget", type=float, default=0.55)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = UNetConfig.sd21()
    av, ratio, _ = arch_vector_for_budget(cfg, a.budget, hw=64)
    unet = UNet2DConditionModelPruned(cfg, av, dev, torch.bfloat16, train=False, seed=0)
    vae = AutoencoderKL(None, dev, torch.bfloat16, seed=0)
    txt = CLIPTextModel(None, dev, torch.bfloat16, seed=0)
    pipe = StableDiffusionPruningPipeline(vae, txt, unet)
    ids = torch.randint(0, 49408, (a.batch, 77), device=dev)
    empty = torch.zeros(a.batch, 77, dtype=torch.int64, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    pipe(prompt_ids=ids, negative_prompt_ids=empty, num_inference_steps=2, generator=gen)          # warm-up: GEMM plans
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = pipe(prompt_ids=ids, negative_prompt_ids=empty, num_inference_steps=a.steps, generator=gen, output_type="pt").images
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    macs = plan_macs(cfg, unet.blocks, 64, 77)[0]
    calls = a.steps + 1
    print(f"sampler B={a.batch} {a.steps} PNDM steps ({calls} U-Net calls on 2B), CFG 7.5, 512x512, bf16, budget {ratio:.3f}: "
          f"{el:.2f} s/batch = {a.batch / el:.2f} img/s; U-Net {2 * macs * 2 * a.batch * calls / el / 1e12:.0f} TFLOP/s incl. "
          f"text encode + VAE decode; image range [{img.min().item():.2f}, {img.max().item():.2f}]")


if __name__ == "__main__":
    main()
