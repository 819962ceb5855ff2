#!/usr/bin/env python3
"""Time the CLIP text encode in front of the step on one MI355X: B x 77 token ids -> prompt_embeds."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unlearn-ft_amd"))
from pdm.models.clip.text_encoder import CLIPTextModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = CLIPTextModel(None, dev, torch.bfloat16, seed=0)
    ids = torch.randint(0, m.cfg.vocab_size, (a.batch, 77), device=dev)
    for _ in range(3):
        y = m(ids)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = m(ids)[0]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    c = m.cfg
    E, Fd, T = c.hidden_size, c.intermediate_size, 77
    macs = a.batch * c.num_hidden_layers * (T * (4 * E * E + 2 * E * Fd) + 2 * T * T * E)
    wbytes = c.num_hidden_layers * (4 * E * E + 2 * E * Fd) * 2
    print(f"clip text encode B={a.batch} x 77 bf16: {ms:.3f} ms/batch, {a.batch / ms * 1e3:.0f} prompts/s, "
          f"{2 * macs / ms / 1e9:.1f} TFLOP/s; weights streamed {wbytes / 1e9:.2f} GB -> {wbytes / ms / 1e9:.2f} TB/s; "
          f"out std {y.float().std().item():.3f}")


if __name__ == "__main__":
    main()
