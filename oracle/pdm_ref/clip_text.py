"""Functional fp32 CLIP text transformer on a state dict (oracle side, plain torch) - SURVEY 8f row N2.

What the reference runs inside its dataset transform (pdm/utils/data_utils.py:155-191, 247-276):
    prompt_embeds = text_encoder(text_input_ids)[0]          # last_hidden_state, [B, 77, 1024]
with `transformers.CLIPTextModel` loaded from stabilityai/stable-diffusion-2-1 `text_encoder/` (trainer.py:2126-2131):
OpenCLIP ViT-H text tower with the last layer dropped - hidden 1024, 23 layers, 16 heads (head dim 64), MLP 4096 with
erf-GELU, learned absolute positions (77), pre-LN blocks, causal mask, final LayerNorm, eps 1e-5.
The arithmetic lives in the third-party dependency `transformers` (reference pin: env.yaml; importable in this container
as 5.15.0): `oracle/validate_clip_against_transformers.py` pins this restatement against that class itself.
State-dict keys are the transformers ones, with or without the `text_model.` prefix.
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 23
    num_attention_heads: int = 16
    max_position_embeddings: int = 77
    layer_norm_eps: float = 1e-5

    @staticmethod
    def sd21():
        return CLIPTextConfig()

    @staticmethod
    def tiny():
        return CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                              num_attention_heads=2)


def _get(sd, key):
    return sd[key] if key in sd else sd["text_model." + key]


def encode(sd, cfg: CLIPTextConfig, input_ids):
    """input_ids [B, T<=77] int64 -> last_hidden_state [B, T, hidden] (after final_layer_norm)."""
    B, T = input_ids.shape
    H, D = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    x = _get(sd, "embeddings.token_embedding.weight")[input_ids] + _get(sd, "embeddings.position_embedding.weight")[:T]
    mask = torch.full((T, T), float("-inf")).triu(1)
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}"
        h = F.layer_norm(x, (cfg.hidden_size,), _get(sd, p + ".layer_norm1.weight"), _get(sd, p + ".layer_norm1.bias"),
                         cfg.layer_norm_eps)
        q, k, v = (F.linear(h, _get(sd, f"{p}.self_attn.{n}_proj.weight"), _get(sd, f"{p}.self_attn.{n}_proj.bias"))
                   .view(B, T, H, D).transpose(1, 2) for n in ("q", "k", "v"))
        s = torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5 + mask, dim=-1)
        o = (s @ v).transpose(1, 2).reshape(B, T, cfg.hidden_size)
        x = x + F.linear(o, _get(sd, p + ".self_attn.out_proj.weight"), _get(sd, p + ".self_attn.out_proj.bias"))
        h = F.layer_norm(x, (cfg.hidden_size,), _get(sd, p + ".layer_norm2.weight"), _get(sd, p + ".layer_norm2.bias"),
                         cfg.layer_norm_eps)
        h = F.gelu(F.linear(h, _get(sd, p + ".mlp.fc1.weight"), _get(sd, p + ".mlp.fc1.bias")))
        x = x + F.linear(h, _get(sd, p + ".mlp.fc2.weight"), _get(sd, p + ".mlp.fc2.bias"))
    return F.layer_norm(x, (cfg.hidden_size,), _get(sd, "final_layer_norm.weight"), _get(sd, "final_layer_norm.bias"),
                        cfg.layer_norm_eps)


def init_state_dict(cfg: CLIPTextConfig, seed=0, prefix=""):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    E = cfg.hidden_size

    def lin(name, co, ci):
        sd[prefix + name + ".weight"] = torch.randn(co, ci, generator=g) * ci ** -0.5
        sd[prefix + name + ".bias"] = torch.randn(co, generator=g) * 0.02

    def norm(name):
        sd[prefix + name + ".weight"] = 1.0 + 0.2 * torch.randn(E, generator=g)
        sd[prefix + name + ".bias"] = 0.1 * torch.randn(E, generator=g)

    sd[prefix + "embeddings.token_embedding.weight"] = torch.randn(cfg.vocab_size, E, generator=g) * 0.5
    sd[prefix + "embeddings.position_embedding.weight"] = torch.randn(cfg.max_position_embeddings, E, generator=g) * 0.1
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}"
        norm(p + ".layer_norm1")
        for n in ("q", "k", "v", "out"):
            lin(f"{p}.self_attn.{n}_proj", E, E)
        norm(p + ".layer_norm2")
        lin(p + ".mlp.fc1", cfg.intermediate_size, E)
        lin(p + ".mlp.fc2", E, cfg.intermediate_size)
    norm("final_layer_norm")
    return sd
