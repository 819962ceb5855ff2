"""Dense weight construction (diffusers SD U-Net state-dict names) and physical pruning (oracle side).

Pruning follows the reference's prune() methods literally:
  ResnetBlock2DWidth[Depth]Gated.prune      pdm/models/unet/blocks.py:434-475, 646-702
  GatedAttention.prune                      blocks.py:162-196   (head slices of to_q/k/v rows, to_out columns)
  GEGLUGated.prune_gate / FeedForward prune blocks.py:62-76, 130-138
  Transformer2DModelWidthDepthGated.prune_module  blocks.py:1323-1334 (dropped => identity)
Weights use PyTorch default initialisers (the reference's random_init path, unet_2d_conditional.py:2406-2408).
"""
import math
import torch

from .arch import block_layout, assign_gates
from .config import UNetConfig


def _linear(sd, name, cin, cout, g, bias=True):
    bound = 1.0 / math.sqrt(cin)
    sd[name + ".weight"] = (torch.rand(cout, cin, generator=g) * 2 - 1) * bound
    if bias:
        sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound


def _conv(sd, name, cin, cout, k, g):
    bound = 1.0 / math.sqrt(cin * k * k)
    sd[name + ".weight"] = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound
    sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound


def _norm(sd, name, c, g, jitter=0.1):
    # ones/zeros like nn.GroupNorm/LayerNorm, plus a small seeded jitter so affine terms are exercised
    sd[name + ".weight"] = 1.0 + jitter * (torch.rand(c, generator=g) * 2 - 1)
    sd[name + ".bias"] = jitter * (torch.rand(c, generator=g) * 2 - 1)


def init_dense_state_dict(cfg: UNetConfig, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    c0 = cfg.block_out_channels[0]
    _conv(sd, "conv_in", cfg.in_channels, c0, 3, g)
    _linear(sd, "time_embedding.linear_1", c0, cfg.temb_dim, g)
    _linear(sd, "time_embedding.linear_2", cfg.temb_dim, cfg.temb_dim, g)
    for b in block_layout(cfg):
        for r in b["resnets"]:
            p = r["prefix"]
            _norm(sd, p + ".norm1", r["cin"], g)
            _conv(sd, p + ".conv1", r["cin"], r["cout"], 3, g)
            _linear(sd, p + ".time_emb_proj", cfg.temb_dim, r["cout"], g)
            _norm(sd, p + ".norm2", r["cout"], g)
            _conv(sd, p + ".conv2", r["cout"], r["cout"], 3, g)
            if r["cin"] != r["cout"]:
                _conv(sd, p + ".conv_shortcut", r["cin"], r["cout"], 1, g)
        for a in b["attns"]:
            p, c = a["prefix"], a["c"]
            _norm(sd, p + ".norm", c, g)
            _linear(sd, p + ".proj_in", c, c, g)
            t = p + ".transformer_blocks.0"
            _norm(sd, t + ".norm1", c, g)
            for nm in ("to_q", "to_k", "to_v"):
                _linear(sd, f"{t}.attn1.{nm}", c, c, g, bias=False)
            _linear(sd, f"{t}.attn1.to_out.0", c, c, g)
            _norm(sd, t + ".norm2", c, g)
            _linear(sd, f"{t}.attn2.to_q", c, c, g, bias=False)
            _linear(sd, f"{t}.attn2.to_k", cfg.cross_attention_dim, c, g, bias=False)
            _linear(sd, f"{t}.attn2.to_v", cfg.cross_attention_dim, c, g, bias=False)
            _linear(sd, f"{t}.attn2.to_out.0", c, c, g)
            _norm(sd, t + ".norm3", c, g)
            _linear(sd, f"{t}.ff.net.0.proj", c, 8 * c, g)
            _linear(sd, f"{t}.ff.net.2", 4 * c, c, g)
            _linear(sd, p + ".proj_out", c, c, g)
        if b["sampler"]:
            nm = "downsamplers" if b["kind"] == "down" else "upsamplers"
            _conv(sd, f"{b['name']}.{nm}.0.conv", b["c"], b["c"], 3, g)
    _norm(sd, "conv_norm_out", c0, g)
    _conv(sd, "conv_out", c0, cfg.out_channels, 3, g)
    return sd


def prune_state_dict(sd, cfg: UNetConfig, arch_vector):
    """Returns (pruned_sd, info) with info[prefix] = dict(dropped=bool, heads1=int, heads2=int, ...)."""
    gates = assign_gates(arch_vector, cfg)
    out = dict(sd)
    info = {}
    G = cfg.norm_num_groups
    for b in block_layout(cfg):
        for r in b["resnets"]:
            p = r["prefix"]
            gt = gates[p]
            if not gt["keep"]:
                for k in [k for k in out if k.startswith(p + ".")]:
                    del out[k]
                info[p] = dict(dropped=True, skip_dim=r["skip_dim"])
                continue
            gh = gt["width"][0]
            gs = r["cout"] // G
            m = gh.repeat_interleave(gs).bool()
            out[p + ".conv1.weight"] = sd[p + ".conv1.weight"][m]
            out[p + ".conv1.bias"] = sd[p + ".conv1.bias"][m]
            out[p + ".time_emb_proj.weight"] = sd[p + ".time_emb_proj.weight"][m]
            out[p + ".time_emb_proj.bias"] = sd[p + ".time_emb_proj.bias"][m]
            out[p + ".norm2.weight"] = sd[p + ".norm2.weight"][m]
            out[p + ".norm2.bias"] = sd[p + ".norm2.bias"][m]
            out[p + ".conv2.weight"] = sd[p + ".conv2.weight"][:, m]
            info[p] = dict(dropped=False, groups2=int(gh.sum().item()))
        for a in b["attns"]:
            p, c, H = a["prefix"], a["c"], a["heads"]
            gt = gates[p]
            if not gt["keep"]:
                for k in [k for k in out if k.startswith(p + ".")]:
                    del out[k]
                info[p] = dict(dropped=True)
                continue
            t = p + ".transformer_blocks.0"
            hd = c // H
            hs = []
            for an, gh in (("attn1", gt["width"][0]), ("attn2", gt["width"][1])):
                hm = gh.bool()
                assert hm.sum() > 0
                for nm in ("to_q", "to_k", "to_v"):
                    w = sd[f"{t}.{an}.{nm}.weight"]
                    out[f"{t}.{an}.{nm}.weight"] = w.view(H, hd, w.shape[1])[hm].reshape(-1, w.shape[1])
                w = sd[f"{t}.{an}.to_out.0.weight"]
                out[f"{t}.{an}.to_out.0.weight"] = w.view(w.shape[0], H, hd)[:, hm].reshape(w.shape[0], -1)
                hs.append(int(hm.sum().item()))
            gf = gt["width"][2]
            inner = 4 * c
            fm = gf.repeat_interleave(inner // cfg.ff_gate_width).bool()
            fm2 = torch.cat([fm, fm])
            out[f"{t}.ff.net.0.proj.weight"] = sd[f"{t}.ff.net.0.proj.weight"][fm2]
            out[f"{t}.ff.net.0.proj.bias"] = sd[f"{t}.ff.net.0.proj.bias"][fm2]
            out[f"{t}.ff.net.2.weight"] = sd[f"{t}.ff.net.2.weight"][:, fm]
            info[p] = dict(dropped=False, heads1=hs[0], heads2=hs[1], ff=int(fm.sum().item()))
    return out, info


def dense_info(cfg: UNetConfig):
    info = {}
    for b in block_layout(cfg):
        for r in b["resnets"]:
            info[r["prefix"]] = dict(dropped=False, groups2=cfg.norm_num_groups)
        for a in b["attns"]:
            info[a["prefix"]] = dict(dropped=False, heads1=a["heads"], heads2=a["heads"], ff=4 * a["c"])
    return info
