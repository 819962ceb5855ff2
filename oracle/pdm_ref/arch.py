"""Arch-vector layout and gate semantics (oracle side).

Follows pdm/models/hypernet.py:100-150 (transform_arch_vector / get_random_arch_vector),
pdm/utils/estimation_utils.py:67-75 (hard_concrete: keep iff value >= 0.5) and the structure walk of
pdm/models/unet/unet_2d_conditional.py:1334-1415 + the block containers' get_gate_structure
(pdm/models/unet/blocks.py:1710-1727): per block all ResBlocks first, then all transformers, each transformer
contributing [heads, heads, ff_gate_width] (blocks.py:782-788).
"""
import torch

from .config import UNetConfig


def hard_concrete(x):
    return (x >= 0.5).to(torch.float32)


def block_layout(cfg: UNetConfig):
    """List of blocks in structure order; each = dict(name, resnets=[...], attns=[...]) where every entry carries
    its state-dict prefix, channel info and whether it is depth-prunable."""
    ch = cfg.block_out_channels
    n = len(ch)
    blocks = []
    # ---- down
    out = ch[0]
    for i in range(n):
        inp, out = out, ch[i]
        res, att = [], []
        for j in range(cfg.layers_per_block):
            last = j == cfg.layers_per_block - 1
            res.append(dict(prefix=f"down_blocks.{i}.resnets.{j}", cin=inp if j == 0 else out, cout=out,
                            depth=last, skip_dim=None))
            if cfg.down_has_attn[i]:
                att.append(dict(prefix=f"down_blocks.{i}.attentions.{j}", c=out, heads=cfg.heads[i], depth=last))
        blocks.append(dict(name=f"down_blocks.{i}", kind="down", idx=i, resnets=res, attns=att,
                           sampler=(i != n - 1), c=out))
    # ---- mid (width gated only, blocks.py:2486-2541)
    c = ch[-1]
    blocks.append(dict(name="mid_block", kind="mid", idx=0, c=c, sampler=False,
                       resnets=[dict(prefix="mid_block.resnets.0", cin=c, cout=c, depth=False, skip_dim=None),
                                dict(prefix="mid_block.resnets.1", cin=c, cout=c, depth=False, skip_dim=None)],
                       attns=[dict(prefix="mid_block.attentions.0", c=c, heads=cfg.heads[-1], depth=False)]))
    # ---- up
    rev = list(reversed(ch))
    rheads = list(reversed(cfg.heads))
    out = rev[0]
    for i in range(n):
        prev, out = out, rev[i]
        inp = rev[min(i + 1, n - 1)]
        res, att = [], []
        nl = cfg.layers_per_block + 1
        for j in range(nl):
            last = j == nl - 1
            skip = inp if last else out
            rin = prev if j == 0 else out
            res.append(dict(prefix=f"up_blocks.{i}.resnets.{j}", cin=rin + skip, cout=out, depth=last, skip_dim=skip))
            if cfg.up_has_attn[i]:
                att.append(dict(prefix=f"up_blocks.{i}.attentions.{j}", c=out, heads=rheads[i], depth=last))
        blocks.append(dict(name=f"up_blocks.{i}", kind="up", idx=i, resnets=res, attns=att,
                           sampler=(i != n - 1), c=out))
    return blocks


def structure(cfg: UNetConfig):
    """{'width': [[...], ...], 'depth': [[...], ...]} exactly as get_structure() builds it."""
    width, depth = [], []
    for b in block_layout(cfg):
        for r in b["resnets"]:
            width.append([cfg.norm_num_groups])
            depth.append([1 if r["depth"] else 0])
        for a in b["attns"]:
            width.append([a["heads"], a["heads"], cfg.ff_gate_width])
            depth.append([1 if a["depth"] else 0])
    return {"width": width, "depth": depth}


def arch_vector_len(cfg):
    s = structure(cfg)
    return sum(sum(w) for w in s["width"]) + sum(sum(d) for d in s["depth"])


def random_arch_vector(cfg, keep, seed=0, drop_depth=()):
    """hypernet.py:128-150: per gate, 0.9 at int(keep*w) random positions; depth entries 0.9 (all kept).
    ``drop_depth`` = indices (into the depth tail) forced to 0.0 (a test variant)."""
    g = torch.Generator().manual_seed(seed)
    s = structure(cfg)
    parts = []
    for sub in s["width"]:
        for w in sub:
            v = torch.zeros(1, w)
            k = int(keep * w)
            idx = torch.randperm(w, generator=g)[:k]
            v[0, idx] = 0.9
            if k == 0:
                v[0, 0] = 0.9          # force_width_non_zero (hypernet.py:113-118)
            parts.append(v)
    nd = sum(sum(d) for d in s["depth"])
    for i in range(nd):
        parts.append(torch.tensor([[0.0 if i in drop_depth else 0.9]]))
    return torch.cat(parts, dim=1)


def transform_arch_vector(vec, cfg):
    """hypernet.py:100-126 -> flat lists of width sub-vectors and depth scalars, in structure order."""
    s = structure(cfg)
    wl = [w for sub in s["width"] for w in sub]
    nd = sum(sum(d) for d in s["depth"])
    assert vec.shape[1] == sum(wl) + nd, (vec.shape, sum(wl), nd)
    out_w, start = [], 0
    for w in wl:
        out_w.append(vec[:, start:start + w])
        start += w
    out_d = [vec[:, start + i] for i in range(nd)]
    return {"width": out_w, "depth": out_d}


def assign_gates(vec, cfg):
    """Walk set_structure (unet_2d_conditional.py:1366-1415 + blocks.py:1729-1760): returns
    {prefix: dict(width=[hard masks...], keep=bool)} for every resnet / transformer."""
    av = transform_arch_vector(vec, cfg)
    wq, dq = list(av["width"]), list(av["depth"])
    gates = {}
    for b in block_layout(cfg):
        # widths are popped for all entries of the block first (resnets then attentions) ...
        pending = []
        for r in b["resnets"]:
            pending.append((r, [wq.pop(0)]))
        for a in b["attns"]:
            pending.append((a, [wq.pop(0), wq.pop(0), wq.pop(0)]))
        # ... then depths in the same order for entries whose flag is 1
        for ent, ws in pending:
            keep = True
            if ent["depth"]:
                keep = bool(hard_concrete(dq.pop(0))[0] >= 0.5)
            gates[ent["prefix"]] = dict(width=[hard_concrete(w)[0] for w in ws], keep=keep)
    assert not wq and not dq
    return gates
