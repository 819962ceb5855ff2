"""Static shape table ("plan") of the gated / pruned SD-2.1 U-Net for the MI355X engine.

What the reference does dynamically with nn.Modules (build dense gated model, set_structure, prune() every module:
pdm/models/unet/unet_2d_conditional.py:718-1175, 1334-1415, 2448-2459; blocks.py prune methods) is resolved here ONCE
into a flat list of layer records with concrete, 8-aligned widths, so the executor can pre-plan every kernel launch.

Arch vector layout (pdm/models/hypernet.py:100-126): for each block in (down0..3, mid, up0..3): one 32-wide entry per
ResBlock, then per transformer [heads, heads, ff_gate_width]; then 14 depth scalars in the same block order (only the
last ResBlock / transformer of each down/up block is depth-gated).  keep <=> value >= 0.5
(pdm/utils/estimation_utils.py:67-75).
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch

HEAD_DIM = 64


CHANNEL_PAD = 32     # every channel / feature dim is zero-padded to a multiple of 32: one thread of the GEMM loaders
                     # moves 64 contiguous bytes (32 bf16) of a tile row, so rows and conv taps are 64-byte granular


def padc(n):
    return (n + CHANNEL_PAD - 1) // CHANNEL_PAD * CHANNEL_PAD


@dataclass(frozen=True)
class UNetConfig:
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    cross_attention_dim: int = 1024
    layers_per_block: int = 2
    norm_num_groups: int = 32
    in_channels: int = 4
    out_channels: int = 4
    ff_gate_width: int = 32
    attn_stages_down: Tuple[bool, ...] = (True, True, True, False)   # CrossAttnDownBlock2D*, ..., DownBlock2D*
    attn_stages_up: Tuple[bool, ...] = (False, True, True, True)

    @property
    def temb_dim(self):
        return 4 * self.block_out_channels[0]

    def heads(self, c):
        assert c % HEAD_DIM == 0, "the MI355X attention kernels are specialised for head dim 64 (SD-2.x)"
        return c // HEAD_DIM

    @staticmethod
    def sd21():
        return UNetConfig()

    @staticmethod
    def tiny():
        return UNetConfig(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64)


@dataclass
class ResSpec:
    name: str
    cin: int                 # input channels (after skip concat on the up path)
    cout: int
    skip: int = 0            # channels that come from the skip tensor (up path), 0 otherwise
    depth_gated: bool = False
    keep_mask: Optional[torch.Tensor] = None   # bool [32] over norm2 groups (None = all kept)
    dropped: bool = False

    def groups2(self, G):
        return G if self.keep_mask is None else int(self.keep_mask.sum())

    def inner(self, G):      # real inner width C'
        return self.groups2(G) * (self.cout // G)


@dataclass
class AttnSpec:
    name: str
    c: int
    heads: int
    depth_gated: bool = False
    keep_h1: Optional[torch.Tensor] = None     # bool [heads]
    keep_h2: Optional[torch.Tensor] = None
    keep_ff: Optional[torch.Tensor] = None     # bool [ff_gate_width]
    dropped: bool = False

    def h1(self):
        return self.heads if self.keep_h1 is None else int(self.keep_h1.sum())

    def h2(self):
        return self.heads if self.keep_h2 is None else int(self.keep_h2.sum())

    def ff(self, gate_width):
        unit = 4 * self.c // gate_width
        return 4 * self.c if self.keep_ff is None else int(self.keep_ff.sum()) * unit


@dataclass
class BlockSpec:
    name: str
    kind: str                # "down" | "mid" | "up"
    idx: int
    c: int
    resnets: List[ResSpec] = field(default_factory=list)
    attns: List[AttnSpec] = field(default_factory=list)
    sampler: bool = False


def build_blocks(cfg: UNetConfig) -> List[BlockSpec]:
    ch = cfg.block_out_channels
    n = len(ch)
    L = cfg.layers_per_block
    blocks = []
    prev = ch[0]
    for i, c in enumerate(ch):
        b = BlockSpec(f"down_blocks.{i}", "down", i, c, sampler=i < n - 1)
        for j in range(L):
            b.resnets.append(ResSpec(f"{b.name}.resnets.{j}", prev if j == 0 else c, c, depth_gated=j == L - 1))
            if cfg.attn_stages_down[i]:
                b.attns.append(AttnSpec(f"{b.name}.attentions.{j}", c, cfg.heads(c), depth_gated=j == L - 1))
        prev = c
        blocks.append(b)
    c = ch[-1]
    mid = BlockSpec("mid_block", "mid", 0, c)
    mid.resnets = [ResSpec("mid_block.resnets.0", c, c), ResSpec("mid_block.resnets.1", c, c)]
    mid.attns = [AttnSpec("mid_block.attentions.0", c, cfg.heads(c))]
    blocks.append(mid)
    rev = ch[::-1]
    prev = rev[0]
    for i, c in enumerate(rev):
        skip_last = rev[min(i + 1, n - 1)]
        b = BlockSpec(f"up_blocks.{i}", "up", i, c, sampler=i < n - 1)
        for j in range(L + 1):
            last = j == L
            skip = skip_last if last else c
            b.resnets.append(ResSpec(f"{b.name}.resnets.{j}", (prev if j == 0 else c) + skip, c, skip=skip,
                                     depth_gated=last))
            if cfg.attn_stages_up[i]:
                b.attns.append(AttnSpec(f"{b.name}.attentions.{j}", c, cfg.heads(c), depth_gated=last))
        prev = c
        blocks.append(b)
    return blocks


def gate_structure(cfg: UNetConfig):
    """Same nested lists as UNet2DConditionModelGated.get_structure() (unet_2d_conditional.py:1334-1364)."""
    width, depth = [], []
    for b in build_blocks(cfg):
        for r in b.resnets:
            width.append([cfg.norm_num_groups])
            depth.append([int(r.depth_gated)])
        for a in b.attns:
            width.append([a.heads, a.heads, cfg.ff_gate_width])
            depth.append([int(a.depth_gated)])
    return {"width": width, "depth": depth}


def arch_vector_size(cfg):
    s = gate_structure(cfg)
    return sum(map(sum, s["width"])) + sum(map(sum, s["depth"]))


def transform_arch_vector(inputs, structure):
    """HyperStructure.transform_arch_vector (pdm/models/hypernet.py:100-126)."""
    wl = [w for sub in structure["width"] for w in sub]
    dl = [d for sub in structure["depth"] for d in sub]
    assert inputs.shape[1] == sum(wl) + sum(dl), f"arch vector has {inputs.shape[1]} entries, expected {sum(wl) + sum(dl)}"
    wv, dv = inputs[:, :sum(wl)], inputs[:, sum(wl):]
    ws, start = [], 0
    for w in wl:
        ws.append(wv[:, start:start + w])
        start += w
    return {"width": ws, "depth": [dv[:, i] for i in range(sum(dl))]}


def get_random_arch_vector(target_ratio, structure, generator=None):
    """HyperStructure.get_random_arch_vector (pdm/models/hypernet.py:128-150)."""
    parts = []
    for sub in structure["width"]:
        for w in sub:
            v = torch.zeros(1, w)
            idx = torch.randperm(w, generator=generator)[: int(target_ratio * w)]
            v[0, idx] = 0.9
            parts.append(v)
    for sub in structure["depth"]:
        for d in sub:
            if d:
                parts.append(torch.tensor([[0.9]]))
    return torch.cat(parts, dim=1)


def force_width_non_zero(arch_vector, structure):
    """transform_arch_vector(..., force_width_non_zero=True) (hypernet.py:112-118): a gate with no kept unit gets its
    first unit raised by 0.5 so no layer ends up with zero width."""
    av = arch_vector.clone()
    start = 0
    for sub in structure["width"]:
        for w in sub:
            if not (av[0, start:start + w] >= 0.5).any():
                av[0, start] += 0.5
            start += w
    return av


def hard_concrete(x):
    """pdm/utils/estimation_utils.py:67-75 (forward value): 1 where x >= 0.5 else 0."""
    return (x >= 0.5).to(torch.float32)


def apply_arch_vector(cfg: UNetConfig, arch_vector) -> List[BlockSpec]:
    """set_structure + prune (unet_2d_conditional.py:1366-1415, 2448-2459): returns blocks with keep masks."""
    blocks = build_blocks(cfg)
    if arch_vector is None:
        return blocks
    av = transform_arch_vector(arch_vector.detach().float().cpu(), gate_structure(cfg))
    wq, dq = list(av["width"]), list(av["depth"])
    for b in blocks:
        got = [(r, [wq.pop(0)]) for r in b.resnets] + [(a, [wq.pop(0) for _ in range(3)]) for a in b.attns]
        for ent, ws in got:
            if ent.depth_gated and not bool(hard_concrete(dq.pop(0))[0]):
                ent.dropped = True
            masks = [hard_concrete(w)[0].bool() for w in ws]
            if isinstance(ent, ResSpec):
                ent.keep_mask = masks[0]
            else:
                ent.keep_h1, ent.keep_h2, ent.keep_ff = masks
                assert masks[0].any() and masks[1].any(), "attention with zero heads (blocks.py:165 asserts > 0)"
    assert not wq and not dq
    return blocks


def plan_macs(cfg: UNetConfig, blocks, hw: int, ctx_len: int):
    """Exact forward multiply-accumulates per image of the (pruned) plan at latent side `hw`: 3x3/1x1 convs, every
    Linear, QK^T and PV of both attentions (logical, un-padded widths).  Returns (total, by_kind dict)."""
    G = cfg.norm_num_groups
    by = {"conv3x3": 0, "conv1x1": 0, "linear": 0, "sdpa": 0}
    c0 = cfg.block_out_channels[0]
    by["conv3x3"] += hw * hw * 9 * cfg.in_channels * c0
    by["linear"] += c0 * cfg.temb_dim + cfg.temb_dim * cfg.temb_dim
    side = hw

    def res(r, px):
        if r.dropped:
            return
        ci = r.inner(G)
        by["conv3x3"] += px * 9 * (r.cin * ci + ci * r.cout)
        by["linear"] += cfg.temb_dim * ci
        if r.cin != r.cout:
            by["conv1x1"] += px * r.cin * r.cout

    def att(a, px):
        if a.dropped:
            return
        c, d1, d2, ff = a.c, a.h1() * 64, a.h2() * 64, a.ff(cfg.ff_gate_width)
        by["linear"] += px * (2 * c * c + 4 * c * d1 + 2 * c * d2 + 3 * c * ff) + 2 * ctx_len * cfg.cross_attention_dim * d2
        by["sdpa"] += 2 * px * px * d1 + 2 * px * ctx_len * d2

    for b in blocks:
        px = side * side
        if b.kind == "mid":
            res(b.resnets[0], px); att(b.attns[0], px); res(b.resnets[1], px)
            continue
        for j, r in enumerate(b.resnets):
            res(r, px)
            if b.attns:
                att(b.attns[j], px)
        if b.sampler:
            side = side // 2 if b.kind == "down" else side * 2
            by["conv3x3"] += side * side * 9 * b.c * b.c
    by["conv3x3"] += hw * hw * 9 * c0 * cfg.out_channels
    return sum(by.values()), by


def arch_vector_for_budget(cfg: UNetConfig, budget, hw=64, ctx_len=77, seed=0, tol=0.004):
    """Random arch vector (get_random_arch_vector, all depth gates kept) whose MAC ratio student/teacher - the
    reference's "Pruning Ratio" (trainer.py:2183) - is `budget`, found by bisection on the per-gate keep ratio."""
    dense, _ = plan_macs(cfg, build_blocks(cfg), hw, ctx_len)
    lo, hi = 0.05, 1.0
    best = None
    for _ in range(24):
        mid = 0.5 * (lo + hi)
        av = force_width_non_zero(get_random_arch_vector(mid, gate_structure(cfg), torch.Generator().manual_seed(seed)),
                                  gate_structure(cfg))
        ratio = plan_macs(cfg, apply_arch_vector(cfg, av), hw, ctx_len)[0] / dense
        if best is None or abs(ratio - budget) < abs(best[1] - budget):
            best = (av, ratio, mid)
        if abs(ratio - budget) <= tol:
            break
        if ratio < budget:
            lo = mid
        else:
            hi = mid
    return best
