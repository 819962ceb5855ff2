"""Flat parameter arenas for the MI355X engine.

One fp32 master arena (+ grad arena, + AdamW moments owned by the optimiser) holds every parameter in its *packed*
kernel layout, in forward order:
  conv3x3   [Co_p][9 taps][Ci_p]            (diffusers OIHW -> O,ky,kx,I; channel dims zero-padded to multiples of 8)
  linear    [N_p][K_p]                      (q|k|v of self-attention and k|v of cross-attention fused row-wise)
  vectors   biases / norm affine, fp32 only
and two compute-dtype arenas with the same offsets: `w` (forward layout) and `wt` (dgrad layout: Linear W^T, conv
[Ci_p][9 flipped][Co_p]) refreshed from the master after every optimiser step.  Zero padding is a fixed point of the
training step (padded rows/cols receive zero gradients), so the packed model is exactly the pruned model.
State-dict import/export uses the diffusers SD U-Net key names with the *pruned* shapes, i.e. the checkpoint format of
trainer.py:314-346 (safetensors written by save_pretrained / read by load_state_dict).
"""
from dataclasses import dataclass
from typing import List, Tuple

import torch

from ... import _pdmk as k
from .spec import UNetConfig, padc


@dataclass
class Entry:
    key: str                              # arena key, e.g. "mid_block.resnets.0.conv1.weight"
    kind: str                             # conv3 | lin | vec
    shape: Tuple[int, ...]                # packed shape
    srcs: List[Tuple]                     # [(state-dict name, rows[, dst_row0, src_row0])]: row blocks of dim 0
    logical: Tuple[int, ...]              # logical (unpadded) shape of the concatenation: (n, k) | (co, ci) | (n,)
    off: int = 0

    def __post_init__(self):
        norm, r = [], 0
        for t in self.srcs:
            if len(t) == 2:
                norm.append((t[0], t[1], r, 0))
                r += t[1]
            else:
                norm.append(tuple(t))
                r = t[2] + t[1]
        self.srcs = norm

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


def _conv(key, co, ci):
    return Entry(key + ".weight", "conv3", (padc(co), 9, padc(ci)), [(key + ".weight", co)], (co, ci))


def _lin(key, srcs, k_in, suffix=".weight", rows_p=None):
    n = sum(t[1] for t in srcs)
    return Entry(key + suffix, "lin", (rows_p or padc(n), padc(k_in)), srcs, (n, k_in))


def _vec(key, srcs, rows_p=None):
    n = sum(t[1] for t in srcs)
    return Entry(key, "vec", (rows_p or padc(n),), srcs, (n,))


def _geglu_rows(name, ff):
    """GEGLU proj rows [hidden | gate] (blocks.py:55) packed INTERLEAVED in blocks of 8: hidden feature j sits in packed row
    16 (j // 8) + j % 8 and its gate 8 rows further, so that one 16-column group of the projection's output holds 8 hidden
    values and their 8 gates - what the GEMM's fused GEGLU epilogue (PDMK_EPI_GEGLU) and the layout-1 GEGLU kernels consume.
    The packed matrix has 2 padc(ff) rows (padding rows are zero: hidden 0 * gelu(0) = 0)."""
    out = []
    for q in range((ff + 7) // 8):
        n = min(8, ff - 8 * q)
        out += [(name, n, 16 * q, 8 * q), (name, n, 16 * q + 8, ff + 8 * q)]
    return out


def temb_layout(cfg: UNetConfig, blocks):
    """Column layout of the ONE batched time-embedding projection (all ResBlocks' time_emb_proj, blocks.py:334-341, share
    the input silu(temb)): {resblock name: (first column, padded width)} in forward order, and the total width."""
    G, off, lay = cfg.norm_num_groups, 0, {}
    for b in blocks:
        for r in b.resnets:
            if not r.dropped:
                lay[r.name] = (off, padc(r.inner(G)), r.inner(G))
                off += padc(r.inner(G))
    return lay, off


def kv_layout(blocks):
    """Column layout of the ONE batched cross-attention K/V projection: every transformer's attn2.to_k / to_v
    (blocks.py:244-285) reads the same prompt embeddings, so their weights are row blocks of one [sum 2 d2, ctx] matrix:
    {transformer name: (first column, d2)} with K at [off, off + d2) and V at [off + d2, off + 2 d2), and the total width."""
    off, lay = 0, {}
    for b in blocks:
        for a in b.attns:
            if not a.dropped:
                d2 = a.h2() * 64
                lay[a.name] = (off, d2)
                off += 2 * d2
    return lay, off


def build_entries(cfg: UNetConfig, blocks) -> List[Entry]:
    G = cfg.norm_num_groups
    E: List[Entry] = []
    c0 = cfg.block_out_channels[0]
    E += [_conv("conv_in", c0, cfg.in_channels), _vec("conv_in.bias", [("conv_in.bias", c0)])]
    for nm, (ki, no) in (("time_embedding.linear_1", (c0, cfg.temb_dim)),
                         ("time_embedding.linear_2", (cfg.temb_dim, cfg.temb_dim))):
        E += [_lin(nm, [(nm + ".weight", no)], ki), _vec(nm + ".bias", [(nm + ".bias", no)])]
    # every ResBlock's time_emb_proj as row blocks of one weight / bias: one GEMM forward, one dgrad, one wgrad per step
    lay, tot = temb_layout(cfg, blocks)
    if tot:
        E += [_lin("time_emb_proj_all", [(f"{n}.time_emb_proj.weight", ci, o, 0) for n, (o, cp, ci) in lay.items()],
                   cfg.temb_dim, rows_p=tot),
              _vec("time_emb_proj_all.bias", [(f"{n}.time_emb_proj.bias", ci, o, 0) for n, (o, cp, ci) in lay.items()],
                   rows_p=tot)]

    # every cross-attention K/V projection as row blocks of one weight: one GEMM forward and one weight gradient per step
    klay, ktot = kv_layout(blocks)
    if ktot:
        srcs = []
        for n, (o, d2) in klay.items():
            t = n + ".transformer_blocks.0"
            srcs += [(f"{t}.attn2.to_k.weight", d2, o, 0), (f"{t}.attn2.to_v.weight", d2, o + d2, 0)]
        E += [_lin("attn2_kv_all", srcs, cfg.cross_attention_dim, rows_p=ktot)]

    def res_entries(r):
        if r.dropped:
            return []
        p, ci = r.name, r.inner(G)
        out = [_vec(f"{p}.norm1.weight", [(f"{p}.norm1.weight", r.cin)]),
               _vec(f"{p}.norm1.bias", [(f"{p}.norm1.bias", r.cin)]),
               _conv(f"{p}.conv1", ci, r.cin), _vec(f"{p}.conv1.bias", [(f"{p}.conv1.bias", ci)]),
               _vec(f"{p}.norm2.weight", [(f"{p}.norm2.weight", ci)]),
               _vec(f"{p}.norm2.bias", [(f"{p}.norm2.bias", ci)]),
               _conv(f"{p}.conv2", r.cout, ci), _vec(f"{p}.conv2.bias", [(f"{p}.conv2.bias", r.cout)])]
        if r.cin != r.cout:
            out += [_lin(f"{p}.conv_shortcut", [(f"{p}.conv_shortcut.weight", r.cout)], r.cin),
                    _vec(f"{p}.conv_shortcut.bias", [(f"{p}.conv_shortcut.bias", r.cout)])]
        return out

    def attn_entries(a):
        if a.dropped:
            return []
        p, c = a.name, a.c
        t = p + ".transformer_blocks.0"
        d1, d2, ff = a.h1() * 64, a.h2() * 64, a.ff(cfg.ff_gate_width)
        out = [_vec(f"{p}.norm.weight", [(f"{p}.norm.weight", c)]), _vec(f"{p}.norm.bias", [(f"{p}.norm.bias", c)]),
               _lin(f"{p}.proj_in", [(f"{p}.proj_in.weight", c)], c), _vec(f"{p}.proj_in.bias", [(f"{p}.proj_in.bias", c)])]
        for i in (1, 2, 3):
            out += [_vec(f"{t}.norm{i}.weight", [(f"{t}.norm{i}.weight", c)]),
                    _vec(f"{t}.norm{i}.bias", [(f"{t}.norm{i}.bias", c)])]
        out += [_lin(f"{t}.attn1.to_qkv", [(f"{t}.attn1.to_q.weight", d1), (f"{t}.attn1.to_k.weight", d1),
                                            (f"{t}.attn1.to_v.weight", d1)], c),
                _lin(f"{t}.attn1.to_out.0", [(f"{t}.attn1.to_out.0.weight", c)], d1),
                _vec(f"{t}.attn1.to_out.0.bias", [(f"{t}.attn1.to_out.0.bias", c)]),
                _lin(f"{t}.attn2.to_q", [(f"{t}.attn2.to_q.weight", d2)], c),
                _lin(f"{t}.attn2.to_out.0", [(f"{t}.attn2.to_out.0.weight", c)], d2),
                _vec(f"{t}.attn2.to_out.0.bias", [(f"{t}.attn2.to_out.0.bias", c)]),
                _lin(f"{t}.ff.net.0.proj", _geglu_rows(f"{t}.ff.net.0.proj.weight", ff), c, rows_p=2 * padc(ff)),
                _vec(f"{t}.ff.net.0.proj.bias", _geglu_rows(f"{t}.ff.net.0.proj.bias", ff), rows_p=2 * padc(ff)),
                _lin(f"{t}.ff.net.2", [(f"{t}.ff.net.2.weight", c)], ff),
                _vec(f"{t}.ff.net.2.bias", [(f"{t}.ff.net.2.bias", c)]),
                _lin(f"{p}.proj_out", [(f"{p}.proj_out.weight", c)], c),
                _vec(f"{p}.proj_out.bias", [(f"{p}.proj_out.bias", c)])]
        return out

    for b in blocks:
        if b.kind == "mid":
            E += res_entries(b.resnets[0]) + attn_entries(b.attns[0]) + res_entries(b.resnets[1])
        else:
            for j, r in enumerate(b.resnets):
                E += res_entries(r)
                if b.attns:
                    E += attn_entries(b.attns[j])
        if b.sampler:
            nm = f"{b.name}.{'downsamplers' if b.kind == 'down' else 'upsamplers'}.0.conv"
            E += [_conv(nm, b.c, b.c), _vec(nm + ".bias", [(nm + ".bias", b.c)])]
    E += [_vec("conv_norm_out.weight", [("conv_norm_out.weight", c0)]),
          _vec("conv_norm_out.bias", [("conv_norm_out.bias", c0)]),
          _conv("conv_out", cfg.out_channels, c0), _vec("conv_out.bias", [("conv_out.bias", cfg.out_channels)])]
    off = 0
    for e in E:
        e.off = off
        off += (e.numel + 127) // 128 * 128          # every entry starts on a 256-byte (bf16) / 512-byte (fp32) boundary
    return E


def reference_param_order(names):
    """`names` (diffusers SD U-Net parameter names) in the order the reference module's `.parameters()` yields them, i.e.
    the index a torch optimizer state dict uses.  Registration order restated from the reference's constructors:
    pdm/models/unet/unet_2d_conditional.py:839-1167 (conv_in, time_embedding, down_blocks and up_blocks [both ModuleLists
    are created at :966-967, before mid_block at :1043], mid_block, conv_norm_out, conv_out); block containers register
    attentions before resnets before samplers (pdm/models/unet/blocks.py:1705-1706, 2038-2039, 2543-2544); a ResnetBlock2D
    registers norm1, conv1, time_emb_proj, norm2, conv2, conv_shortcut; Transformer2DModel norm, proj_in,
    transformer_blocks, proj_out; BasicTransformerBlock norm1, attn1, norm2, attn2, norm3, ff; Attention to_q, to_k, to_v,
    to_out; weight before bias (diffusers-resident)."""
    top = ("conv_in", "time_embedding", "down_blocks", "up_blocks", "mid_block", "conv_norm_out", "conv_out")
    sub = ("attentions", "resnets", "downsamplers", "upsamplers")
    leaf = ("norm1", "conv1", "time_emb_proj", "norm2", "conv2", "conv_shortcut",          # resnet
            "norm", "proj_in", "transformer_blocks", "proj_out",                            # transformer 2D
            "attn1", "attn2", "norm3", "ff",                                                # basic transformer block
            "to_q", "to_k", "to_v", "to_out", "net", "proj", "conv", "linear_1", "linear_2", "weight", "bias")
    # BasicTransformerBlock interleaves norms and attentions: norm1, attn1, norm2, attn2, norm3, ff
    tb = {"norm1": 0, "attn1": 1, "norm2": 2, "attn2": 3, "norm3": 4, "ff": 5}

    def key(name):
        parts = name.split(".")
        out = []
        in_tb = False
        for i, q in enumerate(parts):
            if q.isdigit():
                out.append(int(q))
            elif i == 0:
                out.append(top.index(q))
            elif q in sub:
                out.append(sub.index(q))
            elif in_tb and q in tb:
                out.append(tb[q])
                in_tb = False
            else:
                out.append(leaf.index(q))
            if q == "transformer_blocks":
                in_tb = True
        return out
    return sorted(names, key=key)


class ParamStore:
    """Owns the arenas of one U-Net replica.  `train=False` (teacher): master + forward copy only."""

    def __init__(self, entries: List[Entry], device, dtype, train=True):
        self.entries = entries
        self.by_key = {e.key: e for e in entries}
        self.total = entries[-1].off + (entries[-1].numel + 127) // 128 * 128
        self.dtype = dtype
        self.train = train
        self.master = torch.zeros(self.total, device=device, dtype=torch.float32)
        self.grad = torch.zeros(self.total, device=device, dtype=torch.float32) if train else None
        self.w = self.master if dtype == torch.float32 else torch.zeros(self.total, device=device, dtype=dtype)
        self.wt = torch.zeros(self.total, device=device, dtype=dtype) if train else None
        self.defer_wt = False          # set by the stepper: optimiser steps skip the wt refresh, training steps begin with it

    # ---- views
    def _v(self, arena, key):
        e = self.by_key[key]
        return arena[e.off:e.off + e.numel]

    def p(self, key):
        return self._v(self.master, key)

    def g(self, key):
        return self._v(self.grad, key)

    def wv(self, key):
        return self._v(self.w, key)

    def wtv(self, key):
        return self._v(self.wt, key)

    def has(self, key):
        return key in self.by_key

    def num_logical_params(self):
        n = 0
        for e in self.entries:
            m = 1
            for s in e.logical:
                m *= s
            n += m * (9 if e.kind == "conv3" else 1)
        return n

    # ---- compute copies (call after every optimiser step / weight load)
    def _tile_table(self):
        """64x64-tile records for pdmk_transpose_tiles covering every dgrad copy (built once, lives on the device)."""
        import numpy as np
        recs = []
        for e in self.entries:
            if e.kind == "lin":
                jobs = [(e.off, e.off, e.shape[0], e.shape[1], e.shape[1], e.shape[0])]
            elif e.kind == "conv3":
                co, _, ci = e.shape          # src [co][t][ci] -> dst [ci][8-t][co]
                jobs = [(e.off + t * ci, e.off + (8 - t) * co, co, ci, 9 * ci, 9 * co) for t in range(9)]
            else:
                continue
            for so, do, rows, cols, sld, dld in jobs:
                r0s, c0s = np.arange(0, rows, 64), np.arange(0, cols, 64)
                rr, cc = np.meshgrid(r0s, c0s, indexing="ij")
                n = rr.size
                t = np.zeros((n, 12), dtype=np.int64)
                t[:, 0], t[:, 1] = so & 0xFFFFFFFF, so >> 32
                t[:, 2], t[:, 3] = do & 0xFFFFFFFF, do >> 32
                t[:, 4], t[:, 5], t[:, 6], t[:, 7] = rows, cols, sld, dld
                t[:, 8], t[:, 9] = rr.ravel(), cc.ravel()
                recs.append(t)
        tab = np.concatenate(recs, 0)
        tab = np.where(tab >= 2 ** 31, tab - 2 ** 32, tab).astype(np.int32)
        return torch.from_numpy(tab).to(self.master.device), tab.shape[0]

    # ---- upsampler convs as four 2x2 phase convs (engine.conv3 mode 2, pdmk.h conv_mode 5..12): the phase weights are
    # derived copies like w / wt, re-packed from the fp32 master after every optimiser step
    def up2_weights(self, key):
        """(wp [4, Co, 4 Ci], wpt [Ci, 16 Co] (phase-major inside a row) or None) of the 3x3 conv `key`, packed on first use."""
        if not hasattr(self, "_up2"):
            self._up2 = {}
        if key not in self._up2:
            e = self.by_key[key + ".weight"]
            co, _, ci = e.shape
            wp = torch.empty((4, co, 4 * ci), device=self.master.device, dtype=self.dtype)
            wpt = torch.empty((ci, 16 * co), device=self.master.device, dtype=self.dtype) if self.train else None
            self._up2[key] = (wp, wpt, self._up2_table(co, ci) if wpt is not None else None)
            k.up2_pack_weights(self.p(key + ".weight"), wp, None, co, ci)
            if wpt is not None:
                k.transpose_tiles(wp, wpt, *self._up2[key][2])
        return self._up2[key][:2]

    def _up2_table(self, co, ci):
        """64x64-tile records (pdmk_transpose_tiles) for wpt[ci][p][3 - j][co] = wp[p][co][j][ci]: sixteen [Co][Ci] -> [Ci][Co]
        transposes (the pack kernel writes wp coalesced; writing the transposed copy from it as well was uncoalesced and
        took 170 us per conv - 0.5 ms of every step)."""
        import numpy as np
        recs = []
        for p_ in range(4):
            for j in range(4):
                so, do = p_ * co * 4 * ci + j * ci, (p_ * 4 + (3 - j)) * co
                rr, cc = np.meshgrid(np.arange(0, co, 64), np.arange(0, ci, 64), indexing="ij")
                t = np.zeros((rr.size, 12), dtype=np.int64)
                t[:, 0], t[:, 2] = so, do
                t[:, 4], t[:, 5], t[:, 6], t[:, 7] = co, ci, 4 * ci, 16 * co
                t[:, 8], t[:, 9] = rr.ravel(), cc.ravel()
                recs.append(t)
        tab = np.concatenate(recs, 0).astype(np.int32)
        return torch.from_numpy(tab).to(self.master.device), tab.shape[0]

    def refresh_up2(self):
        """Forward phase weights from the fp32 master (main stream, after the optimiser: the up blocks of the next forward
        read them); the transposed copies follow with the other dgrad copies (refresh_wt)."""
        for key, (wp, wpt, tab) in getattr(self, "_up2", {}).items():
            e = self.by_key[key + ".weight"]
            k.up2_pack_weights(self.p(key + ".weight"), wp, None, e.shape[0], e.shape[2])

    def refresh(self, w_is_fresh=False, wt=True):
        """master -> w (cast) -> wt (tiled transposes), two launches.  `w_is_fresh`: the fused AdamW already wrote w.
        `wt=False` leaves the transposed (dgrad) copies to a later `refresh_wt()` - they are not read before the next
        backward pass, so the stepper launches that pass beside the next forward instead of after the optimiser."""
        if self.dtype != torch.float32 and not w_is_fresh:
            k.cast_permute(self.master, self.w, self.total, 1, 1, 0)
        self.refresh_up2()
        if wt:
            self.refresh_wt()

    def refresh_wt(self):
        if self.wt is not None:
            if not hasattr(self, "_tiles"):
                self._tiles = self._tile_table()
            k.transpose_tiles(self.w, self.wt, self._tiles[0], self._tiles[1])
            for key, (wp, wpt, tab) in getattr(self, "_up2", {}).items():
                if wpt is not None:
                    k.transpose_tiles(wp, wpt, tab[0], tab[1])

    # ---- state dict interchange (diffusers names, pruned shapes)
    @torch.no_grad()
    def load_state_dict(self, sd, strict=True, refresh=True, arena=None):
        """Packs diffusers-named tensors (pruned shapes) into `arena` (default: the master weights, followed by a refresh of
        the compute copies; any other arena of the same layout - optimiser moments - is only filled).
        Shape mismatches raise ValueError, missing / unexpected names KeyError."""
        target = self.master if arena is None else arena
        seen = set()
        for e in self.entries:
            packed = torch.zeros(e.shape)
            for name, rows, d0, s0 in e.srcs:
                if name not in sd:
                    raise KeyError(f"missing key {name}")
                t = sd[name].detach().to(torch.float32).cpu()
                seen.add(name)
                total = sum(r for n_, r, _, _ in e.srcs if n_ == name)
                if e.kind == "conv3":
                    if not (t.dim() == 4 and t.shape[0] == total and t.shape[1] == e.logical[1] and tuple(t.shape[2:]) == (3, 3)):
                        raise ValueError(f"{name}: got {tuple(t.shape)}, expected ({total},{e.logical[1]},3,3)")
                    t = t.permute(0, 2, 3, 1)[s0:s0 + rows]
                    packed[d0:d0 + rows, :, :t.shape[3]] = t.reshape(rows, 9, t.shape[3])
                elif e.kind == "lin":
                    t = t.reshape(t.shape[0], -1)
                    if tuple(t.shape) != (total, e.logical[1]):
                        raise ValueError(f"{name}: got {tuple(t.shape)}, expected {(total, e.logical[1])}")
                    packed[d0:d0 + rows, :t.shape[1]] = t[s0:s0 + rows]
                else:
                    if tuple(t.shape) != (total,):
                        raise ValueError(f"{name}: got {tuple(t.shape)}, expected {(total,)}")
                    packed[d0:d0 + rows] = t[s0:s0 + rows]
            target[e.off:e.off + e.numel].copy_(packed.reshape(-1))
        if strict:
            extra = set(sd) - seen
            if extra:
                raise KeyError(f"unexpected keys in state dict: {sorted(extra)[:5]} ...")
        if refresh and arena is None:
            self.refresh()

    def state_dict_names(self):
        return list(dict.fromkeys(name for e in self.entries for name, _r, _d, _s in e.srcs))

    @torch.no_grad()
    def state_dict(self, arena=None):
        arena = self.master if arena is None else arena
        pieces = {}
        for e in self.entries:
            t = arena[e.off:e.off + e.numel].detach().cpu().reshape(e.shape)
            for name, rows, d0, s0 in e.srcs:
                if e.kind == "conv3":
                    v = t[d0:d0 + rows, :, :e.logical[1]].reshape(rows, 3, 3, e.logical[1]).permute(0, 3, 1, 2).contiguous()
                elif e.kind == "lin":
                    v = t[d0:d0 + rows, :e.logical[1]].contiguous()
                    if name.endswith("conv_shortcut.weight"):
                        v = v.reshape(rows, e.logical[1], 1, 1)
                else:
                    v = t[d0:d0 + rows].clone()
                pieces.setdefault(name, []).append((s0, v))
        return {n: (torch.cat([v for _, v in sorted(ps, key=lambda x: x[0])], 0) if len(ps) > 1 else ps[0][1])
                for n, ps in pieces.items()}

    @torch.no_grad()
    def init_random(self, seed=0):
        """PyTorch-default initialisation of every logical tensor (the reference's random_init path,
        unet_2d_conditional.py:2406-2408), generated directly in packed layout on the device."""
        g = torch.Generator(device=self.master.device).manual_seed(seed)
        dev = self.master.device

        def scatter(view, e, make):
            """One random tensor per SOURCE tensor of the entry (its logical rows), scattered to the packed rows with one
            index_copy_ - an interleaved GEGLU projection has hundreds of 8-row pieces, not hundreds of kernels."""
            by_name = {}
            for name, rows, d0, s0 in e.srcs:
                by_name.setdefault(name, []).append((rows, d0, s0))
            for name, pieces in by_name.items():
                total = sum(r for r, _, _ in pieces)
                full = make(name, total)
                dst = torch.cat([torch.arange(d0, d0 + r) for r, d0, _ in pieces]).to(dev)
                src = torch.cat([torch.arange(s0, s0 + r) for r, _, s0 in pieces]).to(dev)
                view.index_copy_(0, dst, full.index_select(0, src))

        for e in self.entries:
            view = self.master[e.off:e.off + e.numel].view(e.shape)
            view.zero_()
            if e.kind == "vec":
                if e.key.endswith(".weight") and ("norm" in e.key):
                    scatter(view, e, lambda name, n: torch.ones(n, device=dev))
                elif e.key.endswith(".bias") and ("norm" not in e.key):
                    fan = self._fan_in_of_bias(e.key)
                    scatter(view, e, lambda name, n: (torch.rand(n, generator=g, device=dev) * 2 - 1) / fan ** 0.5)
            elif e.kind == "lin":
                kk = e.logical[1]
                scatter(view[:, :kk], e, lambda name, n: (torch.rand(n, kk, generator=g, device=dev) * 2 - 1) / kk ** 0.5)
            else:
                co, ci = e.logical
                view[:co, :, :ci] = (torch.rand(co, 9, ci, generator=g, device=dev) * 2 - 1) / (9 * ci) ** 0.5
        self.refresh()

    def _fan_in_of_bias(self, key):
        wkey = key[:-len(".bias")] + ".weight"
        if wkey in self.by_key:
            e = self.by_key[wkey]
            return e.logical[1] * (9 if e.kind == "conv3" else 1)
        for e in self.entries:          # fused projections
            if any(t_[0] == wkey for t_ in e.srcs):
                return e.logical[1]
        return 1
