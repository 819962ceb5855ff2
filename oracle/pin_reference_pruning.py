#!/usr/bin/env python3
"""Pin the oracle's arch-vector handling and physical pruning with outputs PRODUCED BY THE REFERENCE'S OWN CODE.

Runs ONLY in the build container (reads /root/reference).  The reference's model modules cannot be imported here
(`diffusers` is missing), but the pieces below are plain-tensor code: each method body is taken out of the reference's
source file AT RUN TIME (ast, nothing is copied into this repository) and executed unmodified on torch tensors /
torch.nn layers.  What is committed is data only: tests/golden/reference_pruning.json.

  HyperStructure.transform_arch_vector     pdm/models/hypernet.py:100-126   (incl. the force_width_non_zero branch)
  HyperStructure.get_random_arch_vector    pdm/models/hypernet.py:128-150   (torch global RNG, seeded)
  GEGLUGated.prune_gate                    pdm/models/unet/blocks.py:62-76
  FeedForwardWidthGated.prune              pdm/models/unet/blocks.py:130-138
  GatedAttention.prune                     pdm/models/unet/blocks.py:162-196
  ResnetBlock2DWidthGated.prune            pdm/models/unet/blocks.py:434-475
  ResnetBlock2DWidthDepthGated.prune       pdm/models/unet/blocks.py:646-702   (kept and dropped)

For the pruning methods: every ResBlock / transformer of the TINY topology is rebuilt from torch.nn layers holding the
oracle's seeded dense weights, its gates get the raw (pre-threshold) sub-vectors the oracle's gate walk assigns, the
reference method runs, and the resulting tensors are recorded as (shape, sha256 of the fp32 bytes): physical pruning is
pure indexing, so a hash pins it bit-exactly.  tests/test_oracle_golden.py re-derives the same tensors with
oracle/pdm_ref/weights.prune_state_dict and compares the hashes (and the product's slice_dense_state_dict is in turn
tested bit-exact against the oracle).  NOT pinned by this (needs the diffusers module tree): which gate goes to which
module (set_structure walk) - that stays a reading of unet_2d_conditional.py:1366-1415.
"""
import ast
import hashlib
import json
import os
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

from pdm_ref import arch, weights  # noqa: E402
from pdm_ref.config import UNetConfig  # noqa: E402
from pdm.utils.estimation_utils import hard_concrete as ref_hard_concrete  # noqa: E402  (the reference's own)


def ref_method(relpath, cls, name):
    """The function object of `cls.name` compiled from the reference's source text (decorators dropped: called unbound)."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name == name:
                    f.decorator_list = []
                    ns = {"torch": torch, "nn": nn, "hard_concrete": ref_hard_concrete}
                    exec(compile(ast.Module(body=[f], type_ignores=[]), path, "exec"), ns)
                    return ns[name]
    raise KeyError((relpath, cls, name))


def digest(t):
    t = t.detach().to(torch.float32).contiguous()
    return [list(t.shape), hashlib.sha256(t.numpy().tobytes()).hexdigest()]


transform = ref_method("pdm/models/hypernet.py", "HyperStructure", "transform_arch_vector")
get_random = ref_method("pdm/models/hypernet.py", "HyperStructure", "get_random_arch_vector")
prune_gate = ref_method("pdm/models/unet/blocks.py", "GEGLUGated", "prune_gate")
ff_prune = ref_method("pdm/models/unet/blocks.py", "FeedForwardWidthGated", "prune")
attn_prune = ref_method("pdm/models/unet/blocks.py", "GatedAttention", "prune")
res_prune = ref_method("pdm/models/unet/blocks.py", "ResnetBlock2DWidthGated", "prune")
resd_prune = ref_method("pdm/models/unet/blocks.py", "ResnetBlock2DWidthDepthGated", "prune")

out = {"arch": [], "pruning": []}

# ------------------------------------------------------------------ arch-vector classmethods
for cfg_name in ("tiny", "sd21"):
    cfg = getattr(UNetConfig, cfg_name)()
    structure = arch.structure(cfg)
    n = sum(w for sub in structure["width"] for w in sub) + sum(d for sub in structure["depth"] for d in sub)
    for seed, ratio in ((0, 0.55), (1, 0.82), (2, 0.18), (3, 1.0)):
        torch.manual_seed(seed)
        av = get_random(None, ratio, structure)
        assert av.shape == (1, n)
        rec = {"cfg": cfg_name, "seed": seed, "ratio": ratio, "n": n, "random_arch_vector": digest(av),
               "kept": int((av >= 0.5).sum())}
        # transform on a dense random vector (values on both sides of 0.5), plain and with force_width_non_zero on a vector
        # whose first two width sub-vectors are all below the threshold
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.rand(1, n, generator=g)
        w0, w1 = structure["width"][0][0], structure["width"][0][0] + (structure["width"][0][1] if len(structure["width"][0]) > 1 else structure["width"][1][0])
        x[0, :w1] = 0.3 * x[0, :w1]
        for force in (False, True):
            tv = transform(None, x.clone(), structure, force_width_non_zero=force)
            rec[f"transform_force{int(force)}"] = {"width": [digest(w) for w in tv["width"]],
                                                   "depth": [digest(d) for d in tv["depth"]],
                                                   "nwidth": len(tv["width"]), "ndepth": len(tv["depth"]),
                                                   "first_elems": [float(w[0, 0]) for w in tv["width"][:3]]}
        rec["transform_input_seed"] = 100 + seed
        out["arch"].append(rec)


# ------------------------------------------------------------------ pruning methods on the tiny topology
class Obj:            # attribute bag standing for `self` (the methods only read / replace attributes)
    def __init__(self, **kw):
        self.__dict__.update(kw)


def lin(sd, name, bias=True):
    w = sd[name + ".weight"]
    w = w.reshape(w.shape[0], -1)
    l_ = nn.Linear(w.shape[1], w.shape[0], bias=bias and (name + ".bias") in sd)
    l_.weight.data = w.clone()
    if l_.bias is not None:
        l_.bias.data = sd[name + ".bias"].clone()
    return l_


def conv(sd, name):
    w = sd[name + ".weight"]
    c = nn.Conv2d(w.shape[1], w.shape[0], kernel_size=3, stride=1, padding=1)
    c.weight.data, c.bias.data = w.clone(), sd[name + ".bias"].clone()
    return c


def gnorm(sd, name, groups):
    w = sd[name + ".weight"]
    n_ = nn.GroupNorm(groups, w.shape[0], eps=1e-5, affine=True)
    n_.weight.data, n_.bias.data = w.clone(), sd[name + ".bias"].clone()
    return n_


cfg = UNetConfig.tiny()
dense = weights.init_dense_state_dict(cfg, seed=0)
G = cfg.norm_num_groups
for case, (seed, ratio, drop) in enumerate(((0, 0.55, (1, 5, 9, 12)), (5, 0.3, ()), (7, 0.9, (0, 13)))):
    av = arch.random_arch_vector(cfg, ratio, seed=seed, drop_depth=drop)
    # raw sub-vectors in the order the oracle's walk hands them out (assign_gates applies hard_concrete; redo it raw here)
    tv = arch.transform_arch_vector(av, cfg)
    wq, dq = list(tv["width"]), list(tv["depth"])
    pruned = {}
    for b in arch.block_layout(cfg):
        pending = [(r, "res", [wq.pop(0)]) for r in b["resnets"]] + [(a, "att", [wq.pop(0), wq.pop(0), wq.pop(0)]) for a in b["attns"]]
        for ent, kind, ws in pending:
            depth = dq.pop(0).reshape(1, 1) if ent["depth"] else None
            p = ent["prefix"]
            if kind == "res":
                me = Obj(gate=Obj(gate_f=ws[0], width=G), depth_gate=Obj(gate_f=depth), norm1=gnorm(dense, p + ".norm1", G),
                         conv1=conv(dense, p + ".conv1"), time_emb_proj=lin(dense, p + ".time_emb_proj"),
                         norm2=gnorm(dense, p + ".norm2", G), conv2=conv(dense, p + ".conv2"),
                         conv_shortcut=(object() if (p + ".conv_shortcut.weight") in dense else None),
                         nonlinearity=nn.SiLU(), dropout=nn.Dropout(0.0), pruned=False, dropped=False)
                (resd_prune if depth is not None else res_prune)(me)
                if me.dropped:
                    pruned[p] = "dropped"
                    continue
                assert me.pruned
                for nm in ("conv1", "time_emb_proj", "norm2", "conv2"):
                    m = getattr(me, nm)
                    pruned[f"{p}.{nm}.weight"] = digest(m.weight.data)
                    pruned[f"{p}.{nm}.bias"] = digest(m.bias.data)
                pruned[f"{p}.norm2.num_groups"] = int(me.norm2.num_groups)
            else:
                if depth is not None and float(ref_hard_concrete(depth)[0]) == 0:       # blocks.py:1323-1334: identity
                    pruned[p] = "dropped"
                    continue
                t = p + ".transformer_blocks.0"
                H = ent["heads"]
                for an, gf in (("attn1", ws[0]), ("attn2", ws[1])):
                    me = Obj(gate=Obj(gate_f=gf), heads=H, to_q=lin(dense, f"{t}.{an}.to_q"), to_k=lin(dense, f"{t}.{an}.to_k"),
                             to_v=lin(dense, f"{t}.{an}.to_v"), to_out=[lin(dense, f"{t}.{an}.to_out.0")], pruned=False)
                    attn_prune(me)
                    for nm, m in (("to_q", me.to_q), ("to_k", me.to_k), ("to_v", me.to_v), ("to_out.0", me.to_out[0])):
                        pruned[f"{t}.{an}.{nm}.weight"] = digest(m.weight.data)
                    pruned[f"{t}.{an}.to_out.0.bias"] = digest(me.to_out[0].bias.data)
                    pruned[f"{t}.{an}.heads"] = int(me.heads)
                geglu = Obj(gate=Obj(gate_f=ws[2]), dim_out=4 * ent["c"], proj=lin(dense, f"{t}.ff.net.0.proj"), pruned=False)
                geglu.prune_gate = types.MethodType(prune_gate, geglu)
                ff = Obj(net=[geglu, nn.Dropout(0.0), lin(dense, f"{t}.ff.net.2")])
                ff_prune(ff)
                pruned[f"{t}.ff.net.0.proj.weight"] = digest(geglu.proj.weight.data)
                pruned[f"{t}.ff.net.0.proj.bias"] = digest(geglu.proj.bias.data)
                pruned[f"{t}.ff.net.2.weight"] = digest(ff.net[2].weight.data)
                pruned[f"{t}.ff.net.2.bias"] = digest(ff.net[2].bias.data)
    assert not wq and not dq
    out["pruning"].append({"cfg": "tiny", "dense_seed": 0, "arch_seed": seed, "ratio": ratio, "drop_depth": list(drop),
                           "arch_vector": digest(av), "tensors": pruned})

dst = os.path.join(ROOT, "tests", "golden", "reference_pruning.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print(f"wrote {dst}: {len(out['arch'])} arch records, {sum(len(c['tensors']) for c in out['pruning'])} pruned-tensor records")
