"""Functional fp32 forward of the (pruned or dense) SD-2.1 U-Net on a state dict (oracle side, NCHW, plain torch).

Dataflow: pdm/models/unet/unet_2d_conditional.py:1417-1728.  Leaves (diffusers-resident):
  ResBlock        pdm/models/unet/blocks.py:308-381 (body restated there)   twin: ldm openaimodel.py:164-276
  attention       blocks.py:203-295                                          twin: ldm attention.py:152-196
  GEGLU           blocks.py:44-59 (exact erf GELU)                           twin: ldm attention.py:37-44
  transformer 2D  GN(eps 1e-6) -> (B,HW,C) -> Linear -> block -> Linear -> NCHW -> +residual (blocks.py:1168-1228)
  timestep embed  Timesteps(flip_sin_to_cos=True, shift 0)                   twin: ldm util.py:151-172
Block activations are captured where trainer.py:557-572 hooks them (down_blocks[i] output[0], mid, up_blocks[i]).
"""
import math
import torch
import torch.nn.functional as F

from .arch import block_layout
from .config import UNetConfig


# Mixed-precision policy of the norms / softmax (set by step._mixed): False = the op runs in its input dtype (what torch's CPU
# autocast does: group_norm / layer_norm / softmax of a bf16 tensor stay bf16); True = CUDA autocast's policy, which the reference
# trains under (trainer.py:516-527 + accelerate): group_norm / layer_norm are on the fp32 list (inputs upcast, fp32 result that the
# next conv / linear casts to bf16 once), and the fused F.scaled_dot_product_attention (blocks.py:257-277) keeps QK^T and the
# softmax in fp32, rounding only P to bf16 for the PV product.
FP32_NORMS = False
# "fp8_e4m3" attention precision (BASELINE.json configs[4]; no counterpart in the reference, whose attention runs in the activation
# dtype, blocks.py:257-277): Q, K and V rounded to the nearest OCP e4m3fn value before QK^T and PV, gradients straight through.
ATTN_FP8 = False


def quant_e4m3(x):
    """Nearest e4m3fn value (4 exponent bits, bias 7, 3 mantissa bits; max 448; subnormal step 2^-9), round-half-even,
    saturating; same dtype as x.  torch.frexp / ldexp / round are exact here, so this IS the rounding, not an approximation."""
    xf = x.float()
    ax = xf.abs().clamp(max=448.0)
    _, ex = torch.frexp(ax)                              # ax = m * 2^ex, m in [0.5, 1)  ->  floor(log2 ax) = ex - 1
    e = (ex - 1).clamp(min=-6) - 3
    q = torch.ldexp(torch.ones_like(ax), e)
    r = torch.round(ax / q) * q
    out = torch.where(torch.isnan(xf), xf, torch.copysign(r, xf))
    return out.to(x.dtype)


def _fq(x):
    return x + (quant_e4m3(x) - x).detach() if ATTN_FP8 else x


def _gn(x, groups, w, b, eps):
    if FP32_NORMS and x.dtype != torch.float32:
        return F.group_norm(x.float(), groups, w.float(), b.float(), eps)
    return F.group_norm(x, groups, w, b, eps)


def _ln(x, shape, w, b, eps):
    if FP32_NORMS and x.dtype != torch.float32:
        return F.layer_norm(x.float(), shape, w.float(), b.float(), eps)
    return F.layer_norm(x, shape, w, b, eps)


def timestep_embedding(t, dim):
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def resblock(sd, p, x, temb, groups1, groups2, eps=1e-5):
    h = _gn(x, groups1, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    tp = F.linear(F.silu(temb), sd[p + ".time_emb_proj.weight"], sd[p + ".time_emb_proj.bias"])
    h = h + tp[:, :, None, None]
    h = _gn(h, groups2, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def attention(sd, p, x, ctx, heads, head_dim):
    B, N, _ = x.shape
    src = x if ctx is None else ctx
    q = F.linear(x, sd[p + ".to_q.weight"]).view(B, N, heads, head_dim).transpose(1, 2)
    k = F.linear(src, sd[p + ".to_k.weight"]).view(B, -1, heads, head_dim).transpose(1, 2)
    v = F.linear(src, sd[p + ".to_v.weight"]).view(B, -1, heads, head_dim).transpose(1, 2)
    q, k, v = _fq(q), _fq(k), _fq(v)
    if FP32_NORMS and q.dtype != torch.float32:      # fused SDPA numerics: fp32 scores and softmax, P rounded once
        with torch.autocast("cpu", enabled=False):
            s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * (head_dim ** -0.5)
            pr = torch.softmax(s, dim=-1).to(v.dtype)
            o = torch.matmul(pr, v)
    else:
        s = torch.matmul(q, k.transpose(-1, -2)) * (head_dim ** -0.5)
        o = torch.matmul(torch.softmax(s, dim=-1), v)
    o = o.transpose(1, 2).reshape(B, N, heads * head_dim)
    return F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def transformer2d(sd, p, x, ctx, heads1, heads2, head_dim, groups):
    B, C, H, W = x.shape
    res = x
    h = _gn(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = F.linear(h, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    t = p + ".transformer_blocks.0"
    n = _ln(h, (C,), sd[t + ".norm1.weight"], sd[t + ".norm1.bias"], 1e-5)
    h = attention(sd, t + ".attn1", n, None, heads1, head_dim) + h
    n = _ln(h, (C,), sd[t + ".norm2.weight"], sd[t + ".norm2.bias"], 1e-5)
    h = attention(sd, t + ".attn2", n, ctx, heads2, head_dim) + h
    n = _ln(h, (C,), sd[t + ".norm3.weight"], sd[t + ".norm3.bias"], 1e-5)
    f = F.linear(n, sd[t + ".ff.net.0.proj.weight"], sd[t + ".ff.net.0.proj.bias"])
    a, g = f.chunk(2, dim=-1)
    f = a * F.gelu(g)
    h = F.linear(f, sd[t + ".ff.net.2.weight"], sd[t + ".ff.net.2.bias"]) + h
    h = F.linear(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return h + res


def unet_forward(sd, cfg: UNetConfig, info, sample, timesteps, ehs, acts=None):
    """sample [B,4,H,W], timesteps [B] int64, ehs [B,T,ctx] -> [B,4,H,W]; fills ``acts`` (d0..,m,u0..) if given."""
    G = cfg.norm_num_groups
    c0 = cfg.block_out_channels[0]
    temb = timestep_embedding(timesteps, c0)
    temb = F.linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    temb = F.linear(F.silu(temb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    h = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    skips = [h]
    blocks = block_layout(cfg)

    def run_res(r, x):
        ri = info[r["prefix"]]
        if ri["dropped"]:
            if r["skip_dim"] is not None:      # up path: keep the non-skip channels (blocks.py:502-515)
                return x[:, : x.shape[1] - r["skip_dim"]]
            return x
        return resblock(sd, r["prefix"], x, temb, G, ri["groups2"])

    def run_att(a, x):
        ai = info[a["prefix"]]
        if ai["dropped"]:
            return x
        return transformer2d(sd, a["prefix"], x, ehs, ai["heads1"], ai["heads2"], a["c"] // a["heads"], G)

    for b in blocks:
        if b["kind"] == "down":
            for j, r in enumerate(b["resnets"]):
                h = run_res(r, h)
                if b["attns"]:
                    h = run_att(b["attns"][j], h)
                skips.append(h)
            if b["sampler"]:
                p = f"{b['name']}.downsamplers.0.conv"
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], stride=2, padding=1)
                skips.append(h)
            if acts is not None:
                acts[f"d{b['idx']}"] = h
        elif b["kind"] == "mid":
            h = run_res(b["resnets"][0], h)
            h = run_att(b["attns"][0], h)
            h = run_res(b["resnets"][1], h)
            if acts is not None:
                acts["m"] = h
        else:
            for j, r in enumerate(b["resnets"]):
                h = torch.cat([h, skips.pop()], dim=1)
                h = run_res(r, h)
                if b["attns"]:
                    h = run_att(b["attns"][j], h)
            if b["sampler"]:
                p = f"{b['name']}.upsamplers.0.conv"
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
            if acts is not None:
                acts[f"u{b['idx']}"] = h
    assert not skips
    h = _gn(h, G, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], 1e-5)
    h = F.silu(h)
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def unet_macs(sd, cfg, info, hw, ctx_len):
    """Exact forward MACs per image (convs, linears, QK^T and PV) from the actual (pruned) shapes."""
    macs = 0
    res = {"conv_in": hw}
    # resolution per prefix
    ch = cfg.block_out_channels
    n = len(ch)
    cur = hw
    where = {}
    for b in block_layout(cfg):
        if b["kind"] == "down":
            for r in b["resnets"]:
                where[r["prefix"]] = cur
            for a in b["attns"]:
                where[a["prefix"]] = cur
            if b["sampler"]:
                where[f"{b['name']}.downsamplers.0.conv"] = cur // 2
                cur //= 2
        elif b["kind"] == "mid":
            for r in b["resnets"]:
                where[r["prefix"]] = cur
            where[b["attns"][0]["prefix"]] = cur
        else:
            for r in b["resnets"]:
                where[r["prefix"]] = cur
            for a in b["attns"]:
                where[a["prefix"]] = cur
            if b["sampler"]:
                cur *= 2
                where[f"{b['name']}.upsamplers.0.conv"] = cur
    where["conv_in"] = hw
    where["conv_out"] = hw
    for k, w in sd.items():
        if not k.endswith(".weight") or w.dim() < 2:
            continue
        name = k[: -len(".weight")]
        owner = None
        for pfx, L in where.items():
            if name == pfx or name.startswith(pfx + "."):
                owner = L
        if name.startswith("time_embedding") or name.endswith("time_emb_proj"):
            macs += w.numel()
            continue
        assert owner is not None, name
        px = owner * owner
        if name.endswith("attn2.to_k") or name.endswith("attn2.to_v"):
            macs += w.numel() * ctx_len
        else:
            macs += w.numel() * px
    for b in block_layout(cfg):
        for a in b["attns"]:
            ai = info[a["prefix"]]
            if ai["dropped"]:
                continue
            px = where[a["prefix"]] ** 2
            hd = a["c"] // a["heads"]
            macs += 2 * ai["heads1"] * px * px * hd
            macs += 2 * ai["heads2"] * px * ctx_len * hd
    return macs
