"""`UNet2DConditionModelPruned` — host-side mirror of pdm/models/unet/unet_2d_conditional.py:2183-2495 on the MI355X engine.

Same construction contract as the reference (trainer.py:2165-2176):
    UNet2DConditionModelPruned.from_pretrained(name_or_path, subfolder="unet", down_block_types=..., mid_block_type=...,
        up_block_types=..., gated_ff=..., ff_gate_width=..., arch_vector=Tensor[1, n], random_init=bool)
    model(sample[B,4,H,W], timestep[B], encoder_hidden_states[B,T,ctx]).sample -> [B,4,H,W]
    model.down_blocks[i] / .mid_block / .up_blocks[i].register_forward_hook(fn)  (trainer.py:557-572)
    .parameters(), .state_dict(), .load_state_dict(), .save_pretrained(dir)      (trainer.py:314-346)
but the arch vector is resolved to a static packed shape table up front and all arithmetic runs in libpdmk.
There is no CPU path: constructing the model without a GPU + libpdmk.so raises.
"""
import json
import os
from types import SimpleNamespace

import torch

from ... import _pdmk as k
from .engine import UNetEngine, Act
from .params import ParamStore, build_entries
from .spec import UNetConfig, apply_arch_vector, gate_structure, padc

# Block-type STRINGS the YAML recipes carry (`model.prediction_model.unet_down_blocks / unet_up_blocks`) are the factory names
# of get_down_block / get_up_block (pdm/models/unet/unet_2d_conditional.py:90-243, 382-502): "...HalfGated" selects the
# *WidthHalfDepthGated container class (:119-132, :217-242, :397-410, :477-502), the plain names the diffusers blocks (the
# teacher); a leading "UNetRes" is stripped (:90, :382).  The container class names themselves are accepted as aliases.
# value: does the stage carry transformers
_GATED_DOWN = {"CrossAttnDownBlock2DHalfGated": True, "DownBlock2DHalfGated": False,
               "CrossAttnDownBlock2DWidthHalfDepthGated": True, "DownBlock2DWidthHalfDepthGated": False,
               "CrossAttnDownBlock2D": True, "DownBlock2D": False}
_GATED_UP = {"CrossAttnUpBlock2DHalfGated": True, "UpBlock2DHalfGated": False,
             "CrossAttnUpBlock2DWidthHalfDepthGated": True, "UpBlock2DWidthHalfDepthGated": False,
             "CrossAttnUpBlock2D": True, "UpBlock2D": False}


def _check_block_types(names, allowed, attn_stages, what):
    """The engine's topology is the one every shipped recipe selects; a recipe that names another one is an error, not a
    silently different model."""
    if not names:
        return
    names = [n[7:] if n.startswith("UNetRes") else n for n in names]
    for nme in names:
        if nme not in allowed:
            raise ValueError(f"block type {nme!r} is not supported by the MI355X engine (supported: {tuple(allowed)}; the "
                             f"fully depth-gated '...2DGated' containers are not used by any shipped config)")
    got = tuple(allowed[n] for n in names)
    if got != tuple(attn_stages):
        raise ValueError(f"{what}={names}: transformer stages {got} differ from the engine's topology {tuple(attn_stages)}")


class _BlockHandle:
    """Stand-in for an nn.Module block: only carries forward hooks (trainer.py:557-572 registers them)."""

    def __init__(self, name):
        self.name = name
        self._hooks = []

    def register_forward_hook(self, fn):
        self._hooks.append(fn)
        return SimpleNamespace(remove=lambda: self._hooks.remove(fn))


class UNet2DConditionModelPruned:
    config_name = "config.json"

    def __init__(self, cfg: UNetConfig = None, arch_vector=None, device=None, dtype=torch.bfloat16, train=True,
                 seed=0, init=True, attention_precision=None):
        if not torch.cuda.is_available():
            raise RuntimeError("UNet2DConditionModelPruned (MI355X engine) needs a GPU; there is no CPU fallback")
        self.cfg = cfg or UNetConfig.sd21()
        self.device = torch.device(device or "cuda:0")
        self.dtype = dtype
        self.arch_vector = None if arch_vector is None else arch_vector.detach().float().cpu().clone()
        self.blocks = apply_arch_vector(self.cfg, self.arch_vector)
        self.store = ParamStore(build_entries(self.cfg, self.blocks), self.device, dtype, train=train)
        self.engine = UNetEngine(self.cfg, self.blocks, self.store, dtype)
        self.set_attention_precision(attention_precision)
        self.training = train
        n = len(self.cfg.block_out_channels)
        self.down_blocks = [_BlockHandle(f"down_blocks.{i}") for i in range(n)]
        self.mid_block = _BlockHandle("mid_block")
        self.up_blocks = [_BlockHandle(f"up_blocks.{i}") for i in range(n)]
        self.config = SimpleNamespace(in_channels=self.cfg.in_channels, out_channels=self.cfg.out_channels,
                                      sample_size=96, cross_attention_dim=self.cfg.cross_attention_dim,
                                      block_out_channels=self.cfg.block_out_channels)
        if init:
            self.store.init_random(seed)

    def set_attention_precision(self, precision):
        """None / "bf16" / "fp32": attention operands in the activation dtype (the reference: F.scaled_dot_product_attention on what
        the projections produce, blocks.py:257-277).  "fp8_e4m3" (BASELINE.json configs[4], PDMK_ATTN_FP8=1): Q, K and V are rounded to
        the e4m3fn value grid before the attention kernels, forward and backward alike (pdmk_quantize_e4m3; straight-through for the
        gradients of the projections)."""
        if precision is None and os.environ.get("PDMK_ATTN_FP8") == "1":
            precision = "fp8_e4m3"
        if precision not in (None, "bf16", "fp32", "no", "fp8_e4m3"):
            raise ValueError(f"attention_precision={precision!r}: expected None, 'fp8_e4m3' (or 'bf16' / 'fp32' = the activation dtype)")
        self.attention_precision = precision
        self.engine.attn_fp8 = precision == "fp8_e4m3"

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, revision=None, arch_vector=None,
                        random_init=False, down_block_types=None, mid_block_type=None, up_block_types=None,
                        gated_ff=True, ff_gate_width=32, unet_config=None, torch_dtype=torch.bfloat16, device=None,
                        train=True, seed=0, attention_precision=None, **unused):
        cfg = unet_config or UNetConfig.sd21()
        _check_block_types(down_block_types, _GATED_DOWN, cfg.attn_stages_down, "down_block_types")
        _check_block_types(up_block_types, _GATED_UP, cfg.attn_stages_up, "up_block_types")
        if mid_block_type not in (None, "UNetMidBlock2DCrossAttnWidthGated", "UNetMidBlock2DCrossAttn"):
            raise ValueError(f"mid_block_type {mid_block_type!r} is not supported by the MI355X engine")
        if ff_gate_width != cfg.ff_gate_width:
            cfg = UNetConfig(**{**cfg.__dict__, "ff_gate_width": ff_gate_width})
        if not gated_ff:
            raise ValueError("gated_ff=False is not supported (every shipped config sets gated_ff: true)")
        path = pretrained_model_name_or_path
        if path and subfolder:
            path = os.path.join(path, subfolder)
        have_local = bool(path) and os.path.isdir(path)
        if arch_vector is None and have_local and os.path.exists(os.path.join(path, "arch_vector.pt")):
            arch_vector = torch.load(os.path.join(path, "arch_vector.pt"), map_location="cpu")
        model = cls(cfg, arch_vector, device, torch_dtype, train=train, seed=seed, init=random_init or not have_local,
                    attention_precision=attention_precision)
        if have_local and not random_init:
            model.load_pretrained_dir(path)
        elif not random_init:
            raise FileNotFoundError(f"{pretrained_model_name_or_path!r} is not a local directory and hub downloads are "
                                    f"not available here; pass random_init=True or a local checkpoint directory")
        return model

    def load_pretrained_dir(self, path):
        f = os.path.join(path, "diffusion_pytorch_model.safetensors")
        if os.path.exists(f):
            from safetensors.torch import load_file
            sd = load_file(f)
        else:
            sd = torch.load(os.path.join(path, "diffusion_pytorch_model.bin"), map_location="cpu")
        self.load_dense_or_pruned(sd)

    def load_dense_or_pruned(self, sd):
        """Accepts either an already-pruned state dict or a dense SD-2.1 one (sliced here by the arch vector, the
        equivalent of from_pretrained's load-then-prune, unet_2d_conditional.py:2408-2459)."""
        try:
            self.store.load_state_dict(sd)
        except (ValueError, KeyError):       # dense shapes (ValueError) / keys of dropped blocks (KeyError): prune, then load
            self.store.load_state_dict(slice_dense_state_dict(sd, self.cfg, self.blocks))

    # ------------------------------------------------------------------ nn.Module-ish surface
    def parameters(self):
        return [self.store.master]

    def named_parameters(self):
        return list(self.store.state_dict().items())

    def state_dict(self):
        return self.store.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.store.load_state_dict(sd, strict=strict)

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def requires_grad_(self, flag=True):
        self.training = self.training and flag
        return self

    def to(self, *a, **kw):
        return self

    def get_structure(self):
        return gate_structure(self.cfg)

    def num_parameters(self):
        return self.store.num_logical_params()

    def save_pretrained(self, save_directory, safe_serialization=True):
        os.makedirs(save_directory, exist_ok=True)
        sd = {n: t.contiguous() for n, t in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "diffusion_pytorch_model.safetensors"))
        else:
            torch.save(sd, os.path.join(save_directory, "diffusion_pytorch_model.bin"))
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump({"_class_name": "UNet2DConditionModelPruned", **{a: getattr(self.cfg, a) for a in
                      ("block_out_channels", "cross_attention_dim", "layers_per_block", "norm_num_groups",
                       "in_channels", "out_channels", "ff_gate_width")}}, f, indent=2)
        if self.arch_vector is not None:
            torch.save(self.arch_vector, os.path.join(save_directory, "arch_vector.pt"))

    # ------------------------------------------------------------------ forward
    def forward_nhwc(self, x, timesteps, ehs2d, B, H, W, train=None):
        train = self.training and self.store.train if train is None else train
        return self.engine.forward(x, timesteps, ehs2d, B, H, W, train)

    def __call__(self, sample, timestep, encoder_hidden_states, return_dict=True, **unused):
        """Reference-shaped call: NCHW fp32 in, `.sample` NCHW fp32 out; block hooks fire with NCHW views."""
        B, C, H, W = sample.shape
        dev = self.device
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.int64)
        timestep = timestep.to(dev).to(torch.int64).reshape(-1)
        if timestep.numel() == 1 and B > 1:
            timestep = timestep.expand(B).contiguous()
        cp = padc(C)
        x = torch.empty((B * H * W, cp), device=dev, dtype=self.dtype)
        k.nchw_to_nhwc(sample.to(dev, torch.float32).contiguous(), x, B, C, H * W, cp)
        ehs = encoder_hidden_states.to(dev).to(self.dtype).reshape(B * encoder_hidden_states.shape[1], -1).contiguous()
        pred, acts = self.forward_nhwc(x, timestep, ehs, B, H, W)
        self.last_pred, self.last_acts = pred, acts
        self._fire_hooks(acts, B)
        out = torch.empty((B, self.cfg.out_channels, H, W), device=dev, dtype=torch.float32)
        k.nhwc_to_nchw(pred.t, out, B, self.cfg.out_channels, H * W, pred.t.stride(0))
        return SimpleNamespace(sample=out) if return_dict else (out,)

    def _fire_hooks(self, acts, B):
        def nchw(act):
            M, C = act.t.shape
            side = int(round((M // B) ** 0.5))
            return act.t.reshape(B, side, side, C).permute(0, 3, 1, 2)      # (a skip is a column view of its concat buffer)
        for i, h in enumerate(self.down_blocks):
            for fn in h._hooks:
                fn(h, None, (nchw(acts[f"d{i}"]), ()))
        for fn in self.mid_block._hooks:
            fn(self.mid_block, None, nchw(acts["m"]))
        for i, h in enumerate(self.up_blocks):
            for fn in h._hooks:
                fn(h, None, nchw(acts[f"u{i}"]))


def slice_dense_state_dict(sd, cfg, blocks):
    """Physical pruning of a dense SD-2.1 state dict by the keep masks: the prune() methods of
    pdm/models/unet/blocks.py:62-76, 130-138, 162-196, 434-475, 646-702, 1323-1334 applied to tensors."""
    out = dict(sd)
    G = cfg.norm_num_groups

    def drop(prefix):
        for key in [q for q in out if q.startswith(prefix + ".")]:
            del out[key]

    for b in blocks:
        for r in b.resnets:
            p = r.name
            if r.dropped:
                drop(p)
                continue
            if r.keep_mask is None:
                continue
            m = r.keep_mask.repeat_interleave(r.cout // G)
            for nm in ("conv1.weight", "conv1.bias", "time_emb_proj.weight", "time_emb_proj.bias", "norm2.weight",
                       "norm2.bias"):
                out[f"{p}.{nm}"] = sd[f"{p}.{nm}"][m]
            out[f"{p}.conv2.weight"] = sd[f"{p}.conv2.weight"][:, m]
        for a in b.attns:
            p = a.name
            if a.dropped:
                drop(p)
                continue
            if a.keep_h1 is None:
                continue
            t = p + ".transformer_blocks.0"
            for an, hm in (("attn1", a.keep_h1), ("attn2", a.keep_h2)):
                rows = hm.repeat_interleave(64)
                for nm in ("to_q", "to_k", "to_v"):
                    out[f"{t}.{an}.{nm}.weight"] = sd[f"{t}.{an}.{nm}.weight"][rows]
                out[f"{t}.{an}.to_out.0.weight"] = sd[f"{t}.{an}.to_out.0.weight"][:, rows]
            fm = a.keep_ff.repeat_interleave(4 * a.c // cfg.ff_gate_width)
            fm2 = torch.cat([fm, fm])
            out[f"{t}.ff.net.0.proj.weight"] = sd[f"{t}.ff.net.0.proj.weight"][fm2]
            out[f"{t}.ff.net.0.proj.bias"] = sd[f"{t}.ff.net.0.proj.bias"][fm2]
            out[f"{t}.ff.net.2.weight"] = sd[f"{t}.ff.net.2.weight"][:, fm]
    return out
