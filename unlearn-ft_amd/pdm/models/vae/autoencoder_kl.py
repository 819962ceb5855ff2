"""`AutoencoderKL` on libpdmk: encode = the step right in front of the U-Net boundary; decode = the last
stage of the image-logging / FID sampler (row N3, pdm/pipelines/pruning_pipelines.py:993-995).

What the reference does with diffusers' AutoencoderKL (pdm/training/trainer.py:2405-2406, frozen, cast to the weight
dtype at :516-527, 2730-2733):
    latents = vae.encode(batch["pixel_values"].to(weight_dtype)).latent_dist.sample() * vae.config.scaling_factor
Same call surface here (`from_pretrained(path, subfolder="vae")`, `.encode(x).latent_dist.sample()`,
`.config.scaling_factor`, diffusers state-dict key names), arithmetic in libpdmk: NHWC implicit-GEMM convs, fused
GroupNorm+SiLU, the one single-head mid-block attention as two GEMMs around a row softmax, and a fused
sample-and-scale.  Inference only (the VAE is frozen); no CPU path.
Leaf semantics follow the CompVis encoder the diffusers class was converted from, which is what the oracle is pinned
against (oracle/pdm_ref/vae.py): ldm/modules/diffusionmodules/model.py:60-81, 82-143, 150-204, 368-460 and
ldm/modules/distributions/distributions.py:24-37.
"""
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Tuple

import torch

from ... import _pdmk as k
from ..unet.engine import UNetEngine, Act, _ld
from ..unet.params import ParamStore, _conv, _lin, _vec
from ..unet.spec import padc


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215
    eps: float = 1e-6

    @staticmethod
    def sd21():
        return VAEConfig()


def build_entries(cfg: VAEConfig, decoder=True):
    E = []

    def conv(key, co, ci):
        E.extend([_conv(key, co, ci), _vec(key + ".bias", [(key + ".bias", co)])])

    def norm(key, c):
        E.extend([_vec(key + ".weight", [(key + ".weight", c)]), _vec(key + ".bias", [(key + ".bias", c)])])

    def lin(key, srcs, kin):
        E.extend([_lin(key, [(n + ".weight", r) for n, r in srcs], kin), _vec(key + ".bias", [(n + ".bias", r) for n, r in srcs])])

    def res(p, ci, co):
        norm(p + ".norm1", ci)
        conv(p + ".conv1", co, ci)
        norm(p + ".norm2", co)
        conv(p + ".conv2", co, co)
        if ci != co:
            lin(p + ".conv_shortcut", [(p + ".conv_shortcut", co)], ci)

    ch = cfg.block_out_channels
    conv("encoder.conv_in", ch[0], cfg.in_channels)
    cin = ch[0]
    for i, co in enumerate(ch):
        for j in range(cfg.layers_per_block):
            res(f"encoder.down_blocks.{i}.resnets.{j}", cin, co)
            cin = co
        if i != len(ch) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co)
    res("encoder.mid_block.resnets.0", cin, cin)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", cin)
    lin(a + ".to_qk", [(a + ".to_q", cin), (a + ".to_k", cin)], cin)
    lin(a + ".to_v", [(a + ".to_v", cin)], cin)
    lin(a + ".to_out.0", [(a + ".to_out.0", cin)], cin)
    res("encoder.mid_block.resnets.1", cin, cin)
    norm("encoder.conv_norm_out", cin)
    conv("encoder.conv_out", 2 * cfg.latent_channels, cin)
    lin("quant_conv", [("quant_conv", 2 * cfg.latent_channels)], 2 * cfg.latent_channels)
    if decoder:        # diffusers Decoder: mid block first, then up_blocks over reversed(block_out_channels), 3 ResBlocks each
        lin("post_quant_conv", [("post_quant_conv", cfg.latent_channels)], cfg.latent_channels)
        rev = tuple(reversed(ch))
        conv("decoder.conv_in", rev[0], cfg.latent_channels)
        res("decoder.mid_block.resnets.0", rev[0], rev[0])
        a = "decoder.mid_block.attentions.0"
        norm(a + ".group_norm", rev[0])
        lin(a + ".to_qk", [(a + ".to_q", rev[0]), (a + ".to_k", rev[0])], rev[0])
        lin(a + ".to_v", [(a + ".to_v", rev[0])], rev[0])
        lin(a + ".to_out.0", [(a + ".to_out.0", rev[0])], rev[0])
        res("decoder.mid_block.resnets.1", rev[0], rev[0])
        cin = rev[0]
        for i, co in enumerate(rev):
            for j in range(cfg.layers_per_block + 1):
                res(f"decoder.up_blocks.{i}.resnets.{j}", cin, co)
                cin = co
            if i != len(rev) - 1:
                conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co)
        norm("decoder.conv_norm_out", cin)
        conv("decoder.conv_out", cfg.in_channels, cin)
    off = 0
    for e in E:
        e.off = off
        off += (e.numel + 127) // 128 * 128
    return E


class _Ops(UNetEngine):
    """The U-Net executor's op layer (conv3 / linear / groupnorm launch wrappers) over the VAE's parameter arena."""

    def __init__(self, store, dtype):
        self.cfg, self.blocks, self.P, self.dtype = None, None, store, dtype
        self.dev = store.master.device
        self.ws = k.groupnorm_ws(self.dev, 64, 32)
        self.tape, self.train, self.macs, self.count_macs = [], False, 0, False
        self.grad_ready_cb, self.wgrad_async, self._keep, self.fuse_geglu, self.defer_fanin = None, False, [], True, False
        self.gn_epi, self.gn_count, self.gn_miss = False, [0, 0], None     # (GroupNorm statistics from GEMM epilogues: U-Net only)


class _LatentDist:
    """DiagonalGaussianDistribution over NHWC moments held on the device."""

    def __init__(self, vae, moments, B, H, W):
        self._vae, self.moments, self._shape = vae, moments, (B, H, W)

    def sample(self, generator=None, scale=1.0, noise=None):
        B, H, W = self._shape
        C = self._vae.cfg.latent_channels
        dev = self.moments.device
        if noise is None:
            noise = torch.randn((B, C, H, W), device=dev, dtype=torch.float32, generator=generator)
        out = torch.empty((B, C, H, W), device=dev, dtype=torch.float32)
        k.latent_sample(self.moments, noise.contiguous(), out, B, C, H * W, _ld(self.moments), scale)
        return out

    def mode(self):
        B, H, W = self._shape
        C = self._vae.cfg.latent_channels
        out = torch.empty((B, C, H, W), device=self.moments.device, dtype=torch.float32)
        k.nhwc_to_nchw(self.moments, out, B, C, H * W, _ld(self.moments))
        return out

    def parameters_nchw(self):
        """mean | logvar as an NCHW fp32 tensor [B, 2C, H, W] (what diffusers keeps as `.parameters`)."""
        B, H, W = self._shape
        C2 = 2 * self._vae.cfg.latent_channels
        out = torch.empty((B, C2, H, W), device=self.moments.device, dtype=torch.float32)
        k.nhwc_to_nchw(self.moments, out, B, C2, H * W, _ld(self.moments))
        return out


class AutoencoderKL:
    def __init__(self, cfg: VAEConfig = None, device=None, dtype=torch.bfloat16, seed=0, init=True, decoder=True):
        if not torch.cuda.is_available():
            raise RuntimeError("AutoencoderKL (MI355X engine) needs a GPU; there is no CPU fallback")
        self.cfg = cfg or VAEConfig.sd21()
        self.device = torch.device(device or "cuda:0")
        self.dtype = dtype
        self.has_decoder = decoder
        self.store = ParamStore(build_entries(self.cfg, decoder), self.device, dtype, train=False)
        self.ops = _Ops(self.store, dtype)
        self.config = SimpleNamespace(scaling_factor=self.cfg.scaling_factor, latent_channels=self.cfg.latent_channels,
                                      block_out_channels=self.cfg.block_out_channels, in_channels=self.cfg.in_channels)
        if init:
            self.store.init_random(seed)

    # ------------------------------------------------------------------ construction / interchange
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, revision=None, random_init=False,
                        vae_config=None, torch_dtype=torch.bfloat16, device=None, seed=0, **unused):
        path = pretrained_model_name_or_path
        if path and subfolder:
            path = os.path.join(path, subfolder)
        have_local = bool(path) and os.path.isdir(path)
        model = cls(vae_config, device, torch_dtype, seed=seed, init=random_init or not have_local)
        if have_local and not random_init:
            f = os.path.join(path, "diffusion_pytorch_model.safetensors")
            if os.path.exists(f):
                from safetensors.torch import load_file
                sd = load_file(f)
            else:
                sd = torch.load(os.path.join(path, "diffusion_pytorch_model.bin"), map_location="cpu")
            model.load_state_dict(sd, strict=False)
        elif not random_init:
            raise FileNotFoundError(f"{pretrained_model_name_or_path!r} is not a local directory and hub downloads are "
                                    f"not available here; pass random_init=True or a local checkpoint directory")
        return model

    def load_state_dict(self, sd, strict=True):
        """diffusers AutoencoderKL keys (decoder / post_quant_conv keys are ignored when built with decoder=False); the pre-0.14
        attention names (query/key/value/proj_attn, still written by the reference's converter,
        baselines/erasing/oldcode_erasing_compvis/train-scripts/convertModels.py:120-140) are accepted."""
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        own = {}
        for key, v in sd.items():
            if not (key.startswith("encoder.") or key.startswith("quant_conv.") or
                    (self.has_decoder and (key.startswith("decoder.") or key.startswith("post_quant_conv.")))):
                continue
            parts = key.split(".")
            if "attentions" in parts and parts[-2] in ren:
                key = ".".join(parts[:-2] + [ren[parts[-2]], parts[-1]])
            own[key] = v
        self.store.load_state_dict(own, strict=strict)

    def state_dict(self):
        sd = self.store.state_dict()
        for key in ("quant_conv.weight", "post_quant_conv.weight"):
            if key in sd:
                sd[key] = sd[key].reshape(*sd[key].shape, 1, 1)
        return sd

    def requires_grad_(self, flag=False):
        return self

    def to(self, *a, **kw):
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------ forward
    def _res(self, p, x, B, H, W):
        o, G, eps = self.ops, self.cfg.norm_num_groups, self.cfg.eps
        ci = x.t.shape[1]
        h = o.groupnorm(x, p + ".norm1", B, H * W, G, ci // G, eps, True)
        h, _, _ = o.conv3(h, p + ".conv1", B, H, W, 0, p + ".conv1.bias")
        co = h.t.shape[1]
        h = o.groupnorm(h, p + ".norm2", B, H * W, G, co // G, eps, True)
        if self.store.has(p + ".conv_shortcut.weight"):
            x = o.linear(x, p + ".conv_shortcut", bias=p + ".conv_shortcut.bias")
        h, _, _ = o.conv3(h, p + ".conv2", B, H, W, 0, p + ".conv2.bias", residual=x)
        return h

    def _attn(self, p, x, B, HW):
        o, P, G = self.ops, self.store, self.cfg.norm_num_groups
        C = x.t.shape[1]
        hn = o.groupnorm(x, p + ".group_norm", B, HW, G, C // G, self.cfg.eps, False)
        qk = o.linear(hn, p + ".to_qk", bias=p + ".to_qk.bias").t               # [B*HW, 2C] = q | k
        att = torch.empty((B * HW, C), device=self.device, dtype=self.dtype)
        s = torch.empty((HW, HW), device=self.device, dtype=torch.float32)      # one image's scores at a time
        pr = torch.empty((HW, HW), device=self.device, dtype=self.dtype)
        vt = torch.empty((C, HW), device=self.device, dtype=self.dtype)
        wv, bv = P.wv(p + ".to_v.weight"), P.p(p + ".to_v.bias")
        for b in range(B):
            rows = slice(b * HW, (b + 1) * HW)
            q, kk, hb = qk[rows, :C], qk[rows, C:], hn.t[rows]
            k.gemm(q, kk, s, HW, HW, C, _ld(q), _ld(kk), HW, out_f32=True, alpha=C ** -0.5)
            k.softmax_rows(s, pr, HW, HW, HW, HW)
            # V^T[c, n] = sum_k Wv[c, k] hn[n, k]: the value projection written directly in the [N][K] layout the P.V GEMM
            # reads as its B operand; its bias is added after the contraction (softmax rows sum to one)
            k.gemm(wv, hb, vt, C, HW, C, C, _ld(hb), HW)
            k.gemm(pr, vt, att[rows], HW, C, HW, HW, HW, C, bias=bv)
        return o.linear(Act(att), p + ".to_out.0", bias=p + ".to_out.0.bias", residual=x)

    def encode_moments(self, pixel_values):
        """pixels [B, 3, R, R] (any float dtype, NCHW) -> (moments NHWC [B*h*w, padded 2*latent], B, h, w)."""
        B, C, H, W = pixel_values.shape
        assert C == self.cfg.in_channels
        n = len(self.cfg.block_out_channels)
        assert H % (1 << (n - 1)) == 0 and W % (1 << (n - 1)) == 0, "image sides must be divisible by 2^(levels-1)"
        o = self.ops
        cp = padc(C)
        x = torch.empty((B * H * W, cp), device=self.device, dtype=self.dtype)
        k.nchw_to_nhwc(pixel_values.to(self.device, torch.float32).contiguous(), x, B, C, H * W, cp)
        h, _, _ = o.conv3(Act(x, rg=False), "encoder.conv_in", B, H, W, 0, "encoder.conv_in.bias")
        for i in range(n):
            for j in range(self.cfg.layers_per_block):
                h = self._res(f"encoder.down_blocks.{i}.resnets.{j}", h, B, H, W)
            if i != n - 1:
                p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                h, H, W = o.conv3(h, p, B, H, W, 4, p + ".bias")
        h = self._res("encoder.mid_block.resnets.0", h, B, H, W)
        h = self._attn("encoder.mid_block.attentions.0", h, B, H * W)
        h = self._res("encoder.mid_block.resnets.1", h, B, H, W)
        G = self.cfg.norm_num_groups
        h = o.groupnorm(h, "encoder.conv_norm_out", B, H * W, G, h.t.shape[1] // G, self.cfg.eps, True)
        h, _, _ = o.conv3(h, "encoder.conv_out", B, H, W, 0, "encoder.conv_out.bias")
        mom = o.linear(h, "quant_conv", bias="quant_conv.bias")
        return mom.t, B, H, W

    def decode(self, z, return_dict=True):
        """latents [B, 4, h, w] NCHW (already divided by scaling_factor, as pruning_pipelines.py:994 passes them) ->
        `.sample` image [B, 3, 8h, 8w] NCHW fp32 = decoder(post_quant_conv(z))."""
        if not self.has_decoder:
            raise RuntimeError("this AutoencoderKL was built with decoder=False")
        B, C, H, W = z.shape
        assert C == self.cfg.latent_channels
        o, G = self.ops, self.cfg.norm_num_groups
        cp = padc(C)
        x = torch.empty((B * H * W, cp), device=self.device, dtype=self.dtype)
        k.nchw_to_nhwc(z.to(self.device, torch.float32).contiguous(), x, B, C, H * W, cp)
        h = o.linear(Act(x, rg=False), "post_quant_conv", bias="post_quant_conv.bias")
        h, _, _ = o.conv3(h, "decoder.conv_in", B, H, W, 0, "decoder.conv_in.bias")
        h = self._res("decoder.mid_block.resnets.0", h, B, H, W)
        h = self._attn("decoder.mid_block.attentions.0", h, B, H * W)
        h = self._res("decoder.mid_block.resnets.1", h, B, H, W)
        n = len(self.cfg.block_out_channels)
        for i in range(n):
            for j in range(self.cfg.layers_per_block + 1):
                h = self._res(f"decoder.up_blocks.{i}.resnets.{j}", h, B, H, W)
            if i != n - 1:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                h, H, W = o.conv3(h, p, B, H, W, 2, p + ".bias")            # nearest x2 fused into the conv's gather
        h = o.groupnorm(h, "decoder.conv_norm_out", B, H * W, G, h.t.shape[1] // G, self.cfg.eps, True)
        h, _, _ = o.conv3(h, "decoder.conv_out", B, H, W, 0, "decoder.conv_out.bias")
        img = torch.empty((B, self.cfg.in_channels, H, W), device=self.device, dtype=torch.float32)
        k.nhwc_to_nchw(h.t, img, B, self.cfg.in_channels, H * W, _ld(h.t))
        return SimpleNamespace(sample=img) if return_dict else (img,)

    def encode(self, pixel_values, return_dict=True):
        mom, B, H, W = self.encode_moments(pixel_values)
        dist = _LatentDist(self, mom, B, H, W)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    def encode_latents(self, pixel_values, generator=None, noise=None):
        """trainer.py:2405-2406 in one call: sample() * scaling_factor, NCHW fp32 [B, 4, R/8, R/8]."""
        return self.encode(pixel_values).latent_dist.sample(generator, self.cfg.scaling_factor, noise)
