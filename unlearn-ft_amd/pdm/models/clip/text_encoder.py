"""`CLIPTextModel` on libpdmk - the text conditioning path.

The reference encodes captions inside its dataset transform, per sample and twice (caption + empty prompt), with
transformers' CLIPTextModel on the training device (pdm/utils/data_utils.py:155-191, 247-276; model loaded at
pdm/training/trainer.py:2126-2131):   prompt_embeds = text_encoder(text_input_ids)[0]   # [B, 77, 1024]
Same call surface here (`from_pretrained(path, subfolder="text_encoder")`, `model(input_ids)[0]` /
`.last_hidden_state`, transformers state-dict key names with or without the `text_model.` prefix); arithmetic in
libpdmk: fused token+position gather, LayerNorm, one fused q|k|v projection, causal flash attention (head dim 64, the
U-Net's kernel with a mask), erf-GELU MLP, residuals in the GEMM epilogues.  Inference only (frozen); no CPU path.
Tokenisation stays on the host (transformers' CLIPTokenizer needs its vocabulary files): callers pass token ids.
"""
import gc
import os
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from ... import _pdmk as k
from ..unet.engine import Act, _ld
from ..unet.params import ParamStore, _lin, _vec
from ..unet.spec import padc
from ..vae.autoencoder_kl import _Ops


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


def build_entries(cfg: CLIPTextConfig):
    E, F = cfg.hidden_size, cfg.intermediate_size
    out = [_lin("embeddings.token_embedding", [("embeddings.token_embedding.weight", cfg.vocab_size)], E),
           _lin("embeddings.position_embedding", [("embeddings.position_embedding.weight", cfg.max_position_embeddings)], E)]

    def norm(key):
        out.extend([_vec(key + ".weight", [(key + ".weight", E)]), _vec(key + ".bias", [(key + ".bias", E)])])

    def lin(key, srcs, kin):
        out.extend([_lin(key, [(n + ".weight", r) for n, r in srcs], kin), _vec(key + ".bias", [(n + ".bias", r) for n, r in srcs])])

    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}"
        norm(p + ".layer_norm1")
        lin(p + ".self_attn.qkv_proj", [(f"{p}.self_attn.{n}_proj", E) for n in ("q", "k", "v")], E)
        lin(p + ".self_attn.out_proj", [(p + ".self_attn.out_proj", E)], E)
        norm(p + ".layer_norm2")
        lin(p + ".mlp.fc1", [(p + ".mlp.fc1", F)], E)
        lin(p + ".mlp.fc2", [(p + ".mlp.fc2", E)], F)
    norm("final_layer_norm")
    off = 0
    for e in out:
        e.off = off
        off += (e.numel + 127) // 128 * 128
    return out


class _Output(tuple):
    """`model(ids)[0]` and `model(ids).last_hidden_state`, like transformers' BaseModelOutputWithPooling."""

    @property
    def last_hidden_state(self):
        return self[0]


class CLIPTextModel:
    def __init__(self, cfg: CLIPTextConfig = None, device=None, dtype=torch.bfloat16, seed=0, init=True):
        if not torch.cuda.is_available():
            raise RuntimeError("CLIPTextModel (MI355X engine) needs a GPU; there is no CPU fallback")
        self.cfg = cfg or CLIPTextConfig.sd21()
        assert self.cfg.hidden_size // self.cfg.num_attention_heads == 64, "attention kernels are specialised for head dim 64"
        assert self.cfg.hidden_size % 32 == 0 and self.cfg.intermediate_size % 32 == 0
        self.device = torch.device(device or "cuda:0")
        self.dtype = dtype
        self.store = ParamStore(build_entries(self.cfg), self.device, dtype, train=False)
        self.ops = _Ops(self.store, dtype)
        self.config = SimpleNamespace(**self.cfg.__dict__)
        # ~10 launches per layer on 77-token inputs are launch-bound from Python (3.2 ms eager for 23 layers): each
        # (B, T) shape is captured once as a hipGraph and replayed on a static id buffer
        self.use_graph = os.environ.get("PDMK_CLIP_GRAPH", "1") != "0"
        self._graphs = {}
        if init:
            self.store.init_random(seed)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, revision=None, random_init=False,
                        text_config=None, torch_dtype=torch.bfloat16, device=None, seed=0, **unused):
        path = pretrained_model_name_or_path
        if path and subfolder:
            path = os.path.join(path, subfolder)
        have_local = bool(path) and os.path.isdir(path)
        model = cls(text_config, device, torch_dtype, seed=seed, init=random_init or not have_local)
        if have_local and not random_init:
            f = os.path.join(path, "model.safetensors")
            if os.path.exists(f):
                from safetensors.torch import load_file
                sd = load_file(f)
            else:
                sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
            model.load_state_dict(sd)
        elif not random_init:
            raise FileNotFoundError(f"{pretrained_model_name_or_path!r} is not a local directory and hub downloads are "
                                    f"not available here; pass random_init=True or a local checkpoint directory")
        return model

    def load_state_dict(self, sd, strict=True):
        own = {}
        for key, v in sd.items():
            key = key[len("text_model."):] if key.startswith("text_model.") else key
            if key.endswith("position_ids"):
                continue
            own[key] = v
        self.store.load_state_dict(own, strict=strict)

    def state_dict(self, prefix="text_model."):
        return {prefix + n: t for n, t in self.store.state_dict().items()}

    def requires_grad_(self, flag=False):
        return self

    def to(self, *a, **kw):
        return self

    def eval(self):
        return self

    @property
    def dtype_(self):
        return self.dtype

    # ------------------------------------------------------------------ forward
    def encode_2d(self, input_ids):
        """ids [B, T] (T <= 77) -> last hidden state as a 2-D [B*T, hidden] matrix in the compute dtype."""
        cfg, o, P = self.cfg, self.ops, self.store
        B, T = input_ids.shape
        assert T <= cfg.max_position_embeddings
        E, H = cfg.hidden_size, cfg.num_attention_heads
        ids = input_ids.to(self.device, torch.int64).contiguous()
        x = torch.empty((B * T, E), device=self.device, dtype=self.dtype)
        k.embed_tokens(ids, P.wv("embeddings.token_embedding.weight"), P.wv("embeddings.position_embedding.weight"), x,
                       B * T, T, E, cfg.vocab_size, E, E, E)
        x = Act(x, rg=False)
        lse = torch.empty((B, H, T), device=self.device, dtype=torch.float32)
        for i in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{i}"
            h = o.layernorm(x, p + ".layer_norm1")
            qkv = o.linear(h, p + ".self_attn.qkv_proj", bias=p + ".self_attn.qkv_proj.bias").t
            att = torch.empty((B * T, E), device=self.device, dtype=self.dtype)
            q, kk, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:3 * E]
            st = (T * _ld(qkv), _ld(qkv))
            k.attn_fwd_causal(q, kk, v, att, lse, B, H, T, st, st, st, (T * E, E), 64 ** -0.5)
            x = o.linear(Act(att), p + ".self_attn.out_proj", bias=p + ".self_attn.out_proj.bias", residual=x)
            h = o.layernorm(x, p + ".layer_norm2")
            f = o.linear(h, p + ".mlp.fc1", bias=p + ".mlp.fc1.bias").t
            a = torch.empty_like(f)
            k.gelu_fwd(f, a)
            x = o.linear(Act(a), p + ".mlp.fc2", bias=p + ".mlp.fc2.bias", residual=x)
        return o.layernorm(x, "final_layer_norm").t

    def _replay(self, input_ids):
        key = tuple(input_ids.shape)
        ent = self._graphs.get(key)
        if ent is None:
            static_ids = input_ids.to(self.device, torch.int64).contiguous().clone()
            self.encode_2d(static_ids)                    # eager warm-up: GEMM plans are tuned outside the capture
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            gc.collect()
            gc.disable()                                  # a collection during capture would free graph-pool tensors
            try:
                with torch.cuda.graph(graph):
                    out = self.encode_2d(static_ids)
            finally:
                gc.enable()
            ent = self._graphs[key] = (graph, static_ids, out)
        graph, static_ids, out = ent
        static_ids.copy_(input_ids)
        graph.replay()
        return out.clone()

    def __call__(self, input_ids, output_hidden_states=False, **unused):
        B, T = input_ids.shape
        capturing = torch.cuda.is_current_stream_capturing()
        y = self._replay(input_ids) if self.use_graph and not capturing else self.encode_2d(input_ids)
        return _Output((y.view(B, T, self.cfg.hidden_size),))


def encode_prompt(tokenizer, text_encoder, prompt, max_sequence_length=77, device=None, text_input_ids=None, pooled=False):
    """pdm/utils/data_utils.py:155-191 with the same signature: tokenise on the host when a tokenizer is given, else take
    `text_input_ids`; returns prompt_embeds [B, T, hidden] in the encoder's dtype."""
    if pooled:
        raise NotImplementedError("pooled CLIP output is only used by the SDXL/Flux trainers (out of scope, SURVEY 2.1)")
    prompt = [prompt] if isinstance(prompt, str) else prompt
    if tokenizer is not None:
        text_input_ids = tokenizer(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                                   return_length=False, return_overflowing_tokens=False, return_tensors="pt").input_ids
    elif text_input_ids is None:
        raise ValueError("text_input_ids must be provided when the tokenizer is not specified")
    return text_encoder(text_input_ids)[0]
