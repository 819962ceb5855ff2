"""CLIP text conditioning path (-m gpu): libpdmk through the C ABI against the oracle
(oracle/pdm_ref/clip_text.py) and against the outputs of transformers.CLIPTextModel itself committed in
tests/golden/clip_text_hf.npz.  Tolerances: fp32 3e-4, bf16 5e-2 of the output scale for the whole encoder."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "bf16": torch.bfloat16}
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_text_hf.npz"))


def close(got, ref, tol, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert math.isfinite(err) and err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


@pytest.mark.parametrize("dn,tol", [("f32", 2e-4), ("bf16", 2e-2)])
@pytest.mark.parametrize("B,H,N", [(2, 2, 77), (1, 3, 64), (2, 1, 130), (1, 16, 5)])
def test_causal_attention(dev, dn, tol, B, H, N):
    from pdm import _pdmk as k
    torch.manual_seed(0)
    E = H * 64
    qkv = torch.randn(B * N, 3 * E, device=dev).to(DT[dn])
    o = torch.zeros(B * N, E, device=dev, dtype=DT[dn])
    lse = torch.zeros(B, H, N, device=dev)
    st = (N * 3 * E, 3 * E)
    k.attn_fwd_causal(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], o, lse, B, H, N, st, st, st, (N * E, E), 0.125)
    q, kk, v = (t.float().view(B, N, H, 64).transpose(1, 2) for t in (qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]))
    ref = F.scaled_dot_product_attention(q, kk, v, is_causal=True).transpose(1, 2).reshape(B * N, E)
    close(o, ref, tol, "causal attention")
    # row 0 of every (b, h) sees key 0 only: output = v[0]
    close(o.view(B, N, E)[:, 0], qkv.view(B, N, 3 * E)[:, 0, 2 * E:], 1e-6 if dn == "f32" else 1e-2, "first row")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_gelu_and_embedding(dev, dn):
    from pdm import _pdmk as k
    torch.manual_seed(1)
    x = (torch.randn(37, 1000, device=dev) * 3).to(DT[dn])
    y = torch.empty_like(x)
    k.gelu_fwd(x, y)
    close(y, F.gelu(x.float()), 1e-5 if dn == "f32" else 1e-2, "erf gelu")
    V, T, D, B = 50, 7, 64, 3
    tok, pos = torch.randn(V, D, device=dev).to(DT[dn]), torch.randn(T + 3, D, device=dev).to(DT[dn])
    ids = torch.randint(0, V, (B, T), device=dev)
    ids[0, 0], ids[0, 1] = -4, V + 9                       # out-of-range ids are clamped, never read out of bounds
    out = torch.empty(B * T, D, device=dev, dtype=DT[dn])
    k.embed_tokens(ids, tok, pos, out, B * T, T, D, V, D, D, D)
    ref = tok.float()[ids.clamp(0, V - 1)] + pos.float()[:T]
    close(out.view(B, T, D), ref, 1e-6 if dn == "f32" else 1e-2, "token + position embedding")


def _model(cfg_o, dn, dev, seed=5):
    from pdm.models.clip.text_encoder import CLIPTextModel, CLIPTextConfig
    from pdm_ref import clip_text
    sd = clip_text.init_state_dict(cfg_o, seed=seed, prefix="text_model.")
    m = CLIPTextModel(CLIPTextConfig(**cfg_o.__dict__), dev, DT[dn], init=False)
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize("dn,tol", [("f32", 3e-4), ("bf16", 5e-2)])
@pytest.mark.parametrize("tag", ["tiny", "wide"])
def test_encoder_matches_transformers_outputs(dev, dn, tol, tag):
    """HIP path vs the committed outputs of transformers.CLIPTextModel (the class the reference instantiates)."""
    from pdm_ref import clip_text
    cfg_o = clip_text.CLIPTextConfig.tiny() if tag == "tiny" else clip_text.CLIPTextConfig(
        vocab_size=2000, hidden_size=1024, intermediate_size=4096, num_hidden_layers=2, num_attention_heads=16)
    m, sd = _model(cfg_o, dn, dev)
    ids = torch.from_numpy(GOLD[f"{tag}_ids"])
    out = m(ids.to(dev))
    assert out[0].shape == (ids.shape[0], 77, cfg_o.hidden_size) and out.last_hidden_state is out[0]
    close(out[0], torch.from_numpy(GOLD[f"{tag}_out"]), tol, f"last_hidden_state[{tag}]")
    back = m.state_dict()
    assert set(back) == set(sd) and all(torch.equal(back[n], sd[n]) for n in sd)


def test_sd21_text_encoder_matches_oracle_and_encode_prompt(dev):
    """The real SD-2.1 shape (23 layers, 1024 wide, 16 heads, vocab 49408), bf16, captions + empty prompts in one batch."""
    from pdm.models.clip.text_encoder import encode_prompt
    from pdm_ref import clip_text
    cfg_o = clip_text.CLIPTextConfig.sd21()
    m, sd = _model(cfg_o, "bf16", dev, seed=2)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, cfg_o.vocab_size, (2, 77), generator=g)
    ids[1, 1:] = 0                                        # an "empty prompt" row: BOS then padding
    emb = encode_prompt(None, m, ["ignored"], text_input_ids=ids)
    assert emb.shape == (2, 77, 1024) and emb.dtype == torch.bfloat16
    with torch.no_grad():
        ref = clip_text.encode(sd, cfg_o, ids)
    close(emb, ref, 5e-2, "sd21 text encoder")
    cos = F.cosine_similarity(emb.float().cpu().flatten(), ref.flatten(), dim=0).item()
    assert cos > 0.999, cos
    with pytest.raises(ValueError):
        encode_prompt(None, m, "x")
