#!/usr/bin/env python3
"""Pin oracle/pdm_ref/clip_text.py against `transformers.CLIPTextModel` - the class the reference instantiates for
SD-2.1's text encoder (pdm/training/trainer.py:2126-2131, called at pdm/utils/data_utils.py:183) - and emit a golden
fixture.  Runs in the build container (transformers is importable here, version printed into the report); the fixture
holds token ids + the transformers outputs (plain data); weights are regenerated from the seed by the oracle's
init_state_dict, which is also what was loaded into the transformers model."""
import os
import sys

import numpy as np
import torch
import transformers
from transformers import CLIPTextConfig as HFConfig, CLIPTextModel

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from pdm_ref import clip_text  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
report = [f"transformers {transformers.__version__}"]
golden = {}
for tag, cfg, B in (("tiny", clip_text.CLIPTextConfig.tiny(), 3),
                    ("wide", clip_text.CLIPTextConfig(vocab_size=2000, hidden_size=1024, intermediate_size=4096,
                                                      num_hidden_layers=2, num_attention_heads=16), 2)):
    hf = CLIPTextModel(HFConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                                max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5,
                                bos_token_id=0, eos_token_id=cfg.vocab_size - 1, pad_token_id=1)).eval()
    prefix = "text_model." if any(k.startswith("text_model.") for k in hf.state_dict()) else ""
    sd = clip_text.init_state_dict(cfg, seed=5)
    missing = hf.load_state_dict({prefix + k: v for k, v in sd.items()}, strict=False)
    assert not [k for k in missing.missing_keys if "position_ids" not in k], missing
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(0, cfg.vocab_size, (B, 77), generator=g)
    with torch.no_grad():
        ref = hf(ids)[0]
        mine = clip_text.encode(sd, cfg, ids)
    err = float((mine - ref).abs().max())
    line = f"{'OK ' if err < 2e-5 else 'FAIL'} CLIPTextModel[{tag}] last_hidden_state: max|diff|={err:.3e} (scale {float(ref.abs().max()):.3e})"
    print(line)
    report.append(line)
    assert err < 2e-5
    golden[f"{tag}_ids"], golden[f"{tag}_out"] = ids.numpy(), ref.numpy()
np.savez_compressed(os.path.join(GOLD, "clip_text_hf.npz"), **golden)
open(os.path.join(GOLD, "clip_text_hf.report.txt"), "w").write("\n".join(report) + "\n")
print("fixtures ->", GOLD)
