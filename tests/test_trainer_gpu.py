"""-m gpu: the Trainer classes end to end on the tiny topology with synthetic batches: bilevel cadence (upper step on
every `upper_step_freq`-th iteration with its own AdamW), metric keys of trainer.py:2819-2834, checkpoint layout
(trainer.py:452-477, 2863-2869) and resume."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _config(tmp, steps, resume=None):
    from pdm.utils.config import Cfg
    return Cfg.wrap({
        "seed": 43, "synthetic": True, "tiny": True, "mixed_precision": "bf16", "keep_ratio": 0.7,
        "model": {"prediction_model": {"prediction_type": "v_prediction", "resolution": 128, "gated_ff": True,
                                       "ff_gate_width": 32}},
        "data": {"dataloader": {"train_batch_size": 2}},
        "training": {"max_train_steps": steps, "upper_step_freq": 3,
                     "losses": {"diffusion_loss": {"snr_gamma": 5.0, "weight": 1.0},
                                "distillation_loss": {"weight": 2.0, "upper_weight": 1.0},
                                "block_loss": {"weight": 0.1, "upper_weight": 0.0}},
                     "optim": {"prediction_model_learning_rate": 1e-4, "prediction_model_upper_learning_rate": 5e-4,
                               "lr_warmup_steps": 2},
                     "logging": {"logging_dir": str(tmp), "checkpoint_steps": 4, "resume_from_checkpoint": resume}}})


def test_bilevel_trainer_runs_logs_checkpoints_and_resumes(dev, tmp_path):
    from pdm.training.trainer import BilevelUnetFineTuner, NudityBilevelUnetFineTuner
    tr = BilevelUnetFineTuner(_config(tmp_path, 6))
    w0 = tr.prediction_model.store.master.clone()
    tr.train()
    assert tr.global_step == 6
    assert not torch.equal(w0, tr.prediction_model.store.master)
    recs = [json.loads(l) for l in open(tmp_path / "metrics.jsonl")]
    assert len(recs) == 6
    assert all("finetuning/loss" in r and "finetuning/block_loss" in r for r in recs)
    upper = [r["step"] for r in recs if "finetuning/upper_loss" in r]
    assert upper == [2, 5]                                   # (global_step + 1) % 3 == 0, trainer.py:2795
    assert all(r["finetuning/loss"] == r["finetuning/loss"] and r["finetuning/loss"] > 0 for r in recs)   # finite
    assert recs[0]["finetuning/prediction_model_lr"] == 0.0 and recs[-1]["finetuning/prediction_model_lr"] == 1e-4
    ck = tmp_path / "checkpoint-6"
    for f in ("unet/diffusion_pytorch_model.safetensors", "unet/config.json", "arch_vector.pt", "optimizer.bin",
              "optimizer_1.bin", "random_states_0.pkl"):
        assert (ck / f).exists(), f
    assert (tmp_path / "checkpoint-4").exists()
    # resume: weights and optimiser step counters come back, global_step parsed from the directory name
    tr2 = NudityBilevelUnetFineTuner(_config(tmp_path, 8, resume="latest"))
    tr2.load_checkpoint()
    assert tr2.global_step == 6 and tr2.stepper.opt.t == 6 and tr2.stepper.upper_opt.t == 2
    a, b = tr.prediction_model.state_dict(), tr2.prediction_model.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_pixel_batches_go_through_the_vae(dev, tmp_path):
    """The reference's batch schema (`pixel_values`, trainer.py:2405-2406): latents = vae.encode(...).sample() * 0.18215,
    drawn from the trainer's generator BEFORE the diffusion noise, then the same step as with pre-encoded latents."""
    from pdm.training.trainer import UnetFineTuner
    cfg = _config(tmp_path, 1)
    cfg["synthetic_pixels"] = True
    tr = UnetFineTuner(cfg)
    batch = next(iter(tr.train_dataloader))
    assert batch["pixel_values"].shape == (2, 3, 64, 64) and "latents" not in batch
    st = tr.rng.get_state()
    loss = tr.step(batch)
    assert all(torch.isfinite(x).all() for x in loss) and loss[0] > 0
    # the same latents handed over directly reproduce the step's losses (fresh trainer: same weights, same RNG stream)
    tr2 = UnetFineTuner(cfg)
    tr2.rng.set_state(st)
    lat = tr2.vae.encode_latents(batch["pixel_values"], generator=tr2.rng)
    assert lat.shape == (2, 4, 16, 16)
    loss2 = tr2.step({"latents": lat, "prompt_embeds": batch["prompt_embeds"]})
    for a, b in zip(loss, loss2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    with pytest.raises(KeyError):
        tr.step({"prompt_embeds": batch["prompt_embeds"]})


def test_token_id_batches_go_through_the_text_encoder(dev, tmp_path):
    """`input_ids` / `empty_input_ids` batches: prompt_embeds = text_encoder(ids)[0] (data_utils.py:155-191) computed once
    per step on the device; the empty prompt is encoded once and cached; same losses as with the embeddings handed over."""
    from pdm.training.trainer import BilevelUnetFineTuner
    cfg = _config(tmp_path, 1)
    tr = BilevelUnetFineTuner(cfg)
    b0 = next(iter(tr.train_dataloader))
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 1000, (2, 13), generator=g)
    empty = torch.tensor([[3] + [0] * 12]).expand(2, -1)
    batch = {"latents": b0["latents"], "input_ids": ids, "empty_input_ids": empty}
    st = tr.rng.get_state()
    loss = tr.upper_step(batch)
    assert all(torch.isfinite(x).all() for x in loss)
    assert len(tr._empty_cache) == 1
    tr.upper_step(batch)
    assert len(tr._empty_cache) == 1                          # cached: not encoded again
    tr2 = BilevelUnetFineTuner(cfg)
    tr2.rng.set_state(st)
    pe, ee = tr2.text_encoder(ids)[0], tr2.text_encoder(empty)[0]
    assert pe.shape == (2, 13, 64)
    loss2 = tr2.upper_step({"latents": b0["latents"], "prompt_embeds": pe, "empty_prompt_embeds": ee})
    for a, b in zip(loss, loss2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
