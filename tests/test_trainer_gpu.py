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


def test_missing_pixels_path_fails_loudly(dev, tmp_path):
    from pdm.training.trainer import UnetFineTuner
    tr = UnetFineTuner(_config(tmp_path, 1))
    with pytest.raises(NotImplementedError):
        tr.step({"pixel_values": torch.zeros(1, 3, 8, 8), "prompt_embeds": torch.zeros(1, 13, 64)})
