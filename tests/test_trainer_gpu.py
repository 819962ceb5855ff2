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
              "optimizer_1.bin", "scheduler.bin", "scheduler_1.bin", "random_states_0.pkl"):
        assert (ck / f).exists(), f
    assert (tmp_path / "checkpoint-4").exists()
    # optimizer.bin / scheduler.bin are what accelerator.save_state writes (trainer.py:452-477): a torch AdamW over the
    # student's parameters in the reference module's order loads them as they are, and so does a LambdaLR
    from pdm.models.unet.params import reference_param_order
    sd = tr.prediction_model.state_dict()
    order = reference_param_order(list(sd))
    assert order[0] == "conv_in.weight" and order[-1] == "conv_out.bias" and len(order) == len(sd)
    params = [torch.nn.Parameter(sd[n].clone()) for n in order]
    for fname, t_steps, base in (("optimizer.bin", 6, 1e-4), ("optimizer_1.bin", 2, 5e-4)):
        osd = torch.load(ck / fname, weights_only=False)
        topt = torch.optim.AdamW(params, lr=base, weight_decay=0.0)
        topt.load_state_dict(osd)
        st0 = topt.state[params[0]]
        assert float(st0["step"]) == t_steps and st0["exp_avg"].shape == params[0].shape
        assert all(topt.state[p]["exp_avg_sq"].shape == p.shape for p in params)
        sch = torch.optim.lr_scheduler.LambdaLR(topt, lambda k_: min(1.0, k_ / 2.0))
        sch.load_state_dict(torch.load(ck / fname.replace("optimizer", "scheduler"), weights_only=False))
        assert sch.last_epoch == t_steps and sch.get_last_lr() == [base]
    m_before = tr.stepper.opt.m.clone()
    # resume: weights and optimiser step counters come back, global_step parsed from the directory name
    tr2 = NudityBilevelUnetFineTuner(_config(tmp_path, 8, resume="latest"))
    tr2.load_checkpoint()
    assert tr2.global_step == 6 and tr2.stepper.opt.t == 6 and tr2.stepper.upper_opt.t == 2
    assert tr2.stepper.opt.sched_k == 6 and tr2.stepper.upper_opt.sched_k == 2
    assert torch.equal(tr2.stepper.opt.m, m_before) and torch.equal(tr2.stepper.upper_opt.v, tr.stepper.upper_opt.v)
    assert torch.equal(tr2.rng.get_state(), tr.rng.get_state())          # the noise / timestep stream continues
    a, b = tr.prediction_model.state_dict(), tr2.prediction_model.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_missing_weights_raise_unless_random_init_is_asked_for(dev, tmp_path):
    """A hub id / missing directory must not silently train random weights (the reference's from_pretrained raises)."""
    from pdm.training.trainer import BilevelUnetFineTuner
    cfg = _config(tmp_path, 1)
    cfg["synthetic"] = False
    cfg["pretrained_model_name_or_path"] = "stabilityai/stable-diffusion-2-1"
    with pytest.raises(FileNotFoundError):
        BilevelUnetFineTuner(cfg, train_dataloader=[{}], upper_dataloader=[{}])
    cfg["model"]["prediction_model"]["random_init"] = True           # the reference's random_init path: allowed
    BilevelUnetFineTuner(cfg, train_dataloader=[{}], upper_dataloader=[{}])


def test_epochs_validation_and_input_perturbation(dev, tmp_path):
    """train() re-iterates a short dataloader until max_train_steps (epochs, trainer.py:2769-2870); validate() logs the
    reference's validation/* keys without touching the weights; input_perturbation changes the forward process only."""
    from pdm.training.trainer import UnetFineTuner
    cfg = _config(tmp_path, 5)
    tr = UnetFineTuner(cfg)
    it = iter(tr.train_dataloader)
    two = [next(it), next(it)]
    tr.train_dataloader = two                                  # an "epoch" of 2 batches
    tr.train()
    assert tr.global_step == 5
    w = tr.prediction_model.store.master.clone()
    rec = tr.validate(two)
    assert set(rec) >= {"validation/loss", "validation/diffusion_loss", "validation/distillation_loss", "validation/block_loss"}
    assert rec["validation/loss"] > 0 and torch.equal(w, tr.prediction_model.store.master)
    assert float(tr.prediction_model.store.grad.abs().max()) == 0.0
    st = tr.rng.get_state()
    base = [float(x) for x in tr.step(two[0], backward=False)]
    tr.rng.set_state(st)
    cfg["model"]["prediction_model"]["input_perturbation"] = 0.1
    pert = [float(x) for x in tr.step(two[0], backward=False)]
    assert pert[1] != base[1]
    with pytest.raises(ValueError):
        c2 = _config(tmp_path, 1)
        c2["training"]["optim"]["lr_scheduler"] = "cosine"
        UnetFineTuner(c2)


def test_gradient_accumulation_steps_follow_the_reference_loops(dev, tmp_path):
    """training.gradient_accumulation_steps = k as the reference's fine-tune loops treat it (trainer.py:2769-2800: no
    `accelerator.accumulate` context, so the optimiser steps on every batch): every gradient is divided by k
    (`accelerator.backward`), the logged finetuning/loss is loss / k, and an epoch counts ceil(len(dataloader) / k) update steps
    when max_train_steps is derived from num_train_epochs (update_config_params, :445-450)."""
    from pdm.training.trainer import BilevelUnetFineTuner
    runs = []
    for k_ in (1, 2):
        cfg = _config(tmp_path / f"a{k_}", None)
        cfg["training"]["gradient_accumulation_steps"] = k_
        cfg["training"]["num_train_epochs"] = 1
        tr = BilevelUnetFineTuner(cfg)
        it = iter(tr.train_dataloader)
        tr.train_dataloader = [next(it) for _ in range(4)]           # an epoch of 4 batches
        tr.train()
        recs = [json.loads(l) for l in open(tmp_path / f"a{k_}" / "metrics.jsonl")]
        runs.append((tr.global_step, recs, tr.stepper._gscale, tr.prediction_model.store.master.clone()))
    (s1, r1, g1, w1), (s2, r2, g2, w2) = runs
    assert (s1, s2) == (4, 2) and (g1, g2) == (1.0, 0.5)               # 1 epoch = ceil(4 / k) optimiser steps; gradients / k
    for a, b in zip(r1, r2):                                           # same batches, same draws: step 0 is the same forward
        comp = b["finetuning/diffusion_loss"] + 2.0 * b["finetuning/distillation_loss"] + 0.1 * b["finetuning/block_loss"]
        assert abs(b["finetuning/loss"] - comp / 2) <= 1e-5 * abs(comp) + 1e-7, (b, comp)
    assert abs(r1[0]["finetuning/loss"] - 2 * r2[0]["finetuning/loss"]) <= 2e-2 * abs(r1[0]["finetuning/loss"])
    assert torch.isfinite(w2).all() and not torch.equal(w1, w2)


def test_hip_graph_mode_trains_like_eager_mode(dev, tmp_path):
    """`training.hip_graphs`: Trainer.train() replays the captured step (what bench.py measures) instead of eager launches;
    same seeded batches -> the same loss curve and the same final weights as the eager trainer (bf16 engine; split-K
    atomics order is the only difference), same cadence and log keys.
    In-process again (round 2 ran it in a child process): every captured graph is now single-stream, so hipGraphLaunch never
    enters hip::Graph::UpdateStreams, whose out-of-bounds read of the executor's parallel-stream list was the segfault
    (DESIGN.md 2)."""
    from pdm.training.trainer import BilevelUnetFineTuner
    runs = []
    # third run: `training.teacher_prefetch` - train() looks one batch ahead and the teacher's pass over batch t+1 runs beside step
    # t's backward; the draws keep the eager order (main t, upper t, main t+1), so it is the same curve again, and the generator
    # state a checkpoint stores is the one BEFORE the look-ahead draw (checkpoint-4 is written with batch 4 already drawn)
    for name, mode, pre in (("e", False, False), ("g", True, False), ("p", True, True)):
        cfg = _config(tmp_path / name, 6)
        cfg["training"]["hip_graphs"] = mode
        cfg["training"]["teacher_prefetch"] = pre
        tr = BilevelUnetFineTuner(cfg)
        tr.train()
        recs = [json.loads(l) for l in open(tmp_path / name / "metrics.jsonl")]
        runs.append((recs, tr.prediction_model.store.master.clone(), tr.stepper.opt.t, tr.stepper.upper_opt.t))
        if mode:
            assert len(tr._graphs) == 1
            assert next(iter(tr._graphs.values())).prefetch_hits == (7 if pre else 0)      # main steps 1 .. 5, upper steps 2 and 5
    (re, we, te, ue) = runs[0]
    for (rg, wg, tg, ug) in runs[1:]:
        assert (te, ue) == (tg, ug) == (6, 2) and len(re) == len(rg) == 6
        assert [sorted(r) for r in re] == [sorted(r) for r in rg]
        for a, b in zip(re, rg):
            for key in a:
                assert abs(a[key] - b[key]) <= 2e-2 * abs(a[key]) + 1e-7, (key, a[key], b[key])
        d = (we - wg).abs()
        assert d.max().item() <= 5e-3 and d.mean().item() <= 2e-3 * we.abs().mean().item() + 1e-6, (d.max().item(), d.mean().item())
    import pickle
    states = [pickle.load(open(tmp_path / name / "checkpoint-4" / "random_states_0.pkl", "rb"))["pdm_generator_state"]
              for name in ("g", "p")]
    assert torch.equal(states[0], states[1])


def test_hip_graph_mode_keeps_one_capture_per_batch_shape(dev, tmp_path):
    """A dataloader whose last batch of every epoch is ragged (3 batches of B = 2, 2, 1; two epochs): graph mode captures each
    batch shape ONCE and keeps it (round 2 dropped and re-captured the executors at every shape change, the sequence that
    crashed hipGraphLaunch); the cache holds `training.hip_graph_shapes` shapes and closes the least recently used one at
    an idle point; clip_grad_norm with hip_graphs is rejected when the trainer is built, not at the first step."""
    from pdm.training import bilevel
    from pdm.training.trainer import BilevelUnetFineTuner, SyntheticBatches
    cfg = _config(tmp_path, 6)
    cfg["training"]["hip_graphs"] = True
    cfg["training"]["hip_graph_shapes"] = 2
    full = list(b for _, b in zip(range(2), SyntheticBatches(2, 16, 13, 64, 7, "cuda")))
    tail = {k_: v[:1].contiguous() for k_, v in full[0].items()}
    captures = []
    orig = bilevel.GraphedBilevel.capture
    bilevel.GraphedBilevel.capture = lambda self, bilevel=True: (captures.append(self.shape), orig(self, bilevel))[1]
    try:
        tr = BilevelUnetFineTuner(cfg, train_dataloader=full + [tail], upper_dataloader=[full[0]])
        tr.train()
        assert tr.global_step == 6 and tr.stepper.opt.t == 6
        assert sorted(s_[0] for s_ in captures) == [1, 2], captures            # one capture per shape over two epochs
        assert len(tr._graphs) == 2
        # a third shape evicts the least recently used one (closed, not just dropped)
        first = next(iter(tr._graphs.values()))
        three = {k_: torch.cat([v, v[:1]]) for k_, v in full[0].items()}
        tr.step(three)
        assert len(tr._graphs) == 2 and first.closed and len(captures) == 3
        torch.cuda.synchronize()
        assert torch.isfinite(tr.prediction_model.store.master).all()
    finally:
        bilevel.GraphedBilevel.capture = orig
    c2 = _config(tmp_path / "clip", 1)
    c2["training"]["hip_graphs"] = True
    c2["training"]["optim"]["clip_grad_norm"] = True
    c2["training"]["optim"]["max_grad_norm"] = 1.0
    with pytest.raises(ValueError):
        BilevelUnetFineTuner(c2)


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


def test_local_snapshot_and_pruning_checkpoint_flow_in(dev, tmp_path):
    """a local model directory laid out like the hub snapshot the reference loads (trainer.py:2126-2176:
    unet/, vae/, text_encoder/ safetensors under diffusers / transformers key names) plus a pruning checkpoint
    (`quantizer_embeddings.pt[expert_id]`, trainer.py:2159-2161) build the same models as handing the tensors over."""
    from safetensors.torch import save_file
    from pdm_ref import arch as oarch, weights as oweights, vae as ovae, clip_text
    from pdm_ref.config import UNetConfig as OCfg
    from pdm.training.trainer import BilevelUnetFineTuner
    root, ck = tmp_path / "snapshot", tmp_path / "pruning"
    for d in (root / "unet", root / "vae", root / "text_encoder", ck):
        d.mkdir(parents=True)
    ocfg = OCfg.tiny()
    dense = oweights.init_dense_state_dict(ocfg, seed=11)
    save_file({n: t.contiguous() for n, t in dense.items()}, str(root / "unet" / "diffusion_pytorch_model.safetensors"))
    vcfg = ovae.VAEConfig.tiny()
    vsd = ovae.init_state_dict(vcfg, seed=12)
    save_file({n: t.contiguous() for n, t in vsd.items()}, str(root / "vae" / "diffusion_pytorch_model.safetensors"))
    tcfg = clip_text.CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=256, num_hidden_layers=2,
                                    num_attention_heads=1)
    tsd = clip_text.init_state_dict(tcfg, seed=13, prefix="text_model.")
    save_file({n: t.contiguous() for n, t in tsd.items()}, str(root / "text_encoder" / "model.safetensors"))
    experts = torch.stack([oarch.random_arch_vector(ocfg, kr, seed=s)[0] for s, kr in ((1, 0.5), (2, 0.7), (3, 0.9))])
    torch.save(experts, str(ck / "quantizer_embeddings.pt"))
    cfg = _config(tmp_path / "logs", 1)
    cfg["pretrained_model_name_or_path"], cfg["pruning_ckpt_dir"], cfg["expert_id"] = str(root), str(ck), 1
    tr = BilevelUnetFineTuner(cfg)
    assert torch.equal(tr.arch_vector, experts[1][None])
    # student = dense weights physically pruned by expert 1's arch vector; teacher = the dense weights
    psd, _ = oweights.prune_state_dict(dense, ocfg, experts[1][None])
    got = tr.prediction_model.state_dict()
    assert set(got) == set(psd) and all(torch.equal(got[n], psd[n]) for n in psd)
    tt = tr.teacher_model.state_dict()
    assert all(torch.equal(tt[n], dense[n]) for n in dense)
    # frozen encoders come from the same snapshot
    vs = tr.vae.state_dict()
    assert all(torch.equal(vs[n], vsd[n]) for n in vsd)
    ts = tr.text_encoder.state_dict()
    assert all(torch.equal(ts[n], tsd[n]) for n in tsd)
    # one step on the reference's raw batch schema (pixels + token ids), then the written checkpoint reloads bit-exactly
    g = torch.Generator().manual_seed(0)
    batch = {"pixel_values": torch.rand(2, 3, 64, 64, generator=g) * 2 - 1, "input_ids": torch.randint(0, 1000, (2, 13), generator=g),
             "empty_input_ids": torch.zeros(2, 13, dtype=torch.int64)}
    loss = tr.step(batch)
    tr.stepper.optimizer_step(upper=False)
    up = tr.upper_step(batch)
    assert all(torch.isfinite(x).all() for x in loss + up)
    tr.global_step = 1
    tr.save_checkpoint()
    from pdm.models.unet.unet_2d_conditional import UNet2DConditionModelPruned
    from pdm.models.unet.spec import UNetConfig
    re = UNet2DConditionModelPruned.from_pretrained(str(tmp_path / "logs" / "checkpoint-1"), subfolder="unet",
                                                    unet_config=UNetConfig.tiny(), torch_dtype=torch.bfloat16, device=dev)
    assert torch.equal(re.arch_vector, tr.arch_vector)
    a, b = tr.prediction_model.state_dict(), re.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[n], b[n]) for n in a)


def test_image_logging_samples_with_the_current_student(dev, tmp_path):
    """trainer.py:2543-2575 / :2851-2859: every `image_logging_steps` the prompt batches are sampled with the student
    (PNDM + guidance + VAE decode) - written as uint8 arrays here instead of a wandb grid; seeded, so repeatable."""
    import numpy as np
    from pdm.training.trainer import UnetFineTuner
    cfg = _config(tmp_path, 2)
    cfg["training"]["image_logging_steps"] = 1
    cfg["training"]["num_inference_steps"] = 3
    ids = torch.randint(0, 1000, (2, 13), generator=torch.Generator().manual_seed(0))
    prompts = [{"input_ids": ids, "empty_input_ids": torch.zeros(2, 13, dtype=torch.int64)}]
    tr = UnetFineTuner(cfg, prompt_dataloader=prompts)
    tr.train()
    files = sorted(os.listdir(tmp_path / "images"))
    assert files == ["step-0.npy", "step-1.npy"]
    a, b = np.load(tmp_path / "images" / files[0]), np.load(tmp_path / "images" / files[1])
    assert a.shape == (2, 64, 64, 3) and a.dtype == np.uint8 and a.std() > 0
    assert not np.array_equal(a, b)                       # the student moved between the two logging points
    again = tr.generate_samples_from_prompts()            # same weights + same seed -> same images
    assert np.array_equal((again.permute(0, 2, 3, 1) * 255).round().to(torch.uint8).cpu().numpy(),
                          (tr.generate_samples_from_prompts().permute(0, 2, 3, 1) * 255).round().to(torch.uint8).cpu().numpy())
