"""Trainer classes with the reference's names and call contract (pdm/training/trainer.py:2116-3016), running on the
MI355X engine:  UnetFineTuner(config).train(),  BilevelUnetFineTuner(config).train(),  NudityBilevelUnetFineTuner.

Differences that are deliberate:
 * `step(batch)` / `upper_step(batch)` return the reference's 4-tuple (loss, diff_loss, distillation_loss, block_loss)
   but ALSO run the hand-written backward (there is no autograd graph to hand to accelerator.backward);
 * no accelerate / wandb: ranks come from torch.distributed (RCCL), metrics go to <logging_dir>/metrics.jsonl with the
   reference's key names (trainer.py:2819-2834) and are read back asynchronously (no per-step .item() syncs);
 * batches carry the reference's `pixel_values` (encoded by the frozen VAE on libpdmk, trainer.py:2405-2406) or
   pre-encoded `latents`; `--synthetic` produces seeded batches of either kind (`synthetic_pixels: true` for pixels);
   text conditioning comes as the reference's `prompt_embeds` / `empty_prompt_embeds`, or as token ids (`input_ids` /
   `empty_input_ids`) encoded here by the frozen CLIP text encoder on libpdmk (data_utils.py:155-191) - one batched
   call per step instead of two per sample inside the dataloader, the empty-prompt embedding cached per distinct row;
 * torch.autograd.set_detect_anomaly (scripts/aptp/*.py:21) is not reproduced.
"""
import glob
import json
import logging
import os
import pickle
import time

import torch
import torch.distributed as dist

from ..models.unet.spec import UNetConfig, arch_vector_for_budget
from ..models.unet.unet_2d_conditional import UNet2DConditionModelPruned
from .bilevel import BilevelStepper

logger = logging.getLogger("pdm.trainer")


def _cfg(config, path, default=None):
    cur = config
    for part in path.split("."):
        if cur is None:
            return default
        cur = cur.get(part) if isinstance(cur, dict) else getattr(cur, part, None)
    return default if cur is None else cur


class SyntheticBatches:
    """Seeded batches of the collate_fn schema (pdm/utils/data_utils.py:286-312): `pixel_values` in [-1, 1] (pixels=True)
    or pre-encoded `latents`, plus prompt embeddings."""

    def __init__(self, batch_size, hw, ctx_len, ctx_dim, seed, device, length=1 << 30, pixels=False, vae_factor=8):
        self.bs, self.hw, self.T, self.D, self.device, self.length = batch_size, hw, ctx_len, ctx_dim, device, length
        self.pixels, self.res = pixels, hw * vae_factor
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.empty = torch.randn(1, ctx_len, ctx_dim, generator=torch.Generator().manual_seed(1234)).to(device)

    def __len__(self):
        return self.length

    def __iter__(self):
        for _ in range(self.length):
            img = ({"pixel_values": torch.rand(self.bs, 3, self.res, self.res, device=self.device, generator=self.gen) * 2 - 1}
                   if self.pixels else
                   {"latents": torch.randn(self.bs, 4, self.hw, self.hw, device=self.device, generator=self.gen)})
            yield {**img, "prompt_embeds": torch.randn(self.bs, self.T, self.D, device=self.device, generator=self.gen),
                   "empty_prompt_embeds": self.empty.expand(self.bs, -1, -1).contiguous()}


class Trainer:
    bilevel = False

    def __init__(self, config, train_dataloader=None, upper_dataloader=None, prompt_dataloader=None):
        self.config = config
        self.prompt_dataloader = prompt_dataloader     # batches of {"input_ids" | "prompt_embeds", "empty_input_ids" | ...}
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        local = int(os.environ.get("LOCAL_RANK", 0))
        if not torch.cuda.is_available():
            raise RuntimeError("the MI355X trainer needs a GPU (no CPU fallback for the hot path)")
        torch.cuda.set_device(local)
        self.device = torch.device("cuda", local)
        self.logging_dir = _cfg(config, "training.logging.logging_dir", _cfg(config, "logging_dir", "logs"))
        self.global_step = 0
        self.init_weight_dtype()
        self.init_models()
        self.init_optimizer()
        self.train_dataloader = train_dataloader or self.init_dataloader(upper=False)
        self.upper_dataloader = (upper_dataloader or self.init_dataloader(upper=True)) if self.bilevel else None
        self.rng = torch.Generator(device=self.device).manual_seed(int(_cfg(config, "seed", 43)) + self.rank)
        # `training.hip_graphs: true` (CLI --hip_graphs): the step runs as hipGraph replay (GraphedBilevel: what bench.py
        # measures - captured forward / loss heads / segmented backward, AdamW streamed behind the bucketed all-reduce)
        # instead of eager launches.  Needs fixed batch shapes; no gradient clipping / input perturbation inside a graph.
        self.hip_graphs = bool(_cfg(config, "training.hip_graphs", _cfg(config, "hip_graphs", False)))
        # captured steps by batch shape (the ragged last batch of an epoch has its own): a shape is captured once and kept;
        # at most `training.hip_graph_shapes` (default 3) stay alive, the least recently used one is closed at an idle point
        self._graphs = {}
        self._graph_lr = {}
        # `training.teacher_prefetch: true` (graph mode only; what bench.py measures): train() looks one batch ahead and the frozen
        # teacher's pass over batch t+1 runs beside the backward / AdamW of step t (GraphedBilevel.main(next_batch=)).  The random
        # draws keep the eager order (main t, upper t, main t+1, ...): the loss curve is the eager trainer's.
        self.teacher_prefetch = self.hip_graphs and bool(_cfg(config, "training.teacher_prefetch", _cfg(config, "teacher_prefetch", False)))
        self._rng_snapshot = None            # generator state a checkpoint must store while a look-ahead draw is outstanding
        if self.hip_graphs:
            pert = float(_cfg(config, "model.prediction_model.input_perturbation", 0.0) or 0.0)
            if self.max_grad_norm is not None or pert:
                raise ValueError("training.hip_graphs does not support clip_grad_norm / input_perturbation (use eager mode)")

    # ---- frozen VAE (trainer.py:2128-2131, cast to the weight dtype :516-527); built on first use
    @property
    def vae_factor(self):
        return 4 if _cfg(self.config, "tiny", False) else 8

    def _allow_random(self, what, path):
        """Random weights / a random arch vector stand in for missing files only when asked for (`--synthetic`, or
        `model.prediction_model.random_init` like the reference's random_init path, unet_2d_conditional.py:2406-2408).
        Otherwise a missing checkpoint is an error, as in the reference (every from_pretrained there raises)."""
        if bool(_cfg(self.config, "synthetic", False)) or bool(_cfg(self.config, "model.prediction_model.random_init", False)):
            return True
        raise FileNotFoundError(
            f"{what}: {path!r} is not a local directory / file (hub downloads are not available to this build); pass a local "
            f"snapshot laid out like the hub's, or run with --synthetic / model.prediction_model.random_init for random weights")

    @property
    def vae(self):
        if getattr(self, "_vae", None) is None:
            from ..models.vae.autoencoder_kl import AutoencoderKL, VAEConfig
            root = _cfg(self.config, "pretrained_model_name_or_path")
            local = bool(root) and os.path.isdir(os.path.join(root, "vae"))
            if not local:
                self._allow_random("VAE", os.path.join(str(root), "vae"))
            vcfg = VAEConfig(block_out_channels=(32, 64, 64), layers_per_block=1) if _cfg(self.config, "tiny", False) else None
            self._vae = AutoencoderKL.from_pretrained(root if local else None, subfolder="vae", random_init=not local,
                                                      vae_config=vcfg, torch_dtype=self.weight_dtype, device=self.device)
        return self._vae

    # ---- frozen CLIP text encoder (trainer.py:2126-2131); built on first use
    @property
    def text_encoder(self):
        if getattr(self, "_text_encoder", None) is None:
            from ..models.clip.text_encoder import CLIPTextModel, CLIPTextConfig
            root = _cfg(self.config, "pretrained_model_name_or_path")
            local = bool(root) and os.path.isdir(os.path.join(root, "text_encoder"))
            if not local:
                self._allow_random("text encoder", os.path.join(str(root), "text_encoder"))
            tcfg = None
            if _cfg(self.config, "tiny", False):
                tcfg = CLIPTextConfig(vocab_size=1000, hidden_size=self.unet_config.cross_attention_dim, intermediate_size=256,
                                      num_hidden_layers=2, num_attention_heads=self.unet_config.cross_attention_dim // 64)
            self._text_encoder = CLIPTextModel.from_pretrained(root if local else None, subfolder="text_encoder",
                                                               random_init=not local, text_config=tcfg,
                                                               torch_dtype=self.weight_dtype, device=self.device)
            self._empty_cache = {}
        return self._text_encoder

    @staticmethod
    def _is_empty(batch):
        """collate_fn yields zero-element tensors when every sample of a batch failed to load (data_utils.py:286-312);
        the reference tests `pixel_values.numel() == 0`."""
        for key in ("pixel_values", "latents", "prompt_embeds", "input_ids"):
            if key in batch:
                return batch[key].numel() == 0
        return True

    def _prompt_embeds(self, batch, empty=False):
        """`prompt_embeds` / `empty_prompt_embeds` as given, else encoded from `input_ids` / `empty_input_ids`."""
        key, ids_key = ("empty_prompt_embeds", "empty_input_ids") if empty else ("prompt_embeds", "input_ids")
        if key in batch:
            return batch[key]
        if ids_key not in batch:
            raise KeyError(f"batch has neither {key!r} nor {ids_key!r}")
        ids = batch[ids_key]
        enc = self.text_encoder
        if not empty:
            return enc(ids)[0]
        # the empty prompt is one token row repeated over the batch (data_utils.py:272-274): encode it once, ever
        row = tuple(ids[0].tolist())
        if not bool((ids == ids[:1]).all()):
            return enc(ids)[0]
        if row not in self._empty_cache:
            self._empty_cache[row] = enc(ids[:1])[0]
        return self._empty_cache[row].expand(ids.shape[0], -1, -1).contiguous()

    # ---- trainer.py:516-527: student master weights fp32; bf16 compute under mixed precision
    def init_weight_dtype(self):
        mp = _cfg(self.config, "mixed_precision", None) or _cfg(self.config, "training.mixed_precision", None)
        self.weight_dtype = {"bf16": torch.bfloat16, "no": torch.float32, None: torch.float32}.get(mp)
        if self.weight_dtype is None:
            raise ValueError(f"mixed_precision={mp!r} is not supported on this build (use bf16 or no)")

    # ---- trainer.py:2122-2198
    def init_models(self):
        c = self.config
        ucfg = UNetConfig.tiny() if _cfg(c, "tiny", False) else UNetConfig.sd21()
        self.unet_config = ucfg
        ckpt = _cfg(c, "pruning_ckpt_dir")
        arch = None
        if ckpt and os.path.exists(os.path.join(ckpt, "quantizer_embeddings.pt")):
            emb = torch.load(os.path.join(ckpt, "quantizer_embeddings.pt"), map_location="cpu")
            arch = emb[int(_cfg(c, "expert_id", 0)) % emb.shape[0]][None]         # trainer.py:2159-2161
        elif ckpt and os.path.exists(os.path.join(ckpt, "arch_vector.pt")):
            arch = torch.load(os.path.join(ckpt, "arch_vector.pt"), map_location="cpu")
        else:
            self._allow_random("pruning checkpoint (quantizer_embeddings.pt / arch_vector.pt)", ckpt)
            res = int(_cfg(c, "model.prediction_model.resolution", 512)) // 8
            arch, ratio, _ = arch_vector_for_budget(ucfg, float(_cfg(c, "keep_ratio", 0.55)), hw=res)
            logger.info("no pruning checkpoint: random arch vector at MAC budget %.3f", ratio)
        self.arch_vector = arch
        pm = _cfg(c, "model.prediction_model", {})
        root = _cfg(c, "pretrained_model_name_or_path")
        local = bool(root) and os.path.isdir(os.path.join(root, "unet"))
        if not local:
            self._allow_random("U-Net", os.path.join(str(root), "unet"))
        kw = dict(unet_config=ucfg, torch_dtype=self.weight_dtype, device=self.device,
                  down_block_types=pm.get("unet_down_blocks"), up_block_types=pm.get("unet_up_blocks"),
                  mid_block_type=pm.get("unet_mid_block"),
                  attention_precision=pm.get("attention_precision", _cfg(c, "attention_precision")),
                  gated_ff=pm.get("gated_ff", True), ff_gate_width=pm.get("ff_gate_width", 32))
        self.teacher_model = UNet2DConditionModelPruned.from_pretrained(root if local else None, subfolder="unet",
                                                                        arch_vector=None, random_init=not local,
                                                                        train=False, seed=0, **kw)
        self.prediction_model = UNet2DConditionModelPruned.from_pretrained(
            root if local else None, subfolder="unet", arch_vector=arch,
            random_init=bool(pm.get("random_init", False)) or not local, train=True, seed=0, **kw)
        if not local:      # same dense initialisation sliced by the arch vector, like load-then-prune
            self.prediction_model.load_dense_or_pruned(self.teacher_model.state_dict())
        from ..models.unet.spec import plan_macs
        res = int(_cfg(c, "model.prediction_model.resolution", 512)) // 8
        tm = plan_macs(ucfg, self.teacher_model.blocks, res, 77)[0]
        sm = plan_macs(ucfg, self.prediction_model.blocks, res, 77)[0]
        logger.info("Teacher MACs %.1f G, student MACs %.1f G, Pruning Ratio %.3f", tm / 1e9, sm / 1e9, sm / tm)

    # ---- trainer.py:2233-2250, 2676-2717, 436-443, 2666-2674
    def init_optimizer(self):
        c = self.config
        o = _cfg(c, "training.optim", {})
        lss = _cfg(c, "training.losses", {})
        lr = float(o.get("prediction_model_learning_rate", 1e-6))
        ulr = float(o.get("prediction_model_upper_learning_rate", lr))
        bs = int(_cfg(c, "data.dataloader.train_batch_size", 8))
        if o.get("scale_lr", False):
            s = (int(_cfg(c, "training.gradient_accumulation_steps", 1)) * bs * self.world) ** 0.5
            lr, ulr = lr * s, ulr * s
        g = lambda name, key, d: float((lss.get(name) or {}).get(key, d) or 0.0)
        self.stepper = BilevelStepper(
            self.prediction_model, self.teacher_model,
            w_diff=g("diffusion_loss", "weight", 1.0), w_dist=g("distillation_loss", "weight", 0.0),
            w_block=g("block_loss", "weight", 0.0), snr_gamma=(lss.get("diffusion_loss") or {}).get("snr_gamma", 5.0),
            up_w_dist=g("distillation_loss", "upper_weight", 1.0), up_w_block=g("block_loss", "upper_weight", 0.0),
            prediction_type=_cfg(c, "model.prediction_model.prediction_type", "v_prediction"),
            lr=lr, upper_lr=ulr, betas=(float(o.get("adam_beta1", 0.9)), float(o.get("adam_beta2", 0.999))),
            eps=float(o.get("adam_epsilon", 1e-8)), weight_decay=float(o.get("prediction_model_weight_decay", 0.0)),
            warmup_steps=int(o.get("lr_warmup_steps", 0)),
            upper_warmup_steps=int(o.get("upper_lr_warmup_steps", o.get("lr_warmup_steps", 0))), bilevel=self.bilevel)
        self.max_grad_norm = float(o["max_grad_norm"]) if o.get("clip_grad_norm") else None
        sched = o.get("lr_scheduler", "constant_with_warmup")
        if sched not in ("constant", "constant_with_warmup"):      # diffusers get_scheduler names (trainer.py:436-443)
            raise ValueError(f"training.optim.lr_scheduler={sched!r} is not supported by this build (every shipped config "
                             f"uses constant_with_warmup; 'constant' is the same with lr_warmup_steps 0)")
        if sched == "constant":
            self.stepper.opt.warmup = 0
            if self.stepper.upper_opt is not None:
                self.stepper.upper_opt.warmup = 0

    def init_dataloader(self, upper):
        c = self.config
        if not _cfg(c, "synthetic", False):
            raise NotImplementedError(
                "image datasets and tokenisation are host-side data loading (not built): pass a dataloader yielding "
                "{'pixel_values' | 'latents', 'prompt_embeds' | 'input_ids', 'empty_prompt_embeds' | 'empty_input_ids'} "
                "or run with --synthetic")
        bs = int(_cfg(c, "data.dataloader.train_batch_size", 8))
        res = int(_cfg(c, "model.prediction_model.resolution", 512)) // 8
        seed = int(_cfg(c, "seed", 43)) + self.rank + (7919 if upper else 0)
        T = 13 if _cfg(c, "tiny", False) else 77
        return SyntheticBatches(bs, res, T, self.unet_config.cross_attention_dim, seed, self.device,
                                pixels=bool(_cfg(c, "synthetic_pixels", False)), vae_factor=self.vae_factor)

    # ---- sampling prologue shared by step/upper_step (trainer.py:2405-2423)
    def _sample(self, batch):
        if "latents" in batch:
            lat = batch["latents"].to(self.device, torch.float32)
        elif "pixel_values" in batch:      # trainer.py:2405-2406; the Gaussian draw comes first, as in the reference
            lat = self.vae.encode_latents(batch["pixel_values"], generator=self.rng)
        else:
            raise KeyError("batch has neither 'pixel_values' nor 'latents'")
        noise = torch.randn(lat.shape, device=self.device, generator=self.rng)
        off = float(_cfg(self.config, "model.prediction_model.noise_offset", 0.0) or 0.0)
        if off:
            noise = noise + off * torch.randn((lat.shape[0], lat.shape[1], 1, 1), device=self.device, generator=self.rng)
        # trainer.py:2416-2417, 2427-2428: the forward process uses the perturbed noise, the target the clean one
        pert = float(_cfg(self.config, "model.prediction_model.input_perturbation", 0.0) or 0.0)
        self._input_noise = noise + pert * torch.randn(noise.shape, device=self.device, generator=self.rng) if pert else None
        mx = int(_cfg(self.config, "model.prediction_model.max_scheduler_steps", 1000) or 1000)
        t = torch.randint(0, mx, (lat.shape[0],), device=self.device, generator=self.rng).long()
        return lat, noise, t

    def _graphed(self, lat, ehs):
        """The captured step for this batch shape (built on first use; weights and optimiser state are untouched by the
        capture's warm-up - GraphedBilevel snapshots and restores them)."""
        from .bilevel import GraphedBilevel
        if self.max_grad_norm is not None or self._input_noise is not None:
            raise ValueError("training.hip_graphs does not support clip_grad_norm / input_perturbation (use eager mode)")
        key = (tuple(lat.shape), tuple(ehs.shape))
        g = self._graphs.pop(key, None)
        if g is None:
            keep = max(1, int(_cfg(self.config, "training.hip_graph_shapes", 3)))
            while len(self._graphs) >= keep:                 # evict the least recently used shape: ordered teardown, device idle
                self._graphs.pop(next(iter(self._graphs))).close()
            B, C, H, W = lat.shape
            g = GraphedBilevel(self.stepper, B, C, H, W, ehs.shape[1], ehs.shape[2], prefetch=self.teacher_prefetch)
            g.capture(bilevel=self.bilevel)
        self._graphs[key] = g                                # most recently used last
        return g

    def _prepare(self, batch, upper=False):
        """The host-visible prologue of step() / upper_step(): every random draw of the step and its prompt embeddings."""
        lat, noise, t = self._sample(batch)
        if upper and self._input_noise is not None:      # trainer.py:2917-2932: the upper step diffuses with the perturbed noise
            noise = self._input_noise
        ehs = self._prompt_embeds(batch)
        return (lat, noise, t, ehs, self._prompt_embeds(batch, empty=True)) if upper else (lat, noise, t, ehs)

    @staticmethod
    def _same_graphs(a, lat, ehs):
        """Another batch shape has its own captured graphs: no teacher hand-over between them."""
        return a is not None and a[0].shape == lat.shape and a[3].shape == ehs.shape

    def step(self, batch, backward=True, prepared=None, ahead=None, ids=(None, None), upper_ahead=None, upper_id=None):
        """prepared: `_prepare(batch)` done earlier (look-ahead); ahead: the prepared NEXT main batch, announced to the graphs
        under the token ids[1] - the call that then passes it as `prepared` with ids[0] == that token finds its teacher pass done;
        upper_ahead / upper_id: the same for the upper step that follows this main step."""
        lat, noise, t, ehs = prepared if prepared is not None else self._prepare(batch)
        if backward and self.hip_graphs:          # graph replay: loss heads + backward + all-reduce + AdamW in one go
            self._graph_lr["main"] = self._graphed(lat, ehs).main(
                lat, noise, t, ehs, batch_id=ids[0], next_batch=ahead if self._same_graphs(ahead, lat, ehs) else None, next_id=ids[1],
                next_upper=upper_ahead if self._same_graphs(upper_ahead, lat, ehs) else None, upper_id=upper_id)
            return self._tuple(self.stepper.losses.clone(), upper=False)
        L = self.stepper.main_step(lat, noise, t, ehs, backward=backward, input_noise=self._input_noise)
        return self._tuple(L, upper=False)

    # ---- trainer.py:2490-2541
    @torch.no_grad()
    def validate(self, eval_dataloader=None):
        """`step()` without a backward over the evaluation batches, the four means reduced over the ranks (C4 of SURVEY
        2.4: four scalar all-reduces) and logged under the reference's `validation/*` keys."""
        loader = eval_dataloader if eval_dataloader is not None else getattr(self, "eval_dataloader", None)
        if loader is None:
            return None
        tot = torch.zeros(4, device=self.device, dtype=torch.float64)
        n = 0
        for batch in loader:
            n += 1                                            # the reference divides by len(eval_dataloader), skipped ones too
            if self._is_empty(batch):
                continue
            tot += torch.stack(self.step(batch, backward=False)).double()
        tot /= max(n, 1)
        if self.world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            tot /= self.world
        vals = tot.tolist()
        rec = {"step": self.global_step, "validation/loss": vals[0], "validation/diffusion_loss": vals[1],
               "validation/distillation_loss": vals[2], "validation/block_loss": vals[3]}
        self._log(rec)
        return rec

    def _tuple(self, L, upper):
        w = self.stepper.w
        d, s, b = L[0], L[1], L[2]
        tot = (w["up_dist"] * s + w["up_block"] * b) if upper else (w["diff"] * d + w["block"] * b + w["dist"] * s)
        return tot.float(), (torch.zeros_like(d) if upper else d).float(), s.float(), b.float()

    # ---- image logging (trainer.py:2543-2575 generate_samples_from_prompts; called every image_logging_steps, :2851-2859)
    def get_pipeline(self):
        from ..pipelines.pruning_pipelines import PNDMScheduler, StableDiffusionPruningPipeline
        pt = _cfg(self.config, "model.prediction_model.prediction_type", "v_prediction")
        return StableDiffusionPruningPipeline(self.vae, self.text_encoder, self.prediction_model,
                                              PNDMScheduler(prediction_type=pt))

    def generate_samples_from_prompts(self):
        """Samples every prompt batch with the current student (PNDM, `training.num_inference_steps`, guidance 7.5, the
        configured seed) and writes `<logging_dir>/images/step-<n>.npy` ([N, H, W, 3] uint8) instead of a wandb image grid."""
        if self.prompt_dataloader is None:
            return None
        pipe = self.get_pipeline()
        steps = int(_cfg(self.config, "training.num_inference_steps", 50))
        seed = _cfg(self.config, "seed", None)
        res = int(_cfg(self.config, "model.prediction_model.resolution", 512)) // 8 * self.vae_factor
        images = []
        for batch in self.prompt_dataloader:
            gen = None if seed is None else torch.Generator(device=self.device).manual_seed(int(seed))
            kw = {}
            for src, dst in (("prompt_embeds", "prompt_embeds"), ("empty_prompt_embeds", "negative_prompt_embeds"),
                             ("input_ids", "prompt_ids"), ("empty_input_ids", "negative_prompt_ids")):
                if src in batch:
                    kw[dst] = batch[src]
            images.append(pipe.generate_samples(num_inference_steps=steps, generator=gen, output_type="pt", height=res,
                                                width=res, **kw).images)
        images = torch.cat(images)
        if self.rank == 0:
            d = os.path.join(self.logging_dir, "images")
            os.makedirs(d, exist_ok=True)
            import numpy as np
            np.save(os.path.join(d, f"step-{self.global_step}.npy"),
                    (images.permute(0, 2, 3, 1) * 255).round().to(torch.uint8).cpu().numpy())
        return images

    # ---- checkpointing (trainer.py:452-514, 2863-2869): the directory accelerator.save_state leaves behind
    def save_checkpoint(self):
        """<logging_dir>/checkpoint-<step>/ = unet/ (diffusers safetensors, pruned shapes: the reference's save hook,
        trainer.py:314-333), arch_vector.pt (:2867-2869), optimizer.bin + scheduler.bin (main AdamW / LambdaLR),
        optimizer_1.bin + scheduler_1.bin (upper pair), random_states_<rank>.pkl - each in the layout torch / accelerate
        write (torch.optim.AdamW.state_dict(), LambdaLR.state_dict(), accelerate's RNG dict) so that the reference's
        accelerator.load_state can read what this build writes and vice versa."""
        d = os.path.join(self.logging_dir, f"checkpoint-{self.global_step}")
        os.makedirs(d, exist_ok=True)
        import random
        import numpy as np
        rng = {"step": self.global_step, "random_state": random.getstate(), "numpy_random_seed": np.random.get_state(),
               "torch_manual_seed": torch.get_rng_state(), "torch_cuda_manual_seed": torch.cuda.get_rng_state_all(),
               # this build draws from its own device generator (with a look-ahead batch drawn: its state BEFORE that draw)
               "pdm_generator_state": (self._rng_snapshot if self._rng_snapshot is not None else self.rng.get_state()).cpu()}
        with open(os.path.join(d, f"random_states_{self.rank}.pkl"), "wb") as f:       # every rank writes its own
            pickle.dump(rng, f)
        if self.rank != 0:
            return
        limit = _cfg(self.config, "training.logging.checkpoints_total_limit")
        if limit:                                   # rotate BEFORE saving: at most limit-1 older ones stay (trainer.py:454-473)
            cks = sorted((p for p in glob.glob(os.path.join(self.logging_dir, "checkpoint-*")) if p != d),
                         key=lambda p: int(p.split("-")[-1]))
            import shutil
            for old in cks[:max(0, len(cks) - int(limit) + 1)]:
                shutil.rmtree(old, ignore_errors=True)
        self.prediction_model.save_pretrained(os.path.join(d, "unet"))
        torch.save(self.arch_vector, os.path.join(d, "arch_vector.pt"))
        pairs = [("", self.stepper.opt)] + ([("_1", self.stepper.upper_opt)] if self.stepper.upper_opt is not None else [])
        for suffix, opt in pairs:
            torch.save(opt.state_dict(), os.path.join(d, f"optimizer{suffix}.bin"))
            torch.save(opt.scheduler_state_dict(), os.path.join(d, f"scheduler{suffix}.bin"))
        logger.info("Saved state to %s", d)

    def load_checkpoint(self):
        r = _cfg(self.config, "training.logging.resume_from_checkpoint")
        if not r:
            return
        if r == "latest":
            cks = sorted(glob.glob(os.path.join(self.logging_dir, "checkpoint-*")), key=lambda p: int(p.split("-")[-1]))
            r = cks[-1] if cks else None
        if not r or not os.path.isdir(r):
            logger.info("Checkpoint %r does not exist. Starting a new training run.", r)
            return
        self.prediction_model.load_pretrained_dir(os.path.join(r, "unet"))
        pairs = [("", self.stepper.opt)] + ([("_1", self.stepper.upper_opt)] if self.stepper.upper_opt is not None else [])
        for suffix, opt in pairs:
            f = os.path.join(r, f"optimizer{suffix}.bin")
            if os.path.exists(f):
                opt.load_state_dict(torch.load(f, map_location="cpu", weights_only=False))
            f = os.path.join(r, f"scheduler{suffix}.bin")
            if os.path.exists(f):
                opt.load_scheduler_state_dict(torch.load(f, map_location="cpu", weights_only=False))
        f = os.path.join(r, f"random_states_{self.rank}.pkl")
        if os.path.exists(f):
            with open(f, "rb") as fh:
                st = pickle.load(fh)
            if "pdm_generator_state" in st:          # continue the noise / timestep stream where it stopped
                self.rng.set_state(st["pdm_generator_state"])
        self.global_step = int(os.path.basename(r.rstrip("/")).split("-")[1])        # trainer.py:506
        logger.info("Resumed from %s at global step %d", r, self.global_step)

    def _log(self, rec):
        if self.rank == 0:
            os.makedirs(self.logging_dir, exist_ok=True)
            with open(os.path.join(self.logging_dir, "metrics.jsonl"), "a") as f:
                f.write(json.dumps(rec) + "\n")


class UnetFineTuner(Trainer):
    """Single-level fine-tune (trainer.py:2116-2574)."""

    def train(self):
        c = self.config
        ck_every = int(_cfg(c, "training.logging.checkpoint_steps", _cfg(c, "training.checkpoint_steps", 10000)))
        val_every = int(_cfg(c, "training.validation_steps", 0) or 0)
        freq = int(_cfg(c, "training.upper_step_freq", 10))
        # training.gradient_accumulation_steps = k, AS THE REFERENCE'S FINE-TUNE LOOPS TREAT IT (trainer.py:2293-2340, 2769-2800): they
        # never enter `accelerator.accumulate(...)`, so `sync_gradients` stays True and the optimiser steps on EVERY batch; what k
        # changes is (a) `accelerator.backward(loss)` divides the loss - i.e. every gradient, of the upper step too - by k,
        # (b) the logged `finetuning/loss` is loss / k (:2778-2779), (c) an epoch counts ceil(len(dataloader) / k) update steps in
        # update_config_params / update_train_steps / load_checkpoint (:445-450, 529-537, 508), (d) scale_lr (init_optimizer).
        accum = max(1, int(_cfg(c, "training.gradient_accumulation_steps", 1) or 1))
        self.stepper.accum = accum
        # trainer.py:445-450 update_config_params / update_train_steps: max_train_steps, or num_train_epochs full passes
        max_steps = _cfg(c, "training.max_train_steps")
        try:
            per_epoch = -(-len(self.train_dataloader) // accum)          # num_update_steps_per_epoch
        except TypeError:
            per_epoch = None
        if max_steps is None:
            if per_epoch is None:
                raise ValueError("training.max_train_steps is unset and the dataloader has no length")
            max_steps = int(_cfg(c, "training.num_train_epochs", 1)) * per_epoch
        max_steps = int(max_steps)
        # update_train_steps (trainer.py:529-537): the epoch count is RE-derived from max_train_steps, so the loop below ends
        # after ceil(max_train_steps / steps per epoch) passes over the dataloader - also when skipped (empty) batches left
        # it short of max_train_steps.  A dataloader without a length has no epoch bound.
        epochs = -(-max_steps // per_epoch) if per_epoch else None
        self.load_checkpoint()
        # trainer.py:2744-2767: a resumed run starts at first_epoch = global_step // steps per epoch
        first_epoch = self.global_step // per_epoch if (per_epoch and self.global_step) else 0
        upper_iter = iter(self.upper_dataloader) if self.bilevel else None

        def next_upper():            # sample a batch from the upper dataset, restarting it when it runs out (trainer.py:2798-2803)
            nonlocal upper_iter
            try:
                return next(upper_iter)
            except StopIteration:
                upper_iter = iter(self.upper_dataloader)
                return next(upper_iter)
        pending = None
        t0 = time.time()
        epoch = first_epoch
        log_every = int(_cfg(c, "training.image_logging_steps", 0) or 0)
        ahead = None                # (prepared next main batch, its token) - teacher_prefetch only
        # for epoch in range(first_epoch, num_train_epochs) (trainer.py:2769)
        while self.global_step < max_steps and (epochs is None or epoch < epochs):
            stepped = False
            it = iter(self.train_dataloader)
            nxt = next(it, None)
            while nxt is not None:
                batch, nxt = nxt, next(it, None)
                if self.global_step >= max_steps:
                    break
                if self._is_empty(batch):                        # empty batch is skipped (trainer.py:2771-2772)
                    continue
                stepped = True
                upper_due = self.bilevel and (self.global_step + 1) % freq == 0
                ub = up_prepared = None
                if self.teacher_prefetch:
                    # every draw in the eager order - this batch (drawn one step ago, or now), then the upper batch that follows
                    # it, then the look-ahead - before anything is launched
                    gs = self.global_step
                    prepared = ahead[0] if (ahead is not None and ahead[1] == gs) else self._prepare(batch)
                    ahead, self._rng_snapshot = None, None
                    if upper_due:
                        ub = next_upper()
                        up_prepared = self._prepare(ub, upper=True)
                    # no look-ahead into a step that will not run, across an epoch boundary, or past a point where something else
                    # draws from the generator / stores its state between the steps (validation, image logging)
                    quiet = not (val_every and gs % val_every == 0) and not (log_every and self.prompt_dataloader is not None
                                                                             and gs % log_every == 0)
                    if nxt is not None and not self._is_empty(nxt) and gs + 1 < max_steps and quiet:
                        self._rng_snapshot = self.rng.get_state()
                        ahead = (self._prepare(nxt), gs + 1)
                    # each teacher pass gets the step in front of it as its window: the upper step's is announced to this main step,
                    # the next main batch to the step that runs last in this iteration
                    main_next = dict(ahead=ahead[0], next_id=gs + 1) if ahead else {}
                    if upper_due:
                        loss = self.step(batch, prepared=prepared, ids=(gs, None), upper_ahead=up_prepared, upper_id=("upper", gs))
                    else:
                        loss = self.step(batch, prepared=prepared, ahead=main_next.get("ahead"), ids=(gs, main_next.get("next_id")))
                else:
                    loss = self.step(batch)
                lr = (self._graph_lr["main"] if self.hip_graphs else
                      self.stepper.optimizer_step(upper=False, max_grad_norm=self.max_grad_norm))
                rec = {"step": self.global_step, "finetuning/prediction_model_lr": lr}
                keys = ("finetuning/loss", "finetuning/diffusion_loss", "finetuning/distillation_loss", "finetuning/block_loss")
                vals = [torch.stack(loss)]
                if accum != 1:
                    vals[0] = vals[0] * torch.tensor([1.0 / accum, 1.0, 1.0, 1.0], device=vals[0].device, dtype=vals[0].dtype)
                if upper_due:          # trainer.py:2795-2816
                    if ub is None:
                        ub = next_upper()
                    up = (self.upper_step(ub, prepared=up_prepared, batch_id=("upper", self.global_step), **main_next)
                          if self.teacher_prefetch else self.upper_step(ub))
                    rec["finetuning/upper_prediction_model_lr"] = (
                        self._graph_lr["upper"] if self.hip_graphs else
                        self.stepper.optimizer_step(upper=True, max_grad_norm=self.max_grad_norm))
                    keys += ("finetuning/upper_loss", "finetuning/upper_diffusion_loss",
                             "finetuning/upper_distillation_loss", "finetuning/upper_block_loss")
                    vals.append(torch.stack(up))
                host = torch.cat(vals).to("cpu", non_blocking=True)       # read back asynchronously, log one step late
                if pending is not None:
                    self._flush(*pending)
                pending = (rec, keys, host, torch.cuda.Event())
                pending[3].record()
                if val_every and self.global_step % val_every == 0:        # trainer.py:2848-2850
                    self.validate()
                if log_every and self.prompt_dataloader is not None and self.global_step % log_every == 0:
                    self.generate_samples_from_prompts()
                self.global_step += 1
                if self.global_step % ck_every == 0:
                    if self.world > 1:
                        dist.barrier()
                    self.save_checkpoint()
            epoch += 1
            if not stepped:
                raise RuntimeError("the training dataloader yielded no usable batch in a whole epoch")
        if pending is not None:
            self._flush(*pending)
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        self.save_checkpoint()
        logger.info("finished %d steps (%d epochs) in %.1fs", self.global_step, epoch, time.time() - t0)

    def _flush(self, rec, keys, host, ev):
        ev.synchronize()
        rec.update({k_: float(v) for k_, v in zip(keys, host.tolist())})
        self._log(rec)


class BilevelUnetFineTuner(UnetFineTuner):
    """Bilevel fine-tune + concept suppression (trainer.py:2577-3001)."""
    bilevel = True

    def upper_step(self, batch, prepared=None, batch_id=None, ahead=None, next_id=None):
        lat, noise, t, ehs, empty = prepared if prepared is not None else self._prepare(batch, upper=True)
        if self.hip_graphs:
            self._graph_lr["upper"] = self._graphed(lat, ehs).upper(
                lat, noise, t, ehs, empty, batch_id=batch_id, next_batch=ahead if self._same_graphs(ahead, lat, ehs) else None,
                next_id=next_id)
            return self._tuple(self.stepper.losses.clone(), upper=True)
        L = self.stepper.upper_step(lat, noise, t, ehs, empty)
        return self._tuple(L, upper=True)


    def init_upper_dataset(self, dataset, preprocess_train=None):
        """trainer.py:2634-2650: the upper (concept) dataset is the rows whose `style` column is in `upper_data.style`.
        Dataset loading itself is host-side I/O outside this build: `dataset` is anything with
        `column_names`, `filter` and `with_transform` (a `datasets.Dataset`), handed in by the caller."""
        caption_column = _cfg(self.config, "upper_data.caption_column", "caption")
        if caption_column not in dataset.column_names:
            raise ValueError(f"--caption_column '{caption_column}' needs to be one of: {', '.join(dataset.column_names)}")
        style = _cfg(self.config, "upper_data.style")
        if style is not None:
            dataset = dataset.filter(lambda s: s["style"] in style)        # `in`, as the reference: a string or a list of styles
        return dataset.with_transform(preprocess_train) if preprocess_train is not None else dataset


class NudityBilevelUnetFineTuner(BilevelUnetFineTuner):
    """Same as BilevelUnetFineTuner; only the upper dataset selection differs (trainer.py:3004-3016): no `style` filter."""

    def init_upper_dataset(self, dataset, preprocess_train=None):
        caption_column = _cfg(self.config, "upper_data.caption_column", "caption")
        if caption_column not in dataset.column_names:
            raise ValueError(f"--caption_column '{caption_column}' needs to be one of: {', '.join(dataset.column_names)}")
        return dataset.with_transform(preprocess_train) if preprocess_train is not None else dataset
