"""CLI surface of the fine-tuning entry points: same flag names / defaults as pdm/utils/arg_utils.py:5-133 so that
slurm_scripts/coco/*.slurm drive this build unchanged, plus MI355X-specific additions at the end."""
import argparse
import os


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Fine-tune / unlearn a pruned SD-2.1 U-Net (MI355X-native build).")
    p.add_argument("--base_config_path", type=str, required=False, default=None)
    p.add_argument("--pretrained_model_name_or_path", type=str, default="stabilityai/stable-diffusion-2-1")
    p.add_argument("--prompt_encoder_model_name_or_path", type=str, default="sentence-transformers/all-mpnet-base-v2")
    p.add_argument("--revision", type=str, default=None)
    p.add_argument("--non_ema_revision", type=str, default=None)
    p.add_argument("--cache_dir", type=str, default=None)
    p.add_argument("--seed", type=int, default=43)
    p.add_argument("--logging_dir", type=str, default="logs")
    p.add_argument("--mixed_precision", type=str, default=None, choices=["no", "fp16", "bf16"])
    p.add_argument("--local_rank", type=int, default=-1)
    p.add_argument("--enable_xformers_memory_efficient_attention", action="store_true")
    p.add_argument("--push_to_hub", action="store_true")
    p.add_argument("--hub_token", type=str, default=None)
    p.add_argument("--hub_model_id", type=str, default=None)
    p.add_argument("--tracker_project_name", type=str, default="text2image-fine-tune")
    p.add_argument("--wandb_run_name", type=str, default=None)
    p.add_argument("--pruning_ckpt_dir", type=str, default=None)
    p.add_argument("--finetuning_ckpt_dir", type=str, default=None)
    p.add_argument("--expert_id", type=int, default=None)
    p.add_argument("--base_arch", action="store_true")
    # accepted for command-line compatibility (pdm/utils/arg_utils.py:15-20, 51-54, 101-105, 120-121); the fine-tune /
    # unlearn path reads none of them (CLIP scoring, EMA, pruning-ratio script and erasure baselines are out of scope)
    p.add_argument("--clip_model_name_or_path", type=str, default="laion/CLIP-ViT-H-14-laion2B-s32B-b79K")
    p.add_argument("--use_ema", action="store_true")
    p.add_argument("--pruning_type", type=str, default="multi-expert", choices=["multi-expert", "single-expert"])
    p.add_argument("--erasure_ckpt_path", type=str, default=None)
    # --- MI355X build additions
    p.add_argument("--synthetic", action="store_true", help="seeded synthetic (latent, timestep, prompt-embed) batches")
    p.add_argument("--keep_ratio", type=float, default=0.55, help="MAC budget of the random arch vector in synthetic mode")
    p.add_argument("--tiny", action="store_true", help="tiny U-Net topology (tests / smoke)")
    p.add_argument("--hip_graphs", action="store_true", help="replay the training step as captured hipGraphs (fixed batch shapes)")
    p.add_argument("--teacher_prefetch", action="store_true",
                   help="with --hip_graphs: look one batch ahead, the frozen teacher's pass over it runs beside this step's backward")
    args = p.parse_args(argv)
    env_local_rank = int(os.environ.get("LOCAL_RANK", -1))
    if env_local_rank != -1 and env_local_rank != args.local_rank:
        args.local_rank = env_local_rank
    if args.non_ema_revision is None:
        args.non_ema_revision = args.revision
    return args
