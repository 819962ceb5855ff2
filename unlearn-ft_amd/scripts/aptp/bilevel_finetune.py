#!/usr/bin/env python3
"""Entry point with the reference's name and flags (scripts/aptp/bilevel_finetune.py:19-41): launch as
    python -m torch.distributed.run --nproc-per-node N scripts/aptp/bilevel_finetune.py --base_config_path CFG \\
        --pruning_ckpt_dir DIR --expert_id K [--mixed_precision bf16] [--synthetic]
(accelerate launch works too: it only sets RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch
import torch.distributed as dist

from pdm.training.trainer import BilevelUnetFineTuner
from pdm.utils.arg_utils import parse_args
from pdm.utils.config import load_config


def main():
    args = parse_args()
    config = load_config(args.base_config_path)
    config.update(vars(args))                       # flat CLI overlay at the root, like the reference
    if not args.synthetic:
        assert config.pruning_ckpt_dir is not None, "Please provide a path to the pruning checkpoint directory."
        assert config.expert_id is not None, "Please provide an expert ID."
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(levelname)s %(message)s")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl")
    torch.manual_seed(args.seed)
    BilevelUnetFineTuner(config).train()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
