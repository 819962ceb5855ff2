"""U-Net topology literal.  Follows pdm/models/unet/unet_2d_conditional.py:738-776, 1004-1169 with the SD-2.1
config (block_out_channels (320,640,1280,1280), heads (5,10,20,20), cross_attention_dim 1024, linear projection,
layers_per_block 2; cross-ref baselines/erasing/oldcode_erasing_compvis/train-scripts/convertModels.py:239-259)."""
from dataclasses import dataclass
from typing import Tuple


@dataclass(frozen=True)
class UNetConfig:
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    heads: Tuple[int, ...] = (5, 10, 20, 20)
    cross_attention_dim: int = 1024
    layers_per_block: int = 2
    norm_num_groups: int = 32
    in_channels: int = 4
    out_channels: int = 4
    ff_gate_width: int = 32
    # which stages carry transformers (CrossAttn* block types, configs/baselines/*bilevel.yaml:11-26)
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    up_has_attn: Tuple[bool, ...] = (False, True, True, True)

    @property
    def temb_dim(self):
        return 4 * self.block_out_channels[0]

    @staticmethod
    def sd21():
        return UNetConfig()

    @staticmethod
    def tiny():
        # head dim stays 64 (= C/heads, blocks.py:1634); group sizes 2/4/8 exercise channel padding
        return UNetConfig(block_out_channels=(64, 128, 256, 256), heads=(1, 2, 4, 4), cross_attention_dim=64)
