"""UNet architecture descriptions (diffusers `unet/config.json` fields the hot path needs).

The reference loads these through diffusers (trainscripts/textsliders/model_util.py:67-72, 169-174);
here they are plain dataclasses so the planner has no third-party dependency.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Optional, Tuple


@dataclass
class UNetConfig:
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = (
        "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = (
        "UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 1, 1, 1)
    attention_head_dim: Tuple[int, ...] = (8, 8, 8, 8)   # used as the NUMBER of heads (diffusers quirk)
    cross_attention_dim: int = 768
    use_linear_projection: bool = False
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    addition_embed_type: Optional[str] = None
    addition_time_embed_dim: Optional[int] = None
    projection_class_embeddings_input_dim: Optional[int] = None
    time_embed_dim: int = 0

    def __post_init__(self):
        if not self.time_embed_dim:
            self.time_embed_dim = 4 * self.block_out_channels[0]
        n = len(self.block_out_channels)
        if isinstance(self.attention_head_dim, int):
            self.attention_head_dim = (self.attention_head_dim,) * n
        if isinstance(self.transformer_layers_per_block, int):
            self.transformer_layers_per_block = (self.transformer_layers_per_block,) * n
        self.block_out_channels = tuple(self.block_out_channels)
        self.down_block_types = tuple(self.down_block_types)
        self.up_block_types = tuple(self.up_block_types)
        self.attention_head_dim = tuple(self.attention_head_dim)
        self.transformer_layers_per_block = tuple(self.transformer_layers_per_block)

    @property
    def is_xl(self) -> bool:
        return self.addition_embed_type == "text_time"

    @property
    def pooled_dim(self) -> int:
        return self.projection_class_embeddings_input_dim - 6 * self.addition_time_embed_dim


def sd1_config() -> UNetConfig:
    """CompVis/stable-diffusion-v1-4, runwayml/stable-diffusion-v1-5 (data/config.yaml:3)."""
    return UNetConfig()


def sdxl_config() -> UNetConfig:
    """stabilityai/stable-diffusion-xl-base-1.0 (data/config-xl.yaml:3)."""
    return UNetConfig(
        sample_size=128, block_out_channels=(320, 640, 1280),
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        transformer_layers_per_block=(1, 2, 10), attention_head_dim=(5, 10, 20),
        cross_attention_dim=2048, use_linear_projection=True, addition_embed_type="text_time",
        addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816)


def sd2_config() -> UNetConfig:
    """stabilityai/stable-diffusion-2-1(-base): pretrained_model.v2 (model_util.py:31-50, 85) - 64-wide heads (5/10/20/20 of
    them), Linear proj_in / proj_out, 1024-d OpenCLIP context; 865,910,724 parameters."""
    return UNetConfig(attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True)


def tiny_sd2_config() -> UNetConfig:
    return UNetConfig(sample_size=16, block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4),
                      cross_attention_dim=128, use_linear_projection=True)


def tiny_sd1_config() -> UNetConfig:
    return UNetConfig(sample_size=16, block_out_channels=(64, 128, 320, 320),
                      attention_head_dim=(8, 8, 8, 8), cross_attention_dim=128)   # head dims 8 / 16 / 40 / 40


def tiny_sdxl_config() -> UNetConfig:
    return UNetConfig(
        sample_size=16, block_out_channels=(64, 128, 256),
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        transformer_layers_per_block=(1, 1, 2), attention_head_dim=(1, 2, 4),
        cross_attention_dim=128, use_linear_projection=True, addition_embed_type="text_time",
        addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32)


CONFIGS = {"sd1": sd1_config, "sd2": sd2_config, "sdxl": sdxl_config, "tiny_sd1": tiny_sd1_config,
           "tiny_sd2": tiny_sd2_config, "tiny_sdxl": tiny_sdxl_config}


def config_from_json(path: str) -> UNetConfig:
    """Read a diffusers `unet/config.json`."""
    with open(path) as f:
        raw = json.load(f)
    keys = UNetConfig.__dataclass_fields__.keys()
    kw = {k: raw[k] for k in keys if k in raw and raw[k] is not None}
    for k in ("block_out_channels", "down_block_types", "up_block_types"):
        if k in kw:
            kw[k] = tuple(kw[k])
    return UNetConfig(**kw)
