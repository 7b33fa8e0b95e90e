"""Single-file Stable Diffusion checkpoints (`.safetensors` / `.ckpt` in the original LDM key layout) -> the
diffusers UNet key layout the engine's WeightStore consumes.

The reference loads such files through diffusers (`StableDiffusionPipeline.from_ckpt`,
`StableDiffusionXLPipeline.from_single_file`: trainscripts/textsliders/model_util.py:77-101, 179-197); diffusers is a
third-party dependency that is absent here, so this module restates the UNet part of its published conversion
(`convert_ldm_unet_checkpoint`): a pure renaming, no tensor is reshaped.

LDM layout (prefix `model.diffusion_model.`):
    time_embed.{0,2}                      -> time_embedding.linear_{1,2}
    label_emb.0.{0,2}            (SDXL)   -> add_embedding.linear_{1,2}
    input_blocks.0.0                      -> conv_in
    input_blocks.{1 + b*(L+1) + j}.0      -> down_blocks.b.resnets.j          (L = layers_per_block)
    input_blocks.{1 + b*(L+1) + j}.1      -> down_blocks.b.attentions.j
    input_blocks.{(b+1)*(L+1)}.0.op       -> down_blocks.b.downsamplers.0.conv
    middle_block.{0,1,2}                  -> mid_block.resnets.0 / attentions.0 / resnets.1
    output_blocks.{b*(L+1) + j}.0         -> up_blocks.b.resnets.j
    output_blocks.{b*(L+1) + j}.1         -> up_blocks.b.attentions.j   (cross-attention up block)
    output_blocks.{b*(L+1) + L}.{1|2}.conv-> up_blocks.b.upsamplers.0.conv   (index 2 when the block has attention)
    out.{0,2}                             -> conv_norm_out / conv_out
inside a resnet:  in_layers.0 -> norm1, in_layers.2 -> conv1, emb_layers.1 -> time_emb_proj, out_layers.0 -> norm2,
                  out_layers.3 -> conv2, skip_connection -> conv_shortcut
inside a transformer the names are identical (norm, proj_in, transformer_blocks.*, proj_out).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .config import UNetConfig

LDM_PREFIX = "model.diffusion_model."

_MLP_IDX = {"0": 1, "2": 2}                    # Sequential(Linear, SiLU, Linear) positions -> linear_1 / linear_2
_MLP_POS = {"linear_1": 0, "linear_2": 2}
_RES = [("in_layers.0", "norm1"), ("in_layers.2", "conv1"), ("emb_layers.1", "time_emb_proj"),
        ("out_layers.0", "norm2"), ("out_layers.3", "conv2"), ("skip_connection", "conv_shortcut")]


def _res_to_diffusers(rest: str) -> str:
    for ldm, dif in _RES:
        if rest == ldm or rest.startswith(ldm + "."):
            return dif + rest[len(ldm):]
    raise KeyError(f"unknown resnet sub-key '{rest}'")


def _res_to_ldm(rest: str) -> str:
    for ldm, dif in _RES:
        if rest == dif or rest.startswith(dif + "."):
            return ldm + rest[len(dif):]
    raise KeyError(f"unknown resnet sub-key '{rest}'")


def ldm_to_diffusers_key(key: str, cfg: UNetConfig) -> str:
    """One UNet key of the LDM layout (without `model.diffusion_model.`) -> diffusers layout."""
    L = cfg.layers_per_block
    parts = key.split(".")
    head = parts[0]
    if head == "time_embed":
        return "time_embedding.linear_%d." % _MLP_IDX[parts[1]] + ".".join(parts[2:])
    if head == "label_emb":
        return "add_embedding.linear_%d." % _MLP_IDX[parts[2]] + ".".join(parts[3:])
    if head == "out":
        return {"0": "conv_norm_out", "2": "conv_out"}[parts[1]] + "." + ".".join(parts[2:])
    if head == "middle_block":
        idx, rest = int(parts[1]), ".".join(parts[2:])
        if idx == 1:
            return "mid_block.attentions.0." + rest
        return f"mid_block.resnets.{idx // 2}." + _res_to_diffusers(rest)
    if head == "input_blocks":
        i, sub, rest = int(parts[1]), int(parts[2]), ".".join(parts[3:])
        if i == 0:
            return "conv_in." + rest
        b, j = (i - 1) // (L + 1), (i - 1) % (L + 1)
        if j == L:                                   # the (L+1)-th entry of a group is the downsampler
            assert rest.startswith("op."), key
            return f"down_blocks.{b}.downsamplers.0.conv." + rest[3:]
        if sub == 0:
            return f"down_blocks.{b}.resnets.{j}." + _res_to_diffusers(rest)
        return f"down_blocks.{b}.attentions.{j}." + rest
    if head == "output_blocks":
        i, sub, rest = int(parts[1]), int(parts[2]), ".".join(parts[3:])
        b, j = i // (L + 1), i % (L + 1)
        if sub == 0:
            return f"up_blocks.{b}.resnets.{j}." + _res_to_diffusers(rest)
        has_attn = cfg.up_block_types[b] != "UpBlock2D"
        if sub == 1 and has_attn:
            return f"up_blocks.{b}.attentions.{j}." + rest
        assert rest.startswith("conv."), key         # upsampler: sub 1 without attention, sub 2 with
        return f"up_blocks.{b}.upsamplers.0.conv." + rest[5:]
    raise KeyError(f"unknown LDM UNet key '{key}'")


def diffusers_to_ldm_key(key: str, cfg: UNetConfig) -> str:
    """Inverse of ldm_to_diffusers_key (used by the round-trip tests and to export single-file checkpoints)."""
    L = cfg.layers_per_block
    parts = key.split(".")
    head = parts[0]
    if head == "time_embedding":
        return "time_embed.%d." % _MLP_POS[parts[1]] + ".".join(parts[2:])
    if head == "add_embedding":
        return "label_emb.0.%d." % _MLP_POS[parts[1]] + ".".join(parts[2:])
    if head == "conv_in":
        return "input_blocks.0.0." + ".".join(parts[1:])
    if head == "conv_norm_out":
        return "out.0." + ".".join(parts[1:])
    if head == "conv_out":
        return "out.2." + ".".join(parts[1:])
    if head == "mid_block":
        kind, idx, rest = parts[1], int(parts[2]), ".".join(parts[3:])
        if kind == "attentions":
            return "middle_block.1." + rest
        return f"middle_block.{2 * idx}." + _res_to_ldm(rest)
    if head == "down_blocks":
        b, kind, j, rest = int(parts[1]), parts[2], int(parts[3]), ".".join(parts[4:])
        if kind == "downsamplers":
            return f"input_blocks.{(b + 1) * (L + 1)}.0.op." + rest[len("conv."):]
        i = 1 + b * (L + 1) + j
        if kind == "resnets":
            return f"input_blocks.{i}.0." + _res_to_ldm(rest)
        return f"input_blocks.{i}.1." + rest
    if head == "up_blocks":
        b, kind, j, rest = int(parts[1]), parts[2], int(parts[3]), ".".join(parts[4:])
        has_attn = cfg.up_block_types[b] != "UpBlock2D"
        if kind == "upsamplers":
            return f"output_blocks.{b * (L + 1) + L}.{2 if has_attn else 1}.conv." + rest[len("conv."):]
        i = b * (L + 1) + j
        if kind == "resnets":
            return f"output_blocks.{i}.0." + _res_to_ldm(rest)
        return f"output_blocks.{i}.1." + rest
    raise KeyError(f"unknown diffusers UNet key '{key}'")


def convert_ldm_unet_state_dict(sd: Dict[str, torch.Tensor], cfg: UNetConfig,
                                prefix: Optional[str] = LDM_PREFIX) -> Dict[str, torch.Tensor]:
    """Pick the UNet tensors out of a single-file checkpoint's state dict and rename them to the diffusers layout.
    Keys outside `prefix` (VAE, text encoders, EMA copies) are ignored; an unknown UNet key raises."""
    out = {}
    for k, v in sd.items():
        if prefix:
            if not k.startswith(prefix):
                continue
            k = k[len(prefix):]
        out[ldm_to_diffusers_key(k, cfg)] = v
    if not out:
        raise ValueError(f"no '{prefix}*' tensors in this checkpoint: not an LDM-layout Stable Diffusion file")
    return out


def load_single_file_unet(path: str, cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """`.safetensors` or torch `.ckpt` single-file checkpoint -> diffusers-layout UNet state dict."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
    return convert_ldm_unet_state_dict(sd, cfg)
