"""Seeded random-init UNet weights with the real SD-1.x / SDXL shapes (diffusers state-dict keys).

No checkpoints exist offline (HF_HUB_OFFLINE, no weights on disk), so throughput runs use weights drawn
directly on the GPU with PyTorch's default init bounds (U(-1/sqrt(fan_in), 1/sqrt(fan_in)); norms 1/0).
bench.py says so in its `data` field.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .config import UNetConfig
from .modules import build_tree


def random_state_dict(cfg: UNetConfig, device="cpu", seed: int = 0, dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def uniform(shape, bound):
        return ((torch.rand(shape, generator=g, device=device, dtype=torch.float32) * 2 - 1) * bound).to(dtype)

    for name, node in build_tree(cfg).named_modules():
        if node.cls in ("Linear", "LoRACompatibleLinear"):
            b = 1.0 / math.sqrt(node.in_dim)
            sd[name + ".weight"] = uniform((node.out_dim, node.in_dim), b)
            if node.bias:
                sd[name + ".bias"] = uniform((node.out_dim,), b)
        elif node.cls in ("Conv2d", "LoRACompatibleConv"):
            k = node.kernel
            b = 1.0 / math.sqrt(node.in_dim * k * k)
            sd[name + ".weight"] = uniform((node.out_dim, node.in_dim, k, k), b)
            sd[name + ".bias"] = uniform((node.out_dim,), b)
        elif node.cls in ("GroupNorm", "LayerNorm"):
            sd[name + ".weight"] = torch.ones(node.in_dim, device=device, dtype=dtype)
            sd[name + ".bias"] = torch.zeros(node.in_dim, device=device, dtype=dtype)
    return sd
