"""CPU oracle: the reference's LoRA adapter semantics over the oracle UNet.

TEST INFRASTRUCTURE ONLY (see unet_oracle.py header).

Restates trainscripts/textsliders/lora.py of the reference:
  * LoRAModule (lora.py:50-112): y = org(x) + up(down(x)) * multiplier * (alpha / rank); down is a
    Linear(in, r) or a Conv2d(in, r, k, stride, padding) without bias, up a Linear(r, out) / 1x1 Conv2d;
    down ~ kaiming_uniform(a=1), up = 0; `alpha` is a buffer.
  * LoRANetwork (lora.py:115-258): target discovery over named_modules() by class-name strings and the
    train_method name filters (lora.py:164-218), module names `lora_unet_<path with _>`; `with network:`
    sets multiplier = lora_scale, leaving the block sets it to 0 (lora.py:252-258).
Pinned in this container against the reference's own classes by tests/golden/make_golden.py (same key list,
same forward values on the tiny configs): tests/test_oracle.py::test_lora_oracle_matches_reference_golden.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

_CONV_TARGETS = ["ResnetBlock2D", "Downsample2D", "Upsample2D", "DownBlock2D", "UpBlock2D"]
_LEAVES = ["Linear", "Conv2d", "LoRACompatibleLinear", "LoRACompatibleConv"]


class LoRAModuleOracle(nn.Module):
    def __init__(self, lora_name: str, org: nn.Module, multiplier: float, rank: int, alpha: float, apply: bool = True):
        super().__init__()
        self.lora_name = lora_name
        cls = org.__class__.__name__
        if "Linear" in cls:
            self.lora_dim = rank
            self.lora_down = nn.Linear(org.in_features, rank, bias=False)
            self.lora_up = nn.Linear(rank, org.out_features, bias=False)
        else:
            self.lora_dim = min(rank, org.in_channels, org.out_channels)
            self.lora_down = nn.Conv2d(org.in_channels, self.lora_dim, org.kernel_size, org.stride, org.padding,
                                       bias=False)
            self.lora_up = nn.Conv2d(self.lora_dim, org.out_channels, (1, 1), (1, 1), bias=False)
        alpha = rank if alpha is None or alpha == 0 else alpha
        self.scale = alpha / self.lora_dim
        self.register_buffer("alpha", torch.tensor(alpha))
        nn.init.kaiming_uniform_(self.lora_down.weight, a=1)
        nn.init.zeros_(self.lora_up.weight)
        self.multiplier = multiplier
        if apply:              # LoRAModule.apply_to (lora.py:103-106); duplicate visits are built but never applied
            self._org_forward = org.forward
            org.forward = self.forward

    def forward(self, x):
        return self._org_forward(x) + self.lora_up(self.lora_down(x)) * self.multiplier * self.scale


class LoRANetworkOracle(nn.Module):
    def __init__(self, unet: nn.Module, rank: int = 4, multiplier: float = 1.0, alpha: float = 1.0,
                 train_method: str = "full", c3lier: bool = True):
        super().__init__()
        self.lora_scale = 1
        targets = ["Attention"] + (_CONV_TARGETS if c3lier else [])
        self.unet_loras: List[LoRAModuleOracle] = []
        seen = set()
        for name, module in unet.named_modules():
            if train_method in ("noxattn", "noxattn-hspace", "noxattn-hspace-last"):
                if "attn2" in name or "time_embed" in name:
                    continue
            elif train_method == "innoxattn":
                if "attn2" in name:
                    continue
            elif train_method == "selfattn":
                if "attn1" not in name:
                    continue
            elif train_method in ("xattn", "xattn-strict"):
                if "attn2" not in name:
                    continue
            elif train_method != "full":
                raise NotImplementedError(train_method)
            if module.__class__.__name__ not in targets:
                continue
            for child_name, child in module.named_modules():
                if child.__class__.__name__ not in _LEAVES:
                    continue
                if train_method == "xattn-strict" and "out" in child_name:
                    continue
                if train_method == "noxattn-hspace" and "mid_block" not in name:
                    continue
                if train_method == "noxattn-hspace-last" and (
                        "mid_block" not in name or ".1" not in name or "conv2" not in child_name):
                    continue
                lora_name = ("lora_unet." + name + "." + child_name).replace(".", "_")
                # the reference constructs the LoRAModule (RNG draws) BEFORE the duplicate-name check (lora.py:206-216)
                m = LoRAModuleOracle(lora_name, child, multiplier, rank, alpha, apply=lora_name not in seen)
                if lora_name in seen:
                    continue
                seen.add(lora_name)
                self.unet_loras.append(m)
        for m in self.unet_loras:
            self.add_module(m.lora_name, m)

    def set_lora_slider(self, scale):
        self.lora_scale = scale

    def __enter__(self):
        for m in self.unet_loras:
            m.multiplier = 1.0 * self.lora_scale

    def __exit__(self, *a):
        for m in self.unet_loras:
            m.multiplier = 0

    def prepare_optimizer_params(self):
        params = []
        for m in self.unet_loras:
            params.extend(m.parameters())
        return [{"params": params}]
