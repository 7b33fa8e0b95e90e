"""prompts.yaml schema, prompt-embedding containers and the paired guidance loss of the reference
(trainscripts/textsliders/prompt_util.py:11-174), same names and semantics.

`PromptEmbedsPair.loss` is kept call-compatible (it takes the four epsilon tensors and returns the scalar loss
through `loss_fn`, prompt_util.py:108-148) for code that drives the engine through torch autograd; the fused
trainer (sliders_amd/trainer.py) evaluates the same formula with slh_guidance_loss instead.
"""
from __future__ import annotations

import copy
from typing import List, Literal, Optional, Union

import torch
import yaml
from pydantic import BaseModel, model_validator

ACTION_TYPES = Literal["erase", "enhance"]


class PromptEmbedsXL:
    """SDXL needs (text_embeds, pooled_embeds)."""

    def __init__(self, *args) -> None:
        self.text_embeds = args[0]
        self.pooled_embeds = args[1]


PROMPT_EMBEDDING = Union[torch.Tensor, PromptEmbedsXL]


class PromptEmbedsCache:
    def __init__(self):
        self.prompts = {}

    def __setitem__(self, name: str, value) -> None:
        self.prompts[name] = value

    def __getitem__(self, name: str):
        return self.prompts.get(name)


class PromptSettings(BaseModel):
    target: str
    positive: Optional[str] = None      # if None, target is used
    unconditional: str = ""
    neutral: Optional[str] = None       # if None, unconditional is used
    action: ACTION_TYPES = "erase"
    guidance_scale: float = 1.0
    resolution: int = 512
    dynamic_resolution: bool = False
    batch_size: int = 1
    dynamic_crops: bool = False          # XL only

    @model_validator(mode="before")
    @classmethod
    def fill_prompts(cls, values):
        values = dict(values)
        if "target" not in values:
            raise ValueError("target must be specified")
        if "positive" not in values:
            values["positive"] = values["target"]
        if "unconditional" not in values:
            values["unconditional"] = ""
        if "neutral" not in values:
            values["neutral"] = values["unconditional"]
        return values


class PromptEmbedsPair:
    def __init__(self, loss_fn, target, positive, unconditional, neutral, settings: PromptSettings) -> None:
        self.loss_fn = loss_fn
        self.target = target
        self.positive = positive
        self.unconditional = unconditional
        self.neutral = neutral
        self.guidance_scale = settings.guidance_scale
        self.resolution = settings.resolution
        self.dynamic_resolution = settings.dynamic_resolution
        self.batch_size = settings.batch_size
        self.dynamic_crops = settings.dynamic_crops
        self.action = settings.action

    def _erase(self, target_latents, positive_latents, unconditional_latents, neutral_latents):
        """Target latents are going not to have the positive concept."""
        return self.loss_fn(target_latents,
                            neutral_latents - self.guidance_scale * (positive_latents - unconditional_latents))

    def _enhance(self, target_latents, positive_latents, unconditional_latents, neutral_latents):
        """Target latents are going to have the positive concept."""
        return self.loss_fn(target_latents,
                            neutral_latents + self.guidance_scale * (positive_latents - unconditional_latents))

    def loss(self, **kwargs):
        if self.action == "erase":
            return self._erase(**kwargs)
        if self.action == "enhance":
            return self._enhance(**kwargs)
        raise ValueError("action must be erase or enhance")


def load_prompts_from_yaml(path, attributes: Optional[List[str]] = None) -> List[PromptSettings]:
    attributes = attributes or []
    with open(path, "r") as f:
        prompts = yaml.safe_load(f)
    if len(prompts) == 0:
        raise ValueError("prompts file is empty")
    if len(attributes) != 0:
        newprompts = []
        for prompt in prompts:
            for att in attributes:
                c = copy.deepcopy(prompt)
                for key in ("target", "positive", "neutral", "unconditional"):
                    c[key] = att + " " + c[key]
                newprompts.append(c)
    else:
        newprompts = copy.deepcopy(prompts)
    return [PromptSettings(**prompt) for prompt in newprompts]
