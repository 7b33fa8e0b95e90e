"""Model loading for the MI355X engine (the reference's trainscripts/textsliders/model_util.py:29-278 uses diffusers
pipelines for this; here only the pieces the hot path needs).

 * `load_unet_engine(name_or_path)`: a diffusers-format directory (`unet/config.json` +
   `unet/diffusion_pytorch_model.safetensors`) or a single-file `.ckpt/.safetensors` checkpoint in the LDM key
   layout (renamed by `ldm_convert.py`) -> UNetEngine.
 * `load_text_encoders_xl / encode_prompts_xl`: CLIP text encoders through the installed `transformers`
   (they run once before the loop - train_lora_xl.py:121-156 - and are not on the hot path).
 * `create_noise_scheduler`: DDIM as configured at model_util.py:237-246 (fused HIP step); ddpm / lms / euler_a from
   sliders_amd/schedulers.py.
No checkpoints exist in the build image (HF_HUB_OFFLINE), so `synthetic_engine` provides seeded random-init
weights with the real shapes for throughput runs and tests.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from .config import CONFIGS, UNetConfig, config_from_json
from .random_init import random_state_dict
from .train_util import DDIMScheduler
from .unet import UNetEngine


def create_noise_scheduler(scheduler_name: str = "ddim", prediction_type: str = "epsilon"):
    """model_util.py:230-277.  "ddim" is the fused HIP step (train_util.DDIMScheduler over slh_cfg_ddim); "ddpm", "lms" and
    "euler_a" are the tensor-op schedulers of sliders_amd/schedulers.py (same interface, device noise per step / sigma
    space)."""
    name = scheduler_name.lower().replace(" ", "_")
    if name == "ddim":
        return DDIMScheduler(prediction_type=prediction_type)
    from . import schedulers
    return schedulers.create(name, prediction_type)


def load_unet_state(name_or_path: str):
    """(UNetConfig, diffusers-layout state dict) of a diffusers-format model directory or of a single-file
    `.safetensors` / `.ckpt` checkpoint in the LDM key layout (model_util.py:104-129, 200-227); no device work."""
    if os.path.isfile(name_or_path):
        from .ldm_convert import LDM_PREFIX, convert_ldm_unet_state_dict
        if name_or_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            raw = load_file(name_or_path)
        else:
            raw = torch.load(name_or_path, map_location="cpu", weights_only=True)
            raw = raw.get("state_dict", raw)
        is_xl = any(k.startswith(LDM_PREFIX + "label_emb.") for k in raw)
        probe = raw.get(LDM_PREFIX + "input_blocks.1.1.proj_in.weight")
        # SD-2.x single files: Linear proj_in (2-D weight) without SDXL's label_emb MLP; same key renaming
        is_v2 = not is_xl and probe is not None and probe.ndim == 2
        cfg = CONFIGS["sdxl" if is_xl else ("sd2" if is_v2 else "sd1")]()
        return cfg, convert_ldm_unet_state_dict(raw, cfg)
    unet_dir = os.path.join(name_or_path, "unet")
    cfg_path = os.path.join(unet_dir, "config.json")
    if not os.path.isfile(cfg_path):
        raise FileNotFoundError(
            f"{cfg_path} not found: pass a local diffusers-format model directory (this image has no network; "
            f"use synthetic_engine() for random-init weights)")
    cfg = config_from_json(cfg_path)
    from safetensors.torch import load_file
    wpath = os.path.join(unet_dir, "diffusion_pytorch_model.safetensors")
    if not os.path.isfile(wpath):
        wpath = os.path.join(unet_dir, "diffusion_pytorch_model.fp16.safetensors")
    return cfg, load_file(wpath)


def load_unet_engine(name_or_path: str, device="cuda:0") -> UNetEngine:
    cfg, sd = load_unet_state(name_or_path)
    return UNetEngine(cfg, sd, device)


def synthetic_engine(model: str = "sdxl", device="cuda:0", seed: int = 0) -> UNetEngine:
    cfg = CONFIGS[model]()
    return UNetEngine(cfg, random_state_dict(cfg, device, seed), device)


def load_text_encoder(name_or_path: str, device, dtype=torch.bfloat16, v2: bool = False):
    """SD-1.x / SD-2.x: one CLIP tokenizer + text encoder (model_util.py:29-71).  v2: the OpenCLIP-H encoder is cut after
    its 23rd layer (penultimate-layer embeddings, the reference's "default is clip skip 2")."""
    from transformers import CLIPTextModel, CLIPTokenizer
    tok = CLIPTokenizer.from_pretrained(name_or_path, subfolder="tokenizer")
    kw = {"num_hidden_layers": 23} if v2 else {}
    enc = CLIPTextModel.from_pretrained(name_or_path, subfolder="text_encoder", **kw).to(device, dtype).eval()
    return tok, enc


@torch.no_grad()
def encode_prompts(tokenizer, text_encoder, prompts) -> torch.Tensor:
    """train_util.py:60-74: last hidden state of the CLIP text encoder, (n, 77, 768)."""
    ids = tokenizer(prompts, padding="max_length", max_length=tokenizer.model_max_length, truncation=True,
                    return_tensors="pt").input_ids.to(text_encoder.device)
    return text_encoder(ids)[0]


def load_text_encoders_xl(name_or_path: str, device, dtype=torch.bfloat16):
    from transformers import CLIPTextModel, CLIPTextModelWithProjection, CLIPTokenizer
    toks = [CLIPTokenizer.from_pretrained(name_or_path, subfolder="tokenizer"),
            CLIPTokenizer.from_pretrained(name_or_path, subfolder="tokenizer_2", pad_token_id=0)]
    encs = [CLIPTextModel.from_pretrained(name_or_path, subfolder="text_encoder").to(device, dtype).eval(),
            CLIPTextModelWithProjection.from_pretrained(name_or_path, subfolder="text_encoder_2").to(device, dtype).eval()]
    return toks, encs


@torch.no_grad()
def encode_prompts_xl(tokenizers, text_encoders, prompts, num_images_per_prompt: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    """train_util.py:77-133: penultimate hidden states of both encoders concatenated (77 x 2048) + pooled (1280)."""
    embeds, pooled = [], None
    for tok, enc in zip(tokenizers, text_encoders):
        ids = tok(prompts, padding="max_length", max_length=tok.model_max_length, truncation=True,
                  return_tensors="pt").input_ids.to(enc.device)
        out = enc(ids, output_hidden_states=True)
        pooled = out[0]
        embeds.append(out.hidden_states[-2])
    text = torch.concat(embeds, dim=-1).repeat_interleave(num_images_per_prompt, dim=0)
    return text, pooled.repeat_interleave(num_images_per_prompt, dim=0)
