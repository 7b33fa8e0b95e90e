"""Batch inference with a trained slider at several scales (eval-scripts/generate_images_sd1.py:43-215,
trainscripts/textsliders/generate_images_xl.py:405-512), on the MI355X engine:

    python -m sliders_amd.generate --model sdxl --lora_weight models/age_alpha1.0_rank4_noxattn/age_alpha1.0_rank4_noxattn_last.pt \
        --prompt "image of a person" --scales=-2,-1,0,1,2 --start_noise 750 --seed 0 --out images/ [--model_path DIR | --synthetic]

Like the reference's consumers it reads rank / alpha / train_method from the checkpoint's FILE NAME
(XL-sliders-inference.ipynb cell 6: 'rank4', 'alpha1', 'noxattn' / 'full' substrings), builds the network, strict-loads
the .pt, and for every scale runs the 50-step DDIM loop with the slider switched on below `start_noise`.
"""
from __future__ import annotations

import argparse
import os
import re
from typing import Tuple

import torch


def parse_slider_name(path: str) -> Tuple[int, float, str]:
    """(rank, alpha, train_method) from a slider file name, with the reference notebooks' defaults (4, 1.0, 'noxattn')."""
    name = os.path.basename(path)
    rank, alpha, method = 4, 1.0, "noxattn"
    m = re.search(r"rank(\d+)", name)
    if m:
        rank = int(m.group(1))
    m = re.search(r"alpha(\d+(?:\.\d+)?)", name)
    if m:
        alpha = float(m.group(1))
    for cand in ("noxattn-hspace-last", "noxattn-hspace", "xattn-strict", "innoxattn", "selfattn", "noxattn", "xattn", "full"):
        if cand in name:
            method = cand
            break
    return rank, alpha, method


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("--model", default="sdxl", choices=["sdxl", "sd1"])
    p.add_argument("--model_path", default=None, help="diffusers-format model directory (unet/, vae/, text encoders)")
    p.add_argument("--synthetic", action="store_true", help="random-init weights and embeddings (no model files)")
    p.add_argument("--lora_weight", default=None, help="slider checkpoint (.pt) written by the trainers")
    p.add_argument("--prompt", default="image of a person")
    p.add_argument("--scales", default="-2,-1,0,1,2")
    p.add_argument("--start_noise", type=int, default=750)
    p.add_argument("--ddim_steps", type=int, default=50)
    p.add_argument("--guidance_scale", type=float, default=7.5)
    p.add_argument("--scheduler", default="ddim", choices=["ddim", "lms", "euler", "euler_a", "ddpm"],
                   help="ddim: fused HIP step; lms: what eval-scripts/generate_images_sd1.py:51 constructs; euler: the SDXL "
                        "checkpoints' scheduler_config (generate_images_xl.py)")
    p.add_argument("--res", type=int, default=None)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--out", default="images")
    p.add_argument("--prompts_path", default=None,
                   help="csv with prompt / evaluation_seed / case_number columns (eval-scripts/generate_images_sd1.py:104-107): images go to "
                        "<out>/<slider name>/<scale>/<case_number>_<sample>.png - the layout eval-scripts/clip_score.py and "
                        "sliders_amd.clip_score read")
    p.add_argument("--num_samples", type=int, default=1)
    p.add_argument("--from_case", type=int, default=0)
    p.add_argument("--till_case", type=int, default=1000000)
    return p


def scale_folder(scale: float) -> str:
    """folder name of a slider scale as the reference writes it (generate_images_sd1.py:116-120: '0.5' -> 'half')"""
    return f"{scale:g}".replace("0.5", "half")


def main(argv=None):
    from .cli import _encoded_pairs  # noqa: F401  (text encoders: model_util)
    from .lora_store import LoraStore
    from .model_util import load_unet_engine, synthetic_engine
    from .sampler import SliderSampler
    from .vae import VAE_SCALING, VaeDecoder, random_vae_state_dict
    a = build_parser().parse_args(argv)
    dev = torch.device("cuda", a.device)
    xl = a.model == "sdxl"
    res = a.res or (1024 if xl else 512)
    if a.synthetic:
        eng = synthetic_engine(a.model, dev, a.seed)
        dec = VaeDecoder(random_vae_state_dict(device=dev, seed=a.seed, decoder=True), dev, VAE_SCALING[a.model])
        g = torch.Generator().manual_seed(a.seed)
        ctx = torch.randn(2, 77, eng.cfg.cross_attention_dim, generator=g)
        pooled = torch.randn(2, eng.cfg.pooled_dim, generator=g) if xl else None
    else:
        if not a.model_path:
            raise SystemExit("--model_path (diffusers-format directory) or --synthetic is required")
        from safetensors.torch import load_file
        from . import model_util
        eng = load_unet_engine(a.model_path, dev)
        dec = VaeDecoder(load_file(os.path.join(a.model_path, "vae", "diffusion_pytorch_model.safetensors")), dev, VAE_SCALING[a.model])
        if xl:
            toks, encs = model_util.load_text_encoders_xl(a.model_path, dev, torch.bfloat16)
            (e_u, p_u), (e_t, p_t) = (model_util.encode_prompts_xl(toks, encs, [s]) for s in ("", a.prompt))
            ctx, pooled = torch.cat([e_u, e_t]), torch.cat([p_u, p_t])
        else:
            tok, enc = model_util.load_text_encoder(a.model_path, dev, torch.bfloat16)
            ctx, pooled = torch.cat([model_util.encode_prompts(tok, enc, [s]) for s in ("", a.prompt)]), None
    store = None
    if a.lora_weight:
        rank, alpha, method = parse_slider_name(a.lora_weight)
        store = LoraStore(eng.cfg, rank=rank, alpha=alpha, train_method=method, device=dev, init="none")
        store.load_state_dict(torch.load(a.lora_weight, map_location="cpu"), strict=True)
    smp = SliderSampler(eng, store, dec, scheduler=a.scheduler, scheduler_seed=a.seed)
    os.makedirs(a.out, exist_ok=True)
    from PIL import Image
    scales = [float(v) for v in a.scales.split(",")]
    if a.prompts_path:
        # the reference's evaluation set-up: one row per case, every scale from the same seed, one folder per scale
        import pandas as pd
        df = pd.read_csv(a.prompts_path)
        name = os.path.basename(a.lora_weight).rsplit(".", 1)[0] if a.lora_weight else "no_slider"
        root = os.path.join(a.out, name)
        for s in scales:
            os.makedirs(os.path.join(root, scale_folder(s)), exist_ok=True)
        for _, row in df.iterrows():
            case = int(row.case_number)
            if not (a.from_case <= case <= a.till_case):
                continue
            if a.synthetic:
                c_row, p_row = ctx, pooled
            elif xl:
                (e_u, p_u), (e_t, p_t) = (model_util.encode_prompts_xl(toks, encs, [t]) for t in ("", str(row.prompt)))
                c_row, p_row = torch.cat([e_u, e_t]), torch.cat([p_u, p_t])
            else:
                c_row, p_row = torch.cat([model_util.encode_prompts(tok, enc, [t]) for t in ("", str(row.prompt))]), None
            for num in range(a.num_samples):
                for s in scales:
                    g = torch.Generator().manual_seed(int(row.evaluation_seed) + num)          # the same noise for every scale
                    noise = torch.randn(1, 4, res // 8, res // 8, generator=g)
                    img = smp.generate(c_row.to(dev), noise.to(dev), scale=s, start_noise=a.start_noise, ddim_steps=a.ddim_steps,
                                       guidance_scale=a.guidance_scale, pooled=p_row.to(dev) if p_row is not None else None)
                    Image.fromarray(img[0].cpu().numpy()).save(os.path.join(root, scale_folder(s), f"{case}_{num}.png"))
            print(f"case {case}: {a.num_samples} sample(s) x {len(scales)} scales saved under {root}")
        return root
    for s in scales:
        noise = torch.randn(1, 4, res // 8, res // 8, generator=torch.Generator().manual_seed(a.seed))   # same seed per scale
        img = smp.generate(ctx.to(dev), noise.to(dev), scale=s, start_noise=a.start_noise, ddim_steps=a.ddim_steps,
                           guidance_scale=a.guidance_scale, pooled=pooled.to(dev) if pooled is not None else None)
        Image.fromarray(img[0].cpu().numpy()).save(os.path.join(a.out, f"scale_{s:g}.png"))
        print(f"scale {s:g}: saved {os.path.join(a.out, f'scale_{s:g}.png')}")
    return a.out


if __name__ == "__main__":
    main()
