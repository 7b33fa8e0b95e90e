"""Command line of the reference's image-slider trainers (trainscripts/imagesliders/train_lora-scale-xl.py:418-545,
train_lora-scale.py):

    python trainscripts/imagesliders/train_lora-scale-xl.py --name 'eyeslider' --rank 4 --alpha 1 \
        --config_file 'trainscripts/imagesliders/data/config-xl.yaml' --folder_main 'datasets/eyesize/' \
        --folders 'bigsize, smallsize' --scales '1, -1'

Same flags (`--alpha` is required, `--folders` / `--scales` default to 'verylow, low, high, veryhigh' / '-2, -1, 1, 2',
`--stylecheck a-b` loops over folder_main + i), same save names (`{name}_alpha{a}_rank{r}_{method}`).  `--synthetic` (ours)
replaces checkpoints, text encoders and the image folders by seeded random weights / embeddings / images.
"""
from __future__ import annotations

import argparse
import os
import random
from pathlib import Path

import numpy as np
import torch

from . import config_util, prompt_util
from .train_util import get_add_time_ids, get_random_resolution_in_bucket
from .cli import (LrSchedule, _encoded_pairs, _synthetic_pairs, adapter_state_dtype, check_model_files, check_supported,
                  optimizer_options)
from .image_trainer import ImageSliderTrainer
from .lora_store import LoraStore
from .model_util import load_unet_engine, synthetic_engine
from .parallel import StepSampler, world_info
from .vae import VAE_SCALING, VaeEncoder, random_vae_state_dict

IMAGE_EXTS = (".png", ".jpg", ".jpeg", ".webp")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("--config_file", required=True, help="Config file for training.")
    p.add_argument("--alpha", type=float, required=True, help="LoRA weight.")
    p.add_argument("--rank", type=int, required=False, help="Rank of LoRA.", default=4)
    p.add_argument("--device", type=int, required=False, default=0, help="Device to train on.")
    p.add_argument("--name", type=str, required=False, default=None, help="name of the slider")
    p.add_argument("--attributes", type=str, required=False, default=None, help="attritbutes to disentangle (comma seperated string)")
    p.add_argument("--folder_main", type=str, required=True, help="The folder to check")
    p.add_argument("--stylecheck", type=str, required=False, default=None, help="a-b: train one slider per folder_main + i")
    p.add_argument("--folders", type=str, required=False, default="verylow, low, high, veryhigh",
                   help="folders with different attribute-scaled images")
    p.add_argument("--scales", type=str, required=False, default="-2, -1, 1, 2", help="scales for different attribute-scaled images")
    p.add_argument("--synthetic", action="store_true", help="random-init weights and embeddings (no model files); random images too unless the folders exist")
    p.add_argument("--seed", type=int, default=0)
    return p


def parse_folders_scales(folders: str, scales: str):
    f = [x.strip() for x in folders.split(",")]
    s = [int(x.strip()) for x in scales.split(",")]
    if len(s) != len(f):
        raise Exception("the number of folders need to match the number of scales")
    return f, s


def load_vae_encoder(name_or_path: str, dev, xl: bool) -> VaeEncoder:
    """`vae/diffusion_pytorch_model.safetensors` of a diffusers-format model directory (model_util.py:74-77, 179-180)."""
    from safetensors.torch import load_file
    import json
    vdir = os.path.join(name_or_path, "vae")
    wpath = os.path.join(vdir, "diffusion_pytorch_model.safetensors")
    if not os.path.isfile(wpath):
        raise FileNotFoundError(f"{wpath} not found: the image sliders need the VAE of a diffusers-format model directory")
    scaling = VAE_SCALING["sdxl" if xl else "sd1"]
    cfg_path = os.path.join(vdir, "config.json")
    if os.path.isfile(cfg_path):
        scaling = float(json.load(open(cfg_path)).get("scaling_factor", scaling))
    return VaeEncoder(load_file(wpath), dev, scaling)


def list_images(folder: str):
    return sorted(f for f in os.listdir(folder) if any(e in f for e in IMAGE_EXTS))


def open_image(path: str, size: int) -> torch.Tensor:
    from PIL import Image
    return VaeEncoder.preprocess(Image.open(path).convert("RGB").resize((size, size)))


def train(config, prompts, device: int, xl: bool, folder_main: str, folders, scales, synthetic: bool, seed: int = 0):
    check_supported(config, image_slider=True)
    if not synthetic:
        check_model_files(config.pretrained_model.name_or_path)
    rank, world = world_info()
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    scales = np.array(scales)
    folders = np.array(folders)
    size = 512 if xl else 256                              # the reference resizes every image (train_lora-scale-xl.py:220-221)
    if synthetic:
        eng = synthetic_engine("sdxl" if xl else "sd1", dev, seed)
        vae = VaeEncoder(random_vae_state_dict(device=dev, seed=seed), dev, VAE_SCALING["sdxl" if xl else "sd1"])
    else:
        eng = load_unet_engine(config.pretrained_model.name_or_path, dev)
        vae = load_vae_encoder(config.pretrained_model.name_or_path, dev, xl)
    torch.manual_seed(seed)
    store = LoraStore(eng.cfg, rank=config.network.rank, alpha=config.network.alpha,
                      train_method=config.network.training_method,
                      # train_lora-scale-xl.py:61-63: conv targets only for network.type c3lier; imagesliders/lora.py's list
                      network_type="c3lier-image" if config.network.type == "c3lier" else "lierla", device=dev,
                      kaiming_a=5 ** 0.5, state_dtype=adapter_state_dtype(config, rank))
    opt = optimizer_options(config.train)
    hw = size // 8
    tr = ImageSliderTrainer(eng, store, vae, hw, hw, batch_size=1, lr=config.train.lr, betas=opt["betas"], eps=opt["eps"],
                            weight_decay=opt["weight_decay"], max_denoising_steps=config.train.max_denoising_steps,
                            process_group=torch.distributed.group.WORLD if world > 1 else None, optimizer=opt["name"],
                            optimizer_kwargs={k: v for k, v in opt.items() if k not in ("name", "betas", "eps", "weight_decay")})
    pairs = (_synthetic_pairs(eng.cfg, prompts, dev, seed) if synthetic else
             _encoded_pairs(eng.cfg, prompts, config.pretrained_model.name_or_path, dev,
                            config_util.parse_precision(config.train.precision)))
    # k ~ U{1 .. max-1} for the XL script (train_lora-scale-xl.py:191-193) but U{1 .. max-2} for the SD-1.x one, whose
    # randint has max_denoising_steps - 1 as its exclusive bound (train_lora-scale.py:186-188)
    samp = StepSampler(seed, rank, world, len(pairs), config.train.max_denoising_steps - (0 if xl else 1))
    sched = LrSchedule(config.train.lr_scheduler, config.train.lr, config.train.iterations)
    pyrng = random.Random(seed * 7919 + rank)              # image / scale choice is rank-local (different data per rank)
    save_path = Path(config.save.path)
    dtype = config_util.parse_precision(config.train.precision)
    # --synthetic replaces the model files; image folders are still read when they exist (random images otherwise)
    have_image_files = all(os.path.isdir(f"{folder_main}/{f}") and list_images(f"{folder_main}/{f}/") for f in folders)
    for i in range(config.train.iterations):
        k, pi = samp.next()
        s, pair = pairs[pi]
        scale_to_look = abs(pyrng.choice(list(scales)))
        if synthetic and not have_image_files:
            g = torch.Generator().manual_seed(pyrng.randrange(1 << 30))
            img_low = VaeEncoder.preprocess(torch.randint(0, 256, (size, size, 3), generator=g, dtype=torch.uint8))
            img_high = VaeEncoder.preprocess(torch.randint(0, 256, (size, size, 3), generator=g, dtype=torch.uint8))
        else:
            folder1 = folders[scales == -scale_to_look][0]
            folder2 = folders[scales == scale_to_look][0]
            ims = list_images(f"{folder_main}/{folder1}/")
            im = ims[pyrng.randint(0, len(ims) - 1)]
            img_low = open_image(f"{folder_main}/{folder1}/{im}", size)
            img_high = open_image(f"{folder_main}/{folder2}/{im}", size)
        # one seed for both images, so both get the same draws (train_lora-scale-xl.py:222-246; seed range = quirk D.12).
        # In the reference the diffusion noise is the FIRST draw of the re-seeded CPU generator (randn_tensor with the
        # generator torch.manual_seed returned), while latent_dist.sample(None) draws from the device's own stream.
        img_seed = pyrng.randint(0, 2 * 15)
        noise = torch.randn(1, 4, hw, hw, generator=torch.Generator().manual_seed(img_seed))
        post_noise = torch.randn(1, 4, hw, hw, generator=torch.Generator().manual_seed(img_seed + (1 << 20)))
        # micro-conditioning from the PROMPT's resolution (bucketed when dynamic_resolution), not from the fixed image
        # size, with optional dynamic crops, in bf16 like the reference (train_lora-scale-xl.py:196-200, 249-254; quirk D.8)
        time_ids = None
        if eng.cfg.is_xl:
            height = width = s.resolution
            if s.dynamic_resolution:
                height, width = get_random_resolution_in_bucket(s.resolution)
            if s.dynamic_crops or (height, width) != (size, size):
                ids = get_add_time_ids(height, width, dynamic_crops=s.dynamic_crops, dtype=torch.bfloat16)
                time_ids = ids.float().repeat(2, 1)
        lr = sched.current()
        lh, ll = tr.iteration(pair, k, img_low.to(dev), img_high.to(dev), float(scale_to_look), post_noise.to(dev),
                              noise.to(dev), lr=lr, time_ids=time_ids)
        sched.step()
        if rank == 0 and (i % 10 == 0 or config.logging.verbose):
            print(f"it {i} k={k} scale={scale_to_look} lr={lr:.3e} Loss*1k: high {lh.item() * 1000:.4f} low {ll.item() * 1000:.4f}")
        if rank == 0 and i % config.save.per_steps == 0 and i != 0 and i != config.train.iterations - 1:
            save_path.mkdir(parents=True, exist_ok=True)
            torch.save(store.state_dict(dtype), save_path / f"{config.save.name}_{i}steps.pt")
    if rank == 0:
        save_path.mkdir(parents=True, exist_ok=True)
        torch.save(store.state_dict(dtype), save_path / f"{config.save.name}_last.pt")
        print("Done.")


def main(xl: bool, argv=None):
    args = build_parser().parse_args(argv)
    config = config_util.load_config_from_yaml(args.config_file)
    if args.name is not None:
        config.save.name = args.name
    attributes = [a.strip() for a in args.attributes.split(",")] if args.attributes is not None else []
    config.network.alpha = args.alpha
    config.network.rank = args.rank
    config.save.name += f"_alpha{args.alpha}"
    config.save.name += f"_rank{config.network.rank}"
    config.save.name += f"_{config.network.training_method}"
    config.save.path += f"/{config.save.name}"
    prompts = prompt_util.load_prompts_from_yaml(config.prompts_file, attributes)
    folders, scales = parse_folders_scales(args.folders, args.scales)
    check_supported(config, image_slider=True)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.distributed.init_process_group("nccl")
        args.device = int(os.environ.get("LOCAL_RANK", "0"))
    if args.stylecheck is not None:
        lo, hi = (int(v) for v in args.stylecheck.split("-"))
        for i in range(lo, hi):
            folder_main = args.folder_main + f"{i}"
            config.save.name = f"{os.path.basename(folder_main)}_alpha{args.alpha}_rank{config.network.rank}"
            config.save.path = f"models/{config.save.name}"
            train(config, prompts, args.device, xl, folder_main, folders, scales, args.synthetic, args.seed)
    else:
        train(config, prompts, args.device, xl, args.folder_main, folders, scales, args.synthetic, args.seed)
