"""Command line of the reference trainers (trainscripts/textsliders/train_lora_xl.py:390-474, train_lora.py:343-429):

    python trainscripts/textsliders/train_lora_xl.py --attributes 'male, female' --name 'ageslider' --rank 4 \
        --alpha 1 --config_file 'trainscripts/textsliders/data/config-xl.yaml'

Same flags, same config.yaml / prompts.yaml schema, same output file names
(`{save.path}/{name}_alpha{a}_rank{r}_{method}/{...}_{i}steps.pt` and `_last.pt`).  `--synthetic` (ours) replaces
the checkpoint and the CLIP text encoders by seeded random-init weights / embeddings with the real shapes, for
machines without model files.  Multi-GPU: launch with torch.distributed.run; ranks share k and all-reduce the flat
LoRA gradient once per step (sliders_amd/parallel.py).
"""
from __future__ import annotations

import argparse
import os
from pathlib import Path

import torch

from . import config_util, prompt_util
from .lora_store import LoraStore
from .model_util import load_unet_engine, synthetic_engine
from .parallel import StepSampler, world_info
from .trainer import PairEmbeds, SliderTrainer


def build_parser(xl: bool) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("--config_file", required=True, help="Config file for training.")
    p.add_argument("--prompts_file", required=False, help="Prompts file for training.", default=None)
    p.add_argument("--alpha", type=float, required=False, default=None, help="LoRA weight.")
    p.add_argument("--rank", type=int, required=False, help="Rank of LoRA.", default=4)
    p.add_argument("--device", type=int, required=False, default=0, help="Device to train on.")
    p.add_argument("--name", type=str, required=False, default=None, help="name of the slider")
    p.add_argument("--attributes", type=str, required=False, default=None, help="attritbutes to disentangle")
    p.add_argument("--synthetic", action="store_true", help="random-init weights and embeddings (no model files)")
    p.add_argument("--seed", type=int, default=0)
    return p


def build_pairs(cfg, prompts, encode, dev):
    """One PairEmbeds per PromptSettings.  `encode(text) -> (embeds (1,77,D), pooled (1,P) | None)` is called once per
    distinct prompt string, like the reference's PromptEmbedsCache (train_lora_xl.py:121-151); every ctx tensor is
    cat([unconditional, X]) repeated batch_size times, as concat_embeddings builds it (train_util.py:136-141)."""
    cache = {}

    def emb(text):
        if text not in cache:
            e, p = encode(text)
            cache[text] = (e.to(dev, torch.bfloat16), p.to(dev, torch.bfloat16) if p is not None else None)
        return cache[text]

    pairs = []
    for s in prompts:
        t, po, ne, un = emb(s.target), emb(s.positive), emb(s.neutral), emb(s.unconditional)
        cat = lambda x: torch.cat([un[0], x[0]]).repeat_interleave(s.batch_size, dim=0).contiguous()
        pc = (lambda x: torch.cat([un[1], x[1]]).repeat_interleave(s.batch_size, dim=0).contiguous()) if cfg.is_xl else (lambda x: None)
        pairs.append((s, PairEmbeds(cat(t), cat(po), cat(ne), cat(un), pc(t), pc(po), pc(ne), pc(un),
                                    guidance_scale=s.guidance_scale, action=s.action)))
    return pairs


def _synthetic_pairs(cfg, prompts, dev, seed):
    """Seeded randn embeddings keyed by the prompt string (no text-encoder weights in the build image)."""
    def encode(text):
        g = torch.Generator().manual_seed(seed + (hash(text) & 0xFFFFFF))
        e = torch.randn(1, 77, cfg.cross_attention_dim, generator=g)
        return e, (torch.randn(1, cfg.pooled_dim, generator=g) if cfg.is_xl else None)
    return build_pairs(cfg, prompts, encode, dev)


def _encoded_pairs(cfg, prompts, name_or_path, dev, dtype):
    """Real checkpoints: CLIP text encoder(s) through `transformers`, run once before the loop (not on the hot path)."""
    from . import model_util
    if cfg.is_xl:
        toks, encs = model_util.load_text_encoders_xl(name_or_path, dev, dtype)
        encode = lambda text: model_util.encode_prompts_xl(toks, encs, [text])
    else:
        tok, enc = model_util.load_text_encoder(name_or_path, dev, dtype)
        encode = lambda text: (model_util.encode_prompts(tok, enc, [text]), None)
    pairs = build_pairs(cfg, prompts, encode, dev)
    torch.cuda.empty_cache()                    # the encoders are dropped here, as the reference does (train_lora_xl.py:153-156)
    return pairs


def train(config: config_util.RootConfig, prompts, device: int, xl: bool, synthetic: bool, seed: int = 0):
    rank, world = world_info()
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    if config.network.type != "c3lier":
        print("note: network.type lierla -> attention-only targets")
    if synthetic:
        eng = synthetic_engine("sdxl" if xl else "sd1", dev, seed)
    else:
        eng = load_unet_engine(config.pretrained_model.name_or_path, dev)
        if eng.cfg.is_xl != xl:
            raise ValueError(f"{config.pretrained_model.name_or_path} is {'an SDXL' if eng.cfg.is_xl else 'an SD-1.x'} "
                             f"UNet but the {'XL' if xl else 'SD-1.x'} entry point was used")
    torch.manual_seed(seed)
    store = LoraStore(eng.cfg, rank=config.network.rank, alpha=config.network.alpha,
                      train_method=config.network.training_method, network_type=config.network.type, device=dev)
    res = prompts[0].resolution
    hw = res // 8
    tr = SliderTrainer(eng, store, hw, hw, batch_size=prompts[0].batch_size, lr=config.train.lr,
                       max_denoising_steps=config.train.max_denoising_steps)
    if synthetic:
        pairs = _synthetic_pairs(eng.cfg, prompts, dev, seed)
    else:
        pairs = _encoded_pairs(eng.cfg, prompts, config.pretrained_model.name_or_path, dev,
                               config_util.parse_precision(config.train.precision))
    samp = StepSampler(seed, rank, world, len(pairs), config.train.max_denoising_steps)
    save_path = Path(config.save.path)
    dtype = config_util.parse_precision(config.train.precision)   # the reference ignores save.precision (quirk D.7)
    for i in range(config.train.iterations):
        k, pi = samp.next()
        s, pair = pairs[pi]
        noise = samp.noise((s.batch_size, 4, hw, hw)).to(dev)
        loss = tr.iteration(pair, k, noise)
        if rank == 0 and (i % 10 == 0 or config.logging.verbose):
            print(f"it {i} k={k} Loss*1k: {loss.item() * 1000:.4f}")
        if rank == 0 and i % config.save.per_steps == 0 and i != 0 and i != config.train.iterations - 1:
            save_path.mkdir(parents=True, exist_ok=True)
            torch.save(store.state_dict(dtype), save_path / f"{config.save.name}_{i}steps.pt")
    if rank == 0:
        save_path.mkdir(parents=True, exist_ok=True)
        torch.save(store.state_dict(dtype), save_path / f"{config.save.name}_last.pt")
        print("Done.")


def main(xl: bool, argv=None):
    args = build_parser(xl).parse_args(argv)
    config = config_util.load_config_from_yaml(args.config_file)
    if args.name is not None:
        config.save.name = args.name
    attributes = []
    if args.attributes is not None:
        attributes = [a.strip() for a in args.attributes.split(",")]
    config.network.alpha = args.alpha if args.alpha is not None else config.network.alpha
    config.network.rank = args.rank
    config.save.name += f"_alpha{config.network.alpha}"
    config.save.name += f"_rank{config.network.rank}"
    config.save.name += f"_{config.network.training_method}"
    config.save.path += f"/{config.save.name}"
    if args.prompts_file is not None:
        config.prompts_file = args.prompts_file
    prompts = prompt_util.load_prompts_from_yaml(config.prompts_file, attributes)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.distributed.init_process_group("nccl")
        args.device = int(os.environ.get("LOCAL_RANK", "0"))
    train(config, prompts, args.device, xl, args.synthetic, args.seed)
