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
import ast
import os
import zlib
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
    p.add_argument("--rank", type=int, required=False, help="Rank of LoRA.", default=None)
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
        g = torch.Generator().manual_seed(seed + (zlib.crc32(text.encode("utf-8")) & 0xFFFFFF))   # stable across processes / ranks
        e = torch.randn(1, 77, cfg.cross_attention_dim, generator=g)
        return e, (torch.randn(1, cfg.pooled_dim, generator=g) if cfg.is_xl else None)
    return build_pairs(cfg, prompts, encode, dev)


def _encoded_pairs(cfg, prompts, name_or_path, dev, dtype, v2: bool = False):
    """Real checkpoints: CLIP text encoder(s) through `transformers`, run once before the loop (not on the hot path)."""
    from . import model_util
    if cfg.is_xl:
        toks, encs = model_util.load_text_encoders_xl(name_or_path, dev, dtype)
        encode = lambda text: model_util.encode_prompts_xl(toks, encs, [text])
    else:
        tok, enc = model_util.load_text_encoder(name_or_path, dev, dtype, v2=v2)
        encode = lambda text: (model_util.encode_prompts(tok, enc, [text]), None)
    pairs = build_pairs(cfg, prompts, encode, dev)
    torch.cuda.empty_cache()                    # the encoders are dropped here, as the reference does (train_lora_xl.py:153-156)
    return pairs


class LrSchedule:
    """Per-iteration learning rate of `train.lr_scheduler` (train_util.py:376-404, stepped once per iteration at
    train_lora_xl.py:347), evaluated on the host with torch's own scheduler classes over a dummy optimizer; the fused
    AdamW kernel takes the value as an argument of each launch."""

    def __init__(self, name: str, lr: float, iterations: int):
        from .train_util import get_lr_scheduler
        if name == "linear":
            raise NotImplementedError("lr_scheduler 'linear': the reference passes factor=0.5 to torch's LinearLR, which "
                                      "has no such argument (train_util.py:397-400 raises TypeError); pick another one")
        self._opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=lr)
        self._sched = get_lr_scheduler(name, self._opt, max_iterations=iterations, lr_min=lr / 100)

    def current(self) -> float:
        return float(self._opt.param_groups[0]["lr"])

    def step(self):
        self._opt.step()
        self._sched.step()


def optimizer_options(train_cfg) -> dict:
    """`train.optimizer` + `train.optimizer_args` ("k=v k=v", train_lora_xl.py:92-101) -> arguments of the fused flat
    optimizer kernels (slh_adamw for adam / adamw, slh_lion for lion).  Anything those kernels do not implement is an error,
    never silently ignored."""
    name = (train_cfg.optimizer or "adamw").lower()
    if name not in ("adamw", "adam", "lion", "prodigy", "dadaptadam", "dadaptlion"):
        raise NotImplementedError(f"train.optimizer '{train_cfg.optimizer}': implemented are adam / adamw / lion (fused kernels) and "
                                  f"prodigy / dadaptadam / dadaptlion (sliders_amd.optim); *8bit belongs to bitsandbytes (CUDA-only, "
                                  f"not in this image)")
    kw = {}
    if train_cfg.optimizer_args:
        for arg in train_cfg.optimizer_args.split(" "):
            if not arg:
                continue
            key, value = arg.split("=")
            kw[key] = ast.literal_eval(value)
    if name == "prodigy":   # prodigyopt.Prodigy's own arguments and defaults (requirements.txt: prodigyopt==1.0)
        out = {"betas": (0.9, 0.999), "beta3": None, "eps": 1e-8, "weight_decay": 0.0, "decouple": True,
               "use_bias_correction": False, "safeguard_warmup": False, "d0": 1e-6, "d_coef": 1.0, "growth_rate": float("inf")}
    elif name == "dadaptadam":   # dadaptation.DAdaptAdam's own arguments and defaults (requirements.txt: dadaptation==3.1)
        out = {"betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0.0, "decouple": False, "use_bias_correction": False,
               "d0": 1e-6, "growth_rate": float("inf"), "log_every": 0}
    elif name == "dadaptlion":   # dadaptation.DAdaptLion
        out = {"betas": (0.9, 0.999), "weight_decay": 0.0, "d0": 1e-6, "log_every": 0}
    elif name == "lion":    # lion_pytorch.Lion defaults (requirements.txt:5)
        out = {"betas": (0.9, 0.99), "weight_decay": 0.0}
    else:
        out = {"betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0.01 if name == "adamw" else 0.0}
    for key, value in kw.items():
        if key not in out:
            raise NotImplementedError(f"train.optimizer_args '{key}' is not implemented by the fused {name} kernel")
        out[key] = value
    out.setdefault("eps", 0.0)
    out["name"] = name
    if name == "adam" and out["weight_decay"] != 0:
        raise NotImplementedError("adam with weight_decay != 0 is L2-coupled decay; only decoupled (adamw) decay is implemented")
    out["betas"] = (float(out["betas"][0]), float(out["betas"][1]))
    return out


def check_supported(config: config_util.RootConfig, image_slider: bool = False):
    """Reject, before any model is loaded, every config value the fused path would otherwise have to ignore."""
    t = config.train
    prec = config_util.parse_precision(t.precision)
    if prec not in (torch.bfloat16, torch.float32):
        raise NotImplementedError(f"train.precision '{t.precision}': bfloat16 (the reference's default) and float32 are implemented; "
                                  f"the MFMA path has no fp16 arithmetic")
    if prec == torch.float32 and t.optimizer.lower() not in ("adam", "adamw"):
        raise NotImplementedError(f"train.precision float32 with optimizer '{t.optimizer}': the fp32 adapter state is kept by the fused "
                                  f"adam / adamw kernel only")
    name = t.noise_scheduler.lower().replace(" ", "_")
    if name not in ("ddim", "ddpm", "lms", "euler_a"):            # model_util.py:230-277
        raise ValueError(f"Unknown scheduler name: {name}")
    if image_slider and name not in ("ddim", "ddpm"):
        # the image-slider loop never calls scheduler.step (imagesliders/train_util.py:200-235 uses .timesteps and .add_noise
        # only): "ddpm" has the same leading timestep grid and the same add_noise as "ddim", so both run the fused
        # alpha-table noising step; lms / euler_a noise in sigma space (x + sigma * noise, scaled model input)
        raise NotImplementedError(f"train.noise_scheduler '{t.noise_scheduler}': the image-slider noising step (get_noisy_image, "
                                  f"fused with the VAE encode) is built on the ddim / ddpm alpha table; text sliders also accept lms / euler_a")
    optimizer_options(t)
    LrSchedule(t.lr_scheduler, t.lr, t.iterations)


def adapter_state_dtype(config: config_util.RootConfig, rank: int = 0) -> torch.dtype:
    """train.precision as the reference applies it to the LoRA network (train_lora_xl.py:60-61, 84-90: weight_dtype): the dtype of the
    adapter parameters, of the optimizer's moments and of the saved checkpoint.  With float32 the reference ALSO runs the frozen UNet in
    fp32; here the frozen UNet and the activations stay bf16 on the MFMA path (said once at start-up) - what float32 buys is what it
    buys in the reference's mixed-precision practice: updates of 2e-4 x O(1) no longer vanish below the bf16 spacing of the weights."""
    dt = config_util.parse_precision(config.train.precision)
    if dt == torch.float32 and rank == 0:
        print("train.precision float32: adapter parameters, AdamW moments and the checkpoint are fp32 (fused optimizer on an fp32 master); "
              "the frozen UNet and the activations are computed in bf16 on the MFMA path - narrower than the reference's all-fp32 run")
    return dt


def check_model_files(name_or_path: str):
    """Single-file checkpoints carry the text encoders in an LDM layout this loader does not convert: fail before the
    UNet is loaded and repacked, and say what works."""
    if os.path.isfile(name_or_path):
        raise NotImplementedError(
            f"{name_or_path}: single-file checkpoints are accepted for the UNet only (sliders_amd.model_util."
            f"load_unet_engine); the training CLI also needs the tokenizer and text encoder(s), which it loads from a "
            f"diffusers-format model directory - pass that directory as pretrained_model.name_or_path")


def train(config: config_util.RootConfig, prompts, device: int, xl: bool, synthetic: bool, seed: int = 0):
    from .train_util import get_add_time_ids, get_random_resolution_in_bucket
    check_supported(config)
    if not synthetic:
        check_model_files(config.pretrained_model.name_or_path)
    rank, world = world_info()
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    if config.network.type != "c3lier":
        print("note: network.type lierla -> attention-only targets")
    if synthetic:
        eng = synthetic_engine("sdxl" if xl else ("sd2" if config.pretrained_model.v2 else "sd1"), dev, seed)
    else:
        eng = load_unet_engine(config.pretrained_model.name_or_path, dev)
        if eng.cfg.is_xl != xl:
            raise ValueError(f"{config.pretrained_model.name_or_path} is {'an SDXL' if eng.cfg.is_xl else 'an SD-1.x / 2.x'} "
                             f"UNet but the {'XL' if xl else 'SD-1.x / 2.x'} entry point was used")
        if not xl and (eng.cfg.cross_attention_dim == 1024) != bool(config.pretrained_model.v2):
            raise ValueError(f"pretrained_model.v2 = {config.pretrained_model.v2} but the UNet's context width is "
                             f"{eng.cfg.cross_attention_dim} (SD-2.x: 1024, SD-1.x: 768)")
    torch.manual_seed(seed)
    store = LoraStore(eng.cfg, rank=config.network.rank, alpha=config.network.alpha,
                      train_method=config.network.training_method, network_type=config.network.type, device=dev,
                      state_dtype=adapter_state_dtype(config, rank))
    opt = optimizer_options(config.train)
    hw0 = prompts[0].resolution // 8
    tr = SliderTrainer(eng, store, hw0, hw0, batch_size=prompts[0].batch_size, lr=config.train.lr, betas=opt["betas"],
                       eps=opt["eps"], weight_decay=opt["weight_decay"],
                       max_denoising_steps=config.train.max_denoising_steps,
                       process_group=torch.distributed.group.WORLD if world > 1 else None,
                       prediction_type="v_prediction" if config.pretrained_model.v_pred else "epsilon",
                       optimizer=opt["name"], noise_scheduler=config.train.noise_scheduler,
                       scheduler_seed=seed * 7919 + rank,
                       optimizer_kwargs={k: v for k, v in opt.items() if k not in ("name", "betas", "eps", "weight_decay")})
    if synthetic:
        pairs = _synthetic_pairs(eng.cfg, prompts, dev, seed)
    else:
        pairs = _encoded_pairs(eng.cfg, prompts, config.pretrained_model.name_or_path, dev,
                               config_util.parse_precision(config.train.precision), v2=config.pretrained_model.v2)
    samp = StepSampler(seed, rank, world, len(pairs), config.train.max_denoising_steps)
    sched = LrSchedule(config.train.lr_scheduler, config.train.lr, config.train.iterations)
    wandb = None
    if config.logging.use_wandb and rank == 0:     # optional, as in train_lora_xl.py:57-58
        try:
            import wandb
            wandb.init(project=f"LECO_{config.save.name}", config=config.model_dump())
        except ImportError:
            print("logging.use_wandb: the wandb package is not installed; continuing without it")
            wandb = None
    save_path = Path(config.save.path)
    dtype = config_util.parse_precision(config.train.precision)   # the reference ignores save.precision (quirk D.7)
    for i in range(config.train.iterations):
        k, pi = samp.next()
        s, pair = pairs[pi]
        # per-pair resolution / dynamic_resolution / batch_size / dynamic_crops (train_lora_xl.py:179-203); the draws
        # come from the rank-shared stream so that every rank does the same amount of work in a step
        height, width = samp.resolution(s)
        time_ids = samp.time_ids(s, height, width, eng.cfg.is_xl)
        # get_initial_latents (train_util.py:55): unit noise times the scheduler's init_noise_sigma
        noise = samp.noise((s.batch_size, 4, height // 8, width // 8)).to(dev) * tr.sched.init_noise_sigma
        lr = sched.current()
        loss = tr.iteration(pair, k, noise, lr=lr, time_ids=time_ids)
        sched.step()
        if rank == 0 and (i % 10 == 0 or config.logging.verbose):
            print(f"it {i} k={k} {height}x{width} lr={lr:.3e} Loss*1k: {loss.item() * 1000:.4f}")
        if wandb is not None:
            wandb.log({"loss": loss.item(), "iteration": i, "lr": lr})
        if rank == 0 and i % config.save.per_steps == 0 and i != 0 and i != config.train.iterations - 1:
            save_path.mkdir(parents=True, exist_ok=True)
            torch.save(store.state_dict(dtype), save_path / f"{config.save.name}_{i}steps.pt")
    if rank == 0:
        save_path.mkdir(parents=True, exist_ok=True)
        torch.save(store.state_dict(dtype), save_path / f"{config.save.name}_last.pt")
        print("Done.")


def apply_cli_overrides(config: config_util.RootConfig, args):
    """CLI flags override the YAML only when given (train_lora_xl.py:394-410)."""
    if args.name is not None:
        config.save.name = args.name
    if args.alpha is not None:
        config.network.alpha = args.alpha
    if args.rank is not None:
        config.network.rank = args.rank
    config.save.name += f"_alpha{config.network.alpha}"
    config.save.name += f"_rank{config.network.rank}"
    config.save.name += f"_{config.network.training_method}"
    config.save.path += f"/{config.save.name}"
    if args.prompts_file is not None:
        config.prompts_file = args.prompts_file
    return config


def main(xl: bool, argv=None):
    args = build_parser(xl).parse_args(argv)
    config = apply_cli_overrides(config_util.load_config_from_yaml(args.config_file), args)
    attributes = []
    if args.attributes is not None:
        attributes = [a.strip() for a in args.attributes.split(",")]
    prompts = prompt_util.load_prompts_from_yaml(config.prompts_file, attributes)
    check_supported(config)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.distributed.init_process_group("nccl")
        args.device = int(os.environ.get("LOCAL_RANK", "0"))
    train(config, prompts, args.device, xl, args.synthetic, args.seed)
