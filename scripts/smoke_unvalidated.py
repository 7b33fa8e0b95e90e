"""First hardware run of the paths written after round 2's GPU budget was spent (development aid, see
scripts/round3_first_call.sh): SliderTrainer with each tensor-op noise scheduler and with the Prodigy optimizer, and the
sampler with the LMS scheduler; a few iterations each on the tiny SDXL-shaped net, printing loss / finiteness."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.random_init import random_state_dict
from sliders_amd.sampler import SliderSampler
from sliders_amd.trainer import PairEmbeds, SliderTrainer
from sliders_amd.unet import UNetEngine

dev = torch.device("cuda:0")
cfg = CONFIGS["tiny_sdxl"]()
sd = random_state_dict(cfg, dev, 0)
hw = 16
g = torch.Generator().manual_seed(5)
emb = {k: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for k in ("target", "positive", "neutral", "uncond")}
pool = {k: torch.randn(1, cfg.pooled_dim, generator=g) for k in emb}
cat = lambda x: torch.cat([emb["uncond"], x]).to(dev, torch.bfloat16).contiguous()
pc = lambda x: torch.cat([pool["uncond"], x]).to(dev, torch.bfloat16).contiguous()
pair = PairEmbeds(cat(emb["target"]), cat(emb["positive"]), cat(emb["neutral"]), cat(emb["uncond"]),
                  pc(pool["target"]), pc(pool["positive"]), pc(pool["neutral"]), pc(pool["uncond"]),
                  guidance_scale=4.0, action="enhance")
for sched, opt, lr in (("ddpm", "adamw", 2e-4), ("euler_a", "adamw", 2e-4), ("lms", "adamw", 2e-4), ("ddim", "prodigy", 1.0),
                       ("euler_a", "prodigy", 1.0)):
    store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
    eng = UNetEngine(cfg, sd, dev)
    tr = SliderTrainer(eng, store, hw, hw, lr=lr, noise_scheduler=sched, optimizer=opt,
                       weight_decay=0.01 if opt == "adamw" else 0.0)
    p0 = store.params.clone()
    losses = []
    for it in range(4):
        noise = torch.randn(1, 4, hw, hw, generator=g) * tr.sched.init_noise_sigma
        losses.append(float(tr.iteration(pair, 2 + it, noise.to(dev)).item()))
    torch.cuda.synchronize()
    moved = float((store.params.float() - p0.float()).abs().max())
    extra = f" d={tr._tensor_opt.param_groups[0]['d']:.3e}" if opt == "prodigy" else ""
    print(f"{sched:8s} {opt:8s}: losses {['%.4e' % l for l in losses]} finite={all(l == l and abs(l) < 1e9 for l in losses)} "
          f"max |dparam| {moved:.3e}{extra}", flush=True)
store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
eng = UNetEngine(cfg, sd, dev)
ctx, pooled = cat(emb["target"]), pc(pool["target"])
for name in ("ddim", "lms", "euler", "euler_a", "ddpm"):
    smp = SliderSampler(eng, store, scheduler=name)
    lat = smp.sample_latents(ctx, torch.randn(1, 4, hw, hw, generator=torch.Generator().manual_seed(1)).to(dev), scale=1.0,
                             start_noise=750, ddim_steps=20, guidance_scale=7.5, pooled=pooled)
    torch.cuda.synchronize()
    print(f"sampler {name:8s}: latents rms {float(lat.float().pow(2).mean().sqrt()):.4f} finite={bool(torch.isfinite(lat.float()).all())}")
