"""Run-to-run determinism of one UNet pass: replays the same program twice on identical inputs, snapshots the whole
activation arena after each run and reports, in allocation (= program) order, the first buffers that differ
(development aid: separates fp32-atomics noise in the GroupNorm statistics from genuine races)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.random_init import random_state_dict
from sliders_amd.unet import UNetEngine

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="tiny_sdxl")
ap.add_argument("--hw", type=int, default=16)
ap.add_argument("--mode", default="on")
ap.add_argument("--B", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = CONFIGS[a.model]()
eng = UNetEngine(cfg, random_state_dict(cfg, dev, 0), dev)
store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
store.params.add_(0.01)
eng.attach_lora(store)
eng.set_lora(a.mode != "off", 1.0)
g = torch.Generator().manual_seed(1)
x = torch.randn(a.B, 4, a.hw, a.hw, generator=g).to(dev)
ctx = torch.randn(a.B, 77, cfg.cross_attention_dim, generator=g).to(dev)
kw = {"text_embeds": torch.randn(a.B, cfg.pooled_dim, generator=g).to(dev),
      "time_ids": torch.tensor([[a.hw * 8.0, a.hw * 8.0, 0, 0, a.hw * 8.0, a.hw * 8.0]] * a.B, device=dev)} if cfg.is_xl else None
eng(x, torch.tensor(500), ctx, kw, mode=a.mode)
p = eng.plan(a.B, a.hw, a.hw, a.mode)
s = torch.cuda.current_stream().cuda_stream
snaps = []
for r in range(3):
    p.prog.run(s)
    torch.cuda.synchronize()
    end = p.arena_end
    snaps.append(eng.arena.buf[:end].clone())
    zs = eng.zarena.buf[:eng.zarena.mark()].clone()
    snaps[-1] = (snaps[-1], zs)
for r in (1, 2):
    diff = (snaps[0][0] != snaps[r][0])
    zdiff = (snaps[0][1] != snaps[r][1])
    print(f"run {r} vs run 0: {int(diff.sum())} differing bytes of {diff.numel()} in the activation arena, "
          f"{int(zdiff.sum())} of {zdiff.numel()} in the fp32 accumulator arena")
    if diff.any():
        idx = diff.nonzero().flatten()
        shown = 0
        for (lo, hi, name) in sorted(eng.arena.allocs):
            if lo >= end:
                continue
            n = int(diff[lo:hi].sum())
            if n:
                print(f"   {name:60s} {n:9d} / {hi - lo} bytes differ")
                shown += 1
                if shown >= 12:
                    break
