"""Quick forward-only timing of the HIP UNet engine at full model size (development aid; bench.py is the
contract benchmark)."""
import argparse
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.random_init import random_state_dict
from sliders_amd.unet import UNetEngine

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="sdxl")
ap.add_argument("--hw", type=int, default=128)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--lora", action="store_true")
ap.add_argument("--warm", type=int, default=3)
ap.add_argument("--no-graph", action="store_true", help="plain launches instead of the hipGraph replay")
args = ap.parse_args()

dev = torch.device("cuda:0")
cfg = CONFIGS[args.model]()
t0 = time.time()
sd = random_state_dict(cfg, dev, 0)
eng = UNetEngine(cfg, sd, dev)
del sd
print(f"weights packed in {time.time() - t0:.1f}s, {eng.weights.nbytes() / 2**30:.2f} GiB", flush=True)
B, hw = args.batch, args.hw
x = torch.randn(B, 4, hw, hw, device=dev)
ctx = torch.randn(B, 77, cfg.cross_attention_dim, device=dev)
kw = None
if cfg.is_xl:
    kw = {"text_embeds": torch.randn(B, cfg.pooled_dim, device=dev),
          "time_ids": torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]] * B, device=dev)}
mode = "off"
if args.lora:
    store = LoraStore(cfg, device=dev)
    store.params.add_(0.01)
    eng.attach_lora(store)
    eng.set_lora(True, 1.0)
    mode = "on"
out = eng(x, torch.tensor(500), ctx, kw, mode=mode).sample
torch.cuda.synchronize()
print("eps rms", out.float().pow(2).mean().sqrt().item(), "finite", bool(torch.isfinite(out.float()).all()), flush=True)
p = eng.plan(B, hw, hw, mode)
s = torch.cuda.current_stream().cuda_stream
for _ in range(args.warm):
    p.prog.run(s, graph=not args.no_graph)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(args.iters):
    p.prog.run(s, graph=not args.no_graph)
torch.cuda.synchronize()
dt = (time.time() - t0) / args.iters
print(f"{args.model} hw={hw} B={B} mode={mode}: {dt * 1e3:.2f} ms / UNet forward ({p.prog.n_ops} launches)", flush=True)
