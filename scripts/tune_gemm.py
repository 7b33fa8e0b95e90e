"""Tile tuner for slh_gemm: times every tile variant on every distinct GEMM / implicit-conv shape that occurs in
one UNet pass (forward and backward programs) and writes sliders_amd/tuning/<name>.json, which the planner
consults (sliders_amd/tuning.py).  Development tool; run on the GPU box."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd import lib
from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.random_init import random_state_dict
from sliders_amd.tuning import gemm_key
from sliders_amd.unet import UNetEngine

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="sdxl")
ap.add_argument("--hw", type=int, default=128)
ap.add_argument("--out", default=None)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--fwd-only", action="store_true")
args = ap.parse_args()

dev = torch.device("cuda:0")
cfg = CONFIGS[args.model]()
eng = UNetEngine(cfg, random_state_dict(cfg, dev, 0), dev)
store = LoraStore(cfg, device=dev)
store.params.add_(0.01)
eng.attach_lora(store)
eng.set_lora(True, 1.0)
hw = args.hw
p = eng.plan(2, hw, hw, "train")
stream = torch.cuda.current_stream()
s = stream.cuda_stream
x = torch.randn(2, 4, hw, hw, device=dev)
ctx = torch.randn(2, 77, cfg.cross_attention_dim, device=dev)
kw = {"text_embeds": torch.randn(2, cfg.pooled_dim, device=dev),
      "time_ids": torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]] * 2, device=dev)} if cfg.is_xl else None
eng(x, torch.tensor(500), ctx, kw, mode="train")
eng.run_backward(d_eps=torch.randn(1, 4, hw, hw, device=dev) * 1e-3)
torch.cuda.synchronize()

eng(x, torch.tensor(500), ctx, kw, mode="on")      # the no-grad pass has its own shapes: fused GEGLU, batched text K/V
p_on = eng.plan(2, hw, hw, "on")
shapes = {}
for prog in ((p.prog, p_on.prog) if args.fwd_only else (p.prog, p_on.prog, p.backward.prog)):
    for opcode, d in prog.ops:
        if opcode == lib.OP_GEMM:
            shapes.setdefault(gemm_key(d), []).append(d)
print(f"{len(shapes)} distinct GEMM shapes", flush=True)
table, total_best, total_heur = {}, 0.0, 0.0
for key, ds in sorted(shapes.items(), key=lambda kv: -kv[1][0].M * kv[1][0].N * kv[1][0].K * len(kv[1])):
    d0 = ds[0]
    res = {}
    for tile in (0x22, 0x21, 0x12, 0x11, 0x4022, 0x4012, 0x4011, 0):
        if d0.geglu and (tile & 15) == 1:
            continue
        if (tile >> 12) == 4 and d0.M * d0.N < 256 * 128 * 64:
            continue    # 8-wave tiles only make sense when they still give >= 64 workgroups
        d = type(d0).from_buffer_copy(bytes(d0))
        d.tile = tile
        # never let a tuning launch corrupt live accumulation buffers: drop residual aliasing on c
        lib.call(lib.OP_GEMM, d, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.reps):
            lib.call(lib.OP_GEMM, d, s)
        e1.record(stream)
        e1.synchronize()
        res[tile] = e0.elapsed_time(e1) / args.reps * 1e3
    best = min((t for t in res if t != 0), key=lambda t: res[t])
    fl = 2.0 * d0.M * d0.N * d0.K
    table[key] = best
    total_best += res[best] * len(ds)
    total_heur += res[0] * len(ds)
    print(f"{key:44s} x{len(ds):4d}  " + "  ".join(f"{t:x}:{res[t]:.1f}" for t in res) +
          f"  best {best:x} {fl / res[best] / 1e6:6.0f} TF/s", flush=True)
print(f"sum over one fwd+bwd: heuristic {total_heur / 1e3:.2f} ms, tuned {total_best / 1e3:.2f} ms")
out = args.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sliders_amd", "tuning",
                               f"gfx950_{args.model}_{hw}.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
old = {}
if os.path.exists(out):
    with open(out) as f:
        old = json.load(f)
old.update(table)
with open(out, "w") as f:
    json.dump(old, f, indent=0, sort_keys=True)
print("wrote", out)
