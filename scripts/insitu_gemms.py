"""Per-shape in-situ vs warm timing of every slh_gemm of one LoRA-on pass (development aid): where do cold weights and
under-filled grids cost the most?"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd import lib
from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.random_init import random_state_dict
from sliders_amd.unet import UNetEngine

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="sdxl")
ap.add_argument("--hw", type=int, default=128)
ap.add_argument("--attn", action="store_true", help="also list the attention launches by shape")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = CONFIGS[args.model]()
eng = UNetEngine(cfg, random_state_dict(cfg, dev, 0), dev)
store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
store.params.add_(0.01)
eng.attach_lora(store)
eng.set_lora(True, 1.0)
hw, B = args.hw, 2
x = torch.randn(B, 4, hw, hw, device=dev)
ctx = torch.randn(B, 77, cfg.cross_attention_dim, device=dev)
kw = {"text_embeds": torch.randn(B, cfg.pooled_dim, device=dev),
      "time_ids": torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]] * B, device=dev)} if cfg.is_xl else None
eng(x, torch.tensor(500), ctx, kw, mode="on")
p = eng.plan(B, hw, hw, "on")
stream = torch.cuda.current_stream()
s = stream.cuda_stream


def launch(op, d):
    if op in lib._ENTRY:
        lib.call(op, d, s)
    else:
        one = lib.Program(); one.add(op, d); one.run(s)


for _ in range(2):
    p.prog.run(s)
torch.cuda.synchronize()
recs = []
for (op, d), nm in zip(p.prog.ops, p.prog.op_names):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); launch(op, d); e1.record(stream)
    recs.append((op, d, nm, e0, e1))
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0, 0.0, ""])
other = defaultdict(lambda: [0, 0.0])
for op, d, nm, e0, e1 in recs:
    us = e0.elapsed_time(e1) * 1e3
    if op != lib.OP_GEMM:
        k = lib._ENTRY[op][0] if op in lib._ENTRY else "memset"
        if args.attn and op == lib.OP_ATTN_FWD:
            k = f"slh_attn_fwd B{d.B} H{d.H} Tq{d.Tq} Tk{d.Tk} D{d.D}"
        other[k][0] += 1; other[k][1] += us
        continue
    for _ in range(2):
        lib.call(op, d, s)
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record(stream)
    for _ in range(3):
        lib.call(op, d, s)
    w1.record(stream); w1.synchronize()
    warm = w0.elapsed_time(w1) * 1e3 / 3
    key = (d.M, d.N, d.K, d.mode, bool(d.lora_down), d.geglu, hex(d.tile))
    a = agg[key]
    a[0] += 1; a[1] += us; a[2] += warm; a[3] = nm
tot = sum(a[1] for a in agg.values())
print(f"GEMM in situ total {tot / 1e3:.2f} ms, warm total {sum(a[2] for a in agg.values()) / 1e3:.2f} ms")
print(f"{'M':>6} {'N':>7} {'K':>6} m lora gg tile   n   insitu_us  warm_us   TF/s(insitu)  total_ms  example")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K = k[0], k[1], k[2]
    fl = 2.0 * M * N * K
    print(f"{M:6d} {N:7d} {K:6d} {k[3]} {int(k[4])}    {k[5]}  {k[6]:>6} {a[0]:3d}  {a[1] / a[0]:9.1f} {a[2] / a[0]:8.1f}   {fl * a[0] / a[1] / 1e6:8.0f}   {a[1] / 1e3:8.2f}  {a[3]}")
print(f"all ops in situ: {sum(e0.elapsed_time(e1) for _, _, _, e0, e1 in recs):.2f} ms over {len(recs)} launches")
print("other ops:")
for k, v in sorted(other.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:24s} n={v[0]:4d} total {v[1] / 1e3:7.2f} ms avg {v[1] / v[0]:7.1f} us")
