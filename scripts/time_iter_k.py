"""Per-denoise-step cost inside the real training iteration (development aid): iteration time at k = 1, 9, 17, 33 -> slope (ms per
step) against the replayed pass alone, and the k-independent part (frozen B=3 pass || training forward, backward, optimizer)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.random_init import random_state_dict
from sliders_amd.trainer import PairEmbeds, SliderTrainer
from sliders_amd.unet import UNetEngine

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="sdxl")
ap.add_argument("--hw", type=int, default=128)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = CONFIGS[args.model]()
hw = args.hw
eng = UNetEngine(cfg, random_state_dict(cfg, dev, 0), dev)
store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
tr = SliderTrainer(eng, store, hw, hw, batch_size=1, lr=2e-4)
g = torch.Generator(device="cpu").manual_seed(1234)
e = [torch.randn(1, 77, cfg.cross_attention_dim, generator=g).to(dev, torch.bfloat16) for _ in range(4)]
pl = [torch.randn(1, cfg.pooled_dim, generator=g).to(dev, torch.bfloat16) if cfg.is_xl else None for _ in range(4)]
cat = lambda x: torch.cat([e[3], x]).contiguous()
pcat = lambda x: torch.cat([pl[3], x]).contiguous() if cfg.is_xl else None
pair = PairEmbeds(cat(e[0]), cat(e[1]), cat(e[2]), cat(e[3]), pcat(pl[0]), pcat(pl[1]), pcat(pl[2]), pcat(pl[3]), guidance_scale=4.0, action="enhance")
noise = torch.randn(1, 4, hw, hw, device=dev)
for k in (3, 3):
    tr.iteration(pair, k, noise)
torch.cuda.synchronize()
for step_graphs in (False, True, False, True):
    tr.step_graphs = step_graphs
    res = {}
    tr.iteration(pair, 3, noise)
    for k in (1, 9, 17, 33, 1, 9, 17, 33):
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(2):
            tr.iteration(pair, k, noise)
        torch.cuda.synchronize()
        res.setdefault(k, []).append((time.time() - t0) / 2 * 1e3)
    print("one program per step (pass + combine + DDIM update)" if step_graphs else "fill + pass + combine launch per step")
    for k, v in res.items():
        print(f"  k = {k:2d}: iteration {min(v):7.2f} ms")
    slope = (min(res[33]) - min(res[1])) / 32
    print(f"  per denoise step inside the iteration: {slope:.3f} ms; k-independent part: {min(res[1]) - slope:.2f} ms")
# where the k-dependent time sits: phase stamps on the main stream (the frozen pass runs beside the training forward on a side stream)
for k in (1, 33, 1, 33):
    tr.phase_events = []
    torch.cuda.synchronize()
    tr.iteration(pair, k, noise)
    torch.cuda.synchronize()
    ev = tr.phase_events
    tr.phase_events = None
    print(f"k = {k:2d}: " + "  ".join(f"{b[0]} {a[1].elapsed_time(b[1]):7.2f} ms" for a, b in zip(ev, ev[1:])))
tr.phase_events, tr.phase_steps = [], True
torch.cuda.synchronize()
tr.iteration(pair, 33, noise)
torch.cuda.synchronize()
ev = tr.phase_events
tr.phase_events, tr.phase_steps = None, False
print("k = 33, per step: " + " ".join(f"{a[1].elapsed_time(b[1]):.2f}" for a, b in zip(ev, ev[1:]) if b[0].startswith("step")))
p = eng.plan(2, hw, hw, "on")
s = torch.cuda.current_stream().cuda_stream
# the denoise loop alone (per-step programs, latents evolving as in the iteration) - separates the loop's own cost per step from
# whatever a longer loop does to the passes behind it
eng.set_lora(True, 1.0)
sp = tr._step_programs(p)
sp.capture_all()
for n in (33, 33):
    tr._load_latents(p, noise.to(torch.bfloat16))
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(n):
        sp.program(i).run(s)
    torch.cuda.synchronize()
    print(f"denoise loop alone, {n} per-step programs: {(time.time() - t0) / n * 1e3:.3f} ms per step")
# the same loop right behind a training iteration, no synchronisation in between (what the next iteration's loop sees)
for rep in range(2):
    tr.iteration(pair, 1, noise)
    eng.set_lora(True, 1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for i in range(33):
        sp.program(i).run(s)
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    print(f"the same loop right behind an iteration: {e0.elapsed_time(e1) / 33:.3f} ms per step")
# ... and with the iteration's own host work in front of every step (none is expected to matter: the host runs ahead)
for rep in range(2):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for i in range(33):
        sp.program(i).run(s)
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    print(f"loop alone, event-timed: {e0.elapsed_time(e1) / 33:.3f} ms per step")
for prog, nm in ((p.prog, "full program"), (p.prog_text_cached, "text K/V cached")):
    if prog is None:
        continue
    for _ in range(3):
        prog.run(s)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        prog.run(s)
    torch.cuda.synchronize()
    print(f"replayed pass alone ({nm}): {(time.time() - t0) / 10 * 1e3:.3f} ms")
    t0 = time.time()
    for _ in range(100):       # as long as a k = 100 denoise loop: the clocks settle where a training iteration runs
        prog.run(s)
    torch.cuda.synchronize()
    print(f"replayed pass alone ({nm}), 100 replays back to back: {(time.time() - t0) / 100 * 1e3:.3f} ms")
