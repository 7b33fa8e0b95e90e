"""Time the pieces of one training iteration (denoise pass, frozen B=3 pass, train forward, backward) on SDXL 1024^2."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.random_init import random_state_dict
from sliders_amd.unet import UNetEngine

dev = torch.device("cuda:0")
MODEL = sys.argv[sys.argv.index("--model") + 1] if "--model" in sys.argv else "sdxl"
cfg = CONFIGS[MODEL]()
eng = UNetEngine(cfg, random_state_dict(cfg, dev, 0), dev)
store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
store.params.add_(0.01)
eng.attach_lora(store)
eng.set_lora(True, 1.0)
hw = int(sys.argv[sys.argv.index("--hw") + 1]) if "--hw" in sys.argv else 128
s = torch.cuda.current_stream().cuda_stream


def inputs(B):
    return (torch.randn(B, 4, hw, hw, device=dev), torch.randn(B, 77, cfg.cross_attention_dim, device=dev),
            {"text_embeds": torch.randn(B, cfg.pooled_dim, device=dev),
             "time_ids": torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]] * B, device=dev)} if cfg.is_xl else None)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


x, c, kw = inputs(2)
eng(x, torch.tensor(500), c, kw, mode="on")
p_on = eng.plan(2, hw, hw, "on")
print(f"denoise pass (B=2, adapters on): {timeit(lambda: p_on.prog.run(s)):.2f} ms")
eng.set_lora(False)
x3, c3, kw3 = inputs(3)
eng(x3, torch.tensor(500), c3, kw3, mode="off")
p3 = eng.plan(3, hw, hw, "off")
print(f"frozen pass (B=3, adapters off): {timeit(lambda: p3.prog.run(s)):.2f} ms")
eng.set_lora(True, 1.0)
eng(x, torch.tensor(500), c, kw, mode="train")
p_tr = eng.plan(2, hw, hw, "train")
print(f"train forward (B=2, tape kept):  {timeit(lambda: p_tr.prog.run(s)):.2f} ms")
eng.run_backward(d_eps=torch.randn(1, 4, hw, hw, device=dev) * 1e-3)
print(f"backward (1 sample):             {timeit(lambda: p_tr.backward.prog.run(s)):.2f} ms  ({p_tr.backward.prog.n_ops} launches)")


def breakdown(prog, title):
    """Per entry-point totals of one replay, HIP events around every op (adds ~1 us per op)."""
    from collections import defaultdict
    from sliders_amd import lib
    stream = torch.cuda.current_stream()
    recs = []
    for op, d in prog.ops:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        if op in lib._ENTRY:
            lib.call(op, d, s)
        else:
            one = lib.Program(); one.add(op, d); one.run(s)
        e1.record(stream)
        recs.append((op, e0, e1))
    torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0])
    for op, e0, e1 in recs:
        k = lib._ENTRY[op][0] if op in lib._ENTRY else "memset"
        agg[k][0] += 1
        agg[k][1] += e0.elapsed_time(e1)
    print(f"-- {title}: {sum(v[1] for v in agg.values()):.2f} ms summed over {len(recs)} ops")
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:24s} x{n:5d} {ms:8.3f} ms  ({ms / n * 1e3:7.1f} us each)")


if "--breakdown" in sys.argv:
    breakdown(p_tr.backward.prog, "backward")
    breakdown(p_on.prog, "denoise pass")
