"""Development probe: does running the CFG pair as two B=1 command buffers on two HIP streams (prologues / tails of one
chain overlapping the main loops of the other) beat the single B=2 command buffer?"""
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
cfg = CONFIGS["sdxl"]()
hw = 128


def make(B):
    eng = UNetEngine(cfg, random_state_dict(cfg, dev, 0), dev)
    store = LoraStore(cfg, device=dev)
    store.params.add_(0.01)
    eng.attach_lora(store)
    eng.set_lora(True, 1.0)
    x = torch.randn(B, 4, hw, hw, device=dev)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, device=dev)
    kw = {"text_embeds": torch.randn(B, cfg.pooled_dim, device=dev),
          "time_ids": torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]] * B, device=dev)}
    eng(x, torch.tensor(500), ctx, kw, mode="on")
    return eng, eng.plan(B, hw, hw, "on")


e2, p2 = make(2)
ea, pa = make(1)
eb, pb = make(1)
torch.cuda.synchronize()
s0 = torch.cuda.current_stream().cuda_stream
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters * 1e3


print(f"B=2 one stream          : {timeit(lambda: p2.prog.run(s0)):.2f} ms", flush=True)
print(f"B=1 one stream          : {timeit(lambda: pa.prog.run(s0)):.2f} ms", flush=True)
print(f"B=1 twice, one stream   : {timeit(lambda: (pa.prog.run(s0), pb.prog.run(s0))):.2f} ms", flush=True)
print(f"B=1 + B=1, two streams  : {timeit(lambda: (pa.prog.run(sa.cuda_stream), pb.prog.run(sb.cuda_stream))):.2f} ms", flush=True)
print(f"B=2 + B=2, two streams  : {timeit(lambda: (p2.prog.run(sa.cuda_stream), p2.prog.run(sb.cuda_stream))) / 2:.2f} ms per pass "
      f"(same buffers: timing only)", flush=True)
