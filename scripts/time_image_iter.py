"""Pieces of one image-slider iteration at SDXL 512x512 (development aid): VAE encode, training forward, backward."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.random_init import random_state_dict
from sliders_amd.unet import UNetEngine
from sliders_amd.vae import VAE_SCALING, VaeEncoder, random_vae_state_dict

dev = torch.device("cuda:0")
cfg = CONFIGS["sdxl"]()
eng = UNetEngine(cfg, random_state_dict(cfg, dev, 0), dev)
store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
store.params.add_(0.01)
eng.attach_lora(store)
eng.set_lora(True, 1.0)
hw = 64
s = torch.cuda.current_stream().cuda_stream
vae = VaeEncoder(random_vae_state_dict(device=dev, seed=0), dev, VAE_SCALING["sdxl"])
img = VaeEncoder.preprocess(torch.randint(0, 256, (512, 512, 3), dtype=torch.uint8)).to(dev)
post, noise = torch.randn(1, 4, hw, hw, device=dev), torch.randn(1, 4, hw, hw, device=dev)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


print(f"VAE encode + sample + add_noise (fp32, 512x512): {timeit(lambda: vae.get_noisy_image(img, post, noise, 0.8, 0.6)):.2f} ms")
x = torch.randn(2, 4, hw, hw, device=dev)
c = torch.randn(2, 77, cfg.cross_attention_dim, device=dev)
kw = {"text_embeds": torch.randn(2, cfg.pooled_dim, device=dev),
      "time_ids": torch.tensor([[512.0, 512.0, 0, 0, 512.0, 512.0]] * 2, device=dev)}
eng(x, torch.tensor(500), c, kw, mode="train")
p = eng.plan(2, hw, hw, "train")
print(f"train forward (B=2, 64x64 latents): {timeit(lambda: p.prog.run(s)):.2f} ms ({p.prog.n_ops} launches)")
eng.run_backward(d_eps=torch.randn(1, 4, hw, hw, device=dev) * 1e-3)
print(f"backward (1 sample): {timeit(lambda: p.backward.prog.run(s)):.2f} ms ({p.backward.prog.n_ops} launches)")
if "--breakdown" in sys.argv:
    from collections import defaultdict
    from sliders_amd import lib
    stream = torch.cuda.current_stream()
    for title, prog in (("vae encoder", vae.plan(1, 512, 512).prog), ("backward", p.backward.prog)):
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
            kname = lib._ENTRY[op][0] if op in lib._ENTRY else "memset"
            agg[kname][0] += 1
            agg[kname][1] += e0.elapsed_time(e1)
        print(f"-- {title}: {sum(v[1] for v in agg.values()):.2f} ms over {len(recs)} ops")
        for kname, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"   {kname:24s} x{n:5d} {ms:8.3f} ms")
        if title == "vae encoder":
            big = sorted(((e0.elapsed_time(e1), nm, d) for (op, e0, e1), nm, (_, d) in zip(recs, prog.op_names, prog.ops)
                          if op == lib.OP_SGEMM), key=lambda t: -t[0])[:12]
            for ms, nm, d in big:
                print(f"   {nm:44s} M{d.M} N{d.N} K{d.K}  {ms * 1e3:7.1f} us  {2.0 * d.M * d.N * d.K / ms * 1e-9:6.1f} TF/s")
