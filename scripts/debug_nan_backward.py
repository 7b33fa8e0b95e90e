"""Debug aid: replay the backward program of a full-width SDXL training step op by op and report the first op after which
the flat gradient buffer holds a non-finite value."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sliders_amd import lib
from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.random_init import random_state_dict
from sliders_amd.unet import UNetEngine

dev = torch.device("cuda:0")
cfg = CONFIGS["sdxl"]()
hw = 64
sd = random_state_dict(cfg, dev, 0, torch.bfloat16)
for attempt in range(4):
    store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
    g = torch.Generator().manual_seed(3)
    for e in store.entries:
        store.params[e.up_off:e.up_off + e.up_numel] = (torch.randn(e.up_numel, generator=g) * 0.02).to(dev, torch.bfloat16)
    eng = UNetEngine(cfg, sd, dev)
    eng.attach_lora(store)
    eng.set_lora(True, 1.0)
    x = torch.randn(2, 4, hw, hw, device=dev)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, device=dev)
    kw = {"text_embeds": torch.randn(2, cfg.pooled_dim, device=dev),
          "time_ids": torch.tensor([[512.0, 512.0, 0, 0, 512.0, 512.0]] * 2, device=dev)}
    out = eng(x, torch.tensor(600), ctx, kw, mode="train").sample
    print("attempt", attempt, "forward finite:", bool(torch.isfinite(out.float()).all()), flush=True)
    p = eng.plan(2, hw, hw, "train")
    bw = p.backward
    G = torch.randn(1, 4, hw, hw, device=dev)
    bw.deps_pix.tensor.copy_(G.permute(0, 2, 3, 1).reshape(-1, 4))
    store.grads.zero_()
    s = torch.cuda.current_stream().cuda_stream
    bw.prog.run(s)
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(store.grads).all())
    print("  whole-program backward: gradients finite:", ok, flush=True)
    if not ok:
        base, buf = eng.arena.base, eng.arena.buf
        zb, zbuf = eng.zarena.base, eng.zarena.buf

        def view(ptr, nbytes, dt):
            if base <= ptr < base + buf.numel():
                return buf[ptr - base: ptr - base + nbytes].view(dt)
            if zb <= ptr < zb + zbuf.numel():
                return zbuf[ptr - zb: ptr - zb + nbytes].view(dt)
            return None

        shown = 0
        for i, ((op, d), nm) in enumerate(zip(bw.prog.ops, bw.prog.op_names)):
            t = None
            if op == lib.OP_GEMM:
                t = view(d.c, d.M * d.ldc * 2, torch.bfloat16)
            elif op == lib.OP_SKINNY:
                t = view(d.out, d.M * d.ldo * 4, torch.float32)
            elif op == lib.OP_LAYERNORM_BWD:
                t = view(d.dx, d.M * d.lddx * 2, torch.bfloat16)
            elif op == lib.OP_ATTN_BWD:
                t = view(d.dq, d.B * d.Tq * d.lddq * 2, torch.bfloat16) if hasattr(d, "lddq") else None
            if t is not None and not bool(torch.isfinite(t.float()).all()):
                n_bad = int((~torch.isfinite(t.float())).sum())
                print(f"    op {i} {nm}: output has {n_bad} non-finite of {t.numel()}", flush=True)
                shown += 1
                if shown >= 6:
                    break
        # forward-side inputs the backward reads
        for (op, d), nm in zip(p.prog.ops, p.prog.op_names):
            if op == lib.OP_GEMM and d.lora_t_out:
                t = view(d.lora_t_out, d.M * d.ld_t * 4, torch.float32)
                if t is not None and not bool(torch.isfinite(t).all()):
                    print("    forward T not finite:", nm, flush=True)
                    break
        break
    del eng
    torch.cuda.empty_cache()
