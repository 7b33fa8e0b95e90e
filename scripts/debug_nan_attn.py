"""Debug aid for the open issue in docs/ROUND_NOTES.md: at the first failing cross-attention launch of a training forward with
the batched text K/V arrays, (1) relaunch that one op on the unchanged buffers, (2) recompute it with torch from the same
buffers, (3) report the operand magnitudes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLIDERS_TRAIN_KV_BATCHED"] = "1"
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
for attempt in range(10):
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
    hit = False
    for rep in range(4):
        out = eng(x, torch.tensor(600), ctx, kw, mode="train").sample
        torch.cuda.synchronize()
        if bool(torch.isfinite(out.float()).all()):
            continue
        hit = True
        p = eng.plan(2, hw, hw, "train")
        base, buf = eng.arena.base, eng.arena.buf

        def view(ptr, rows, ld, cols):
            return buf[ptr - base: ptr - base + rows * ld * 2].view(torch.bfloat16).view(rows, ld)[:, :cols]

        for i, ((op, d), nm) in enumerate(zip(p.prog.ops, p.prog.op_names)):
            if op != lib.OP_ATTN_FWD or not d.vt_batch_heads:
                continue
            D, H, B, Tq, Tk = d.D or 64, d.H, d.B, d.Tq, d.Tk
            o = view(d.o, B * Tq, d.ldo, H * D).float()
            nbad = int((~torch.isfinite(o)).sum())
            if not nbad:
                continue
            q = view(d.q, B * Tq, d.ldq, H * D).float().view(B, Tq, H, D).permute(0, 2, 1, 3)
            k = view(d.k, B * Tk, d.ldk, H * D).float().view(B, Tk, H, D).permute(0, 2, 1, 3)
            vh = d.vt_batch_heads
            vt = buf[d.vt - base: d.vt - base + ((B - 1) * vh + H) * 64 * d.ldvt * 2].view(torch.bfloat16)
            vt = torch.stack([vt[(b * vh) * 64 * d.ldvt:(b * vh + H) * 64 * d.ldvt].view(H, 64, d.ldvt) for b in range(B)]).float()
            v = vt[:, :, :D, :Tk].permute(0, 1, 3, 2)                                   # [B][H][Tk][D]
            ref = torch.softmax(q @ k.transpose(-1, -2) * d.scale, -1) @ v           # [B][H][Tq][D]
            ref = ref.permute(0, 2, 1, 3).reshape(B * Tq, H * D)
            bad_rows = (~torch.isfinite(o)).any(1).nonzero().flatten()
            print(f"engine {attempt} replay {rep}: op {i} {nm}: {nbad} non-finite outputs in rows {bad_rows[:8].tolist()}.. "
                  f"| max|q| {q.abs().max():.3g} max|k| {k.abs().max():.3g} max|v| {v.abs().max():.3g} "
                  f"| torch from the same buffers finite: {bool(torch.isfinite(ref).all())}", flush=True)
            hb = ((~torch.isfinite(o)).view(B * Tq, H, D).any(2)).nonzero()
            print("   (row, head) of the bad outputs:", hb[:10].tolist(), flush=True)
            lib.call(op, d, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            o2 = view(d.o, B * Tq, d.ldo, H * D).float()
            good = torch.isfinite(o)
            print(f"   relaunch of the same op on the same buffers: non-finite {int((~torch.isfinite(o2)).sum())}, "
                  f"rel diff to torch on the rest {float((o2 - ref).norm() / ref.norm()):.3e}", flush=True)
            break
        break
    del eng
    torch.cuda.empty_cache()
    if hit:
        break
