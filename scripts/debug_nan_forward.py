"""Debug aid: how often does a fresh engine's first training-mode forward (SDXL, 64x64 latents) come out non-finite?"""
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
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
bad = 0
N = int(os.environ.get("NAN_TRIALS", "8"))
for attempt in range(N):
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
    outs = []
    p = None
    for rep in range(4):
        out = eng(x, torch.tensor(600), ctx, kw, mode=mode).sample
        torch.cuda.synchronize()
        fin = bool(torch.isfinite(out.float()).all())
        outs.append(fin)
        if not fin:
            p = eng.plan(2, hw, hw, mode)
            base, buf = eng.arena.base, eng.arena.buf

            def view(ptr, rows, ld, cols, dt=torch.bfloat16):
                if not (base <= ptr < base + buf.numel()):
                    return None
                es = 2 if dt == torch.bfloat16 else 4
                return buf[ptr - base: ptr - base + rows * ld * es].view(dt).view(rows, ld)[:, :cols]

            # where are the non-finite values of the batched V^T array, and does re-running the producers clean it?
            for i, ((op, d), nm) in enumerate(zip(p.prog.ops, p.prog.op_names)):
                if nm == "attn2_vt_all":
                    Dp = 64
                    t = view(d.dst, d.B * d.H * Dp, d.ldt, d.ldt).float().view(d.B, d.H, Dp, d.ldt)
                    idx = (~torch.isfinite(t)).nonzero()
                    print("   vt_all non-finite count", idx.shape[0], "max |v|", float(t.abs().max()), flush=True)
                    s_ = torch.cuda.current_stream().cuda_stream
                    for j in range(0, i + 1):
                        o2, d2 = p.prog.ops[j]
                        if o2 in lib._ENTRY:
                            lib.call(o2, d2, s_)
                        else:
                            one = lib.Program(); one.add(o2, d2); one.run(s_)
                    torch.cuda.synchronize()
                    t2 = view(d.dst, d.B * d.H * Dp, d.ldt, d.ldt).float()
                    print("   after re-running ops 0..%d: non-finite in vt_all: %d" % (i, int((~torch.isfinite(t2)).sum())), flush=True)
                    # which later op's output range overlaps the V^T array?
                    lo, hi = d.dst, d.dst + d.B * d.H * Dp * d.ldt * 2
                    for j in range(i + 1, len(p.prog.ops)):
                        o2, d2 = p.prog.ops[j]
                        for fld in ("c", "y", "o", "out", "dst", "lse", "lora_t_out", "mean_rstd", "vt_out"):
                            ptr = getattr(d2, fld, None)
                            if isinstance(ptr, int) and lo <= ptr < hi:
                                print("   op", j, p.prog.op_names[j], "field", fld, "points INTO the V^T array at +", ptr - lo, flush=True)
                    break
            shown = 0
            for i, ((op, d), nm) in enumerate(zip(p.prog.ops, p.prog.op_names)):
                t = None
                if op == lib.OP_GEMM:
                    t = view(d.c, d.M, d.ldc, d.N // 2 if d.geglu else d.N)
                elif op == lib.OP_ATTN_FWD:
                    t = view(d.o, d.B * d.Tq, d.ldo, d.H * (d.D or 64))
                elif op == lib.OP_LAYERNORM:
                    t = view(d.y, d.M, d.ldy, d.C)
                elif op == lib.OP_TRANSPOSE_HEADS:
                    t = view(d.dst, d.B * d.H * (((d.D or 64) + 63) // 64 * 64), d.ldt, d.ldt)
                if t is not None:
                    nb_ = int((~torch.isfinite(t.float())).sum())
                    if nb_:
                        print(f"   replay {rep}: op {i} {nm} ({lib._ENTRY[op][0]}) output: {nb_} non-finite of {t.numel()}"
                              + (f" tile {d.tile:x} M{d.M} N{d.N} K{d.K}" if op == lib.OP_GEMM else ""), flush=True)
                        shown += 1
                        if shown >= 5:
                            break
            break
    if not all(outs):
        bad += 1
        print(f"engine {attempt}: finite per replay {outs}", flush=True)
    del eng
    torch.cuda.empty_cache()
print(f"mode {mode}: {bad}/{N} engines produced a non-finite forward", flush=True)
