"""Debug aid: build the SDXL 512x512 training plan with the batched text K/V arrays and list every op whose OUTPUT byte range
intersects a byte range some other, later-read buffer occupies (the batched K/V array, the batched V^T array), and any two
output ranges of different ops that intersect at all.  Pure descriptor analysis; nothing is launched."""
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
eng = UNetEngine(cfg, random_state_dict(cfg, dev, 0, torch.bfloat16), dev)
store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
eng.attach_lora(store)
eng.set_lora(True, 1.0)
p = eng.plan(2, hw, hw, "train")
ops, names = p.prog.ops, p.prog.op_names


def outs(op, d):
    """(field, ptr, nbytes) of everything the op writes."""
    r = []
    if op == lib.OP_GEMM:
        ncols = d.N // 2 if d.geglu else d.N
        if not d.vt_out:
            r.append(("c", d.c, ((d.M - 1) * d.ldc + ncols) * 2))
        if d.lora_t_out:
            r.append(("lora_t_out", d.lora_t_out, d.M * d.ld_t * 4))
        if d.splitk_c32 and (d.tile >> 16) & 15:
            r.append(("splitk_c32", d.splitk_c32, d.M * d.N * 4))
    elif op == lib.OP_LAYERNORM:
        r.append(("y", d.y, ((d.M - 1) * d.ldy + d.C) * 2))
        if d.mean_rstd:
            r.append(("mean_rstd", d.mean_rstd, d.M * 8))
    elif op == lib.OP_ATTN_FWD:
        r.append(("o", d.o, ((d.B * d.Tq - 1) * d.ldo + d.H * (d.D or 64)) * 2))
        if d.lse:
            r.append(("lse", d.lse, d.B * d.H * d.Tq * 4))
    elif op == lib.OP_TRANSPOSE_HEADS:
        Dp = ((d.D or 64) + 63) // 64 * 64
        r.append(("dst", d.dst, d.B * d.H * Dp * d.ldt * 2))
    elif op in (lib.OP_GN_APPLY,):
        r.append(("y", d.y, ((d.batch * d.hw - 1) * d.ldy + d.c0 + d.c1) * 2))
    elif op == lib.OP_ELEMENTWISE:
        r.append(("out", d.out, ((d.M - 1) * d.ldo + d.C) * 2))
    elif op == lib.OP_CONV_IN:
        r.append(("y", d.y, d.batch * d.h * d.wd * d.ldy * 2))
    return r


def reads(op, d):
    r = []
    if op == lib.OP_ATTN_FWD:
        D = d.D or 64
        r.append(("q", d.q, ((d.B * d.Tq - 1) * d.ldq + d.H * D) * 2))
        r.append(("k", d.k, ((d.B * d.Tk - 1) * d.ldk + d.H * D) * 2))
        vh = d.vt_batch_heads or d.H
        r.append(("vt", d.vt, (((d.B - 1) * vh + d.H) * 64 * d.ldvt) * 2))
    return r


W = []
for i, ((op, d), nm) in enumerate(zip(ops, names)):
    for f, ptr, nb in outs(op, d):
        W.append((ptr, ptr + nb, i, nm, f))
W.sort()
n_ov = 0
for a, b in zip(W, W[1:]):
    if b[0] < a[1] and a[2] != b[2]:
        n_ov += 1
        if n_ov <= 20:
            print(f"OUTPUT ranges intersect: op {a[2]} {a[3]}.{a[4]} [{a[0]:#x},{a[1]:#x}) and op {b[2]} {b[3]}.{b[4]} [{b[0]:#x},{b[1]:#x})")
print("intersecting output pairs (adjacent in address order):", n_ov)
# every cross-attention launch: which LATER-or-EARLIER written ranges (other than its producers, ops <= 10) fall into what it reads?
bad = 0
for i, ((op, d), nm) in enumerate(zip(ops, names)):
    if op != lib.OP_ATTN_FWD or not d.vt_batch_heads:
        continue
    for f, ptr, nb in reads(op, d):
        for (lo, hi, j, nm2, f2) in W:
            if j > 10 and j != i and lo < ptr + nb and ptr < hi and not (f == "q" and j == i - 1) :
                bad += 1
                if bad <= 20:
                    print(f"op {i} {nm} reads {f} [{ptr:#x},{ptr + nb:#x}) which op {j} {nm2} WRITES ({f2} [{lo:#x},{hi:#x}))")
print("read/write intersections for the batched cross-attention launches:", bad)
print("arena base", hex(eng.arena.base), "capacity", eng.arena.capacity, "high water", eng.arena.high_water)
