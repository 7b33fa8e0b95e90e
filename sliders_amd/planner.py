"""Static planner: turns one UNet2DConditionModel forward (and, for the training pass, its backward through
the frozen net into the LoRA adapters) into a flat command buffer for libsliders_hip.so.

Op order follows diffusers-0.20.2 UNet2DConditionModel.forward as called by the reference at
trainscripts/textsliders/train_util.py:159-163 / 242-247, re-expressed on pixel-major
activations [B*H*W][C]:
  * every Linear / 1x1 conv / 3x3 conv is slh_gemm (implicit GEMM for 3x3, channel concat and nearest-2x
    upsample folded into the A-operand addressing),
  * the LoRA branch of lora.py:108-112 is slh_skinny (down) + a rank-4 epilogue term (up) - skipped entirely
    when the adapters are switched off (multiplier 0 adds an exact 0 in the reference),
  * q/k/v (and cross k/v) projections are fused into one GEMM, all time_emb_proj layers into one GEMV.
Modes: "off" (LoRA disabled), "on" (LoRA enabled, no grad), "train" (LoRA enabled, activations kept, GEGLU
unfused so its pre-activation is available; `build_backward()` then emits the backward program).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple, Union

import torch

from . import lib
from .arena import Arena, Buf
from .config import UNetConfig
from .lora_store import LoraEntry, LoraStore
from .tuning import settle_tile, tuned_tile
from .weights import WeightStore


@dataclass
class Act:
    """Pixel-major activation view: rows = B*H*W pixels, C channels, row stride ld (elements)."""
    ptr: int
    B: int
    H: int
    W: int
    C: int
    ld: int
    buf: Optional[Buf] = None
    name: str = ""
    ln: Optional[tuple] = None     # (Buf of per-row 64-column chunk statistics written by the producing GEMM, chunks): folded LayerNorm

    @property
    def HW(self) -> int:
        return self.H * self.W

    @property
    def M(self) -> int:
        return self.B * self.H * self.W

    def cols(self, c0: int, c: int) -> "Act":
        return Act(self.ptr + 2 * c0, self.B, self.H, self.W, c, self.ld, self.buf, self.name + f"[:,{c0}:{c0 + c}]")

    def sample(self, b: int, nb: int = 1) -> "Act":
        return Act(self.ptr + 2 * b * self.HW * self.ld, nb, self.H, self.W, self.C, self.ld, self.buf,
                   self.name + f"[b{b}]")

    def tensor(self) -> torch.Tensor:
        """Debug view (real arenas only): [M][C] strided tensor."""
        base = self.buf.tensor.view(-1)
        off = (self.ptr - self.buf.ptr) // 2
        return base.as_strided((self.M, self.C), (self.ld, 1), off)


Src = Union[Act, Tuple[Act, Act]]

SPLITK_MAX_MN = 6 << 20          # output elements up to which split-K is considered (one 24 MB fp32 slab per slice at most)
SPLITK_MIN_K = int(os.environ.get("SLIDERS_SPLITK_MIN_K", "2048"))


def splitk_candidate(d) -> bool:
    """Few output tiles and a long reduction (the 1280-channel convolutions at 8x8 / 16x16 of SD-1.x and of SDXL at
    512x512: 10-40 tiles for 256 CUs, 29 MB of weights each): such a product gets a zeroed fp32 workspace so that
    slh_gemm may cut K into slices (tile bits 16-19, chosen by the tuner or by default_splitk below)."""
    return d.M * d.N <= SPLITK_MAX_MN and d.K >= SPLITK_MIN_K and not d.geglu and d.N % 4 == 0


def splitk_wanted(d) -> bool:
    """Provision the workspace when the tile that will run splits K: the tuned tile says so, or there is no tuned tile and
    the default would; in tuning mode (SLIDERS_NO_TUNING: scripts/tune_insitu.py tries split-K tiles on every candidate)
    for every candidate."""
    if not splitk_candidate(d):
        return False
    if os.environ.get("SLIDERS_NO_TUNING") or os.environ.get("SLIDERS_SPLITK_ALL"):
        return True
    return ((d.tile >> 16) & 15) > 1 if d.tile else bool(default_splitk(d))


def default_splitk(d) -> int:
    """Untuned shape: slices so that tiles x slices is about one workgroup per CU, at least 8 K tiles per slice."""
    tiles = ((d.M + 127) // 128) * ((d.N + 63) // 64)
    if tiles >= 128 or d.K < 4096:
        return 0
    s = min(8, max(1, 256 // tiles), d.K // 512)
    return 0 if s < 2 else (s << 16) | 0x412


SPLITK_TUNING_SLABS = 8          # slabs provisioned per candidate when the in-situ tuner may try any split factor


SPLITK_TICKETS = 4096            # 64-bit arrival tickets per plan (one per output tile of the largest split-K product)


def _plan_tickets(plan, n: int) -> int:
    """The arrival tickets of a plan's split-K products: ONE array per plan - every launch leaves its tickets zero and the
    launches of a plan are ordered on one stream.  Lives in the zero-init arena (zeroed at the head of the program)."""
    assert n <= SPLITK_TICKETS, f"split-K product with {n} output tiles"
    if getattr(plan, "_splitk_tickets", None) is None:
        plan._splitk_tickets = plan.zarena.alloc((2 * SPLITK_TICKETS,), torch.float32, "splitk.tickets")
    return plan._splitk_tickets.ptr


def provision_splitk(plan, d, name: str):
    """Give a product that will (or, in tuning mode, may) run split-K its slab workspace: one fp32 slab (M, N rounded up to whole
    tiles) per K slice
    (each slice writes its own, the last slice of a tile to arrive adds them in slice order inside the launch - bit-reproducible) and, with a
    fused adapter, two [M][ld_t] slabs per slice for T.  Fixes d.tile for untuned shapes."""
    if not splitk_wanted(d):
        return
    if not d.tile and d.M * d.N <= (1 << 20):      # the untuned default only for the small products it was measured on
        d.tile = default_splitk(d)
    tuning = os.environ.get("SLIDERS_NO_TUNING") or os.environ.get("SLIDERS_SPLITK_ALL")
    slabs = max(SPLITK_TUNING_SLABS if tuning else 0, (d.tile >> 16) & 15)
    if slabs < 2:
        return
    d.splitk_slabs = slabs
    d.splitk_c32 = plan.arena.alloc((slabs, (d.M + 255) // 256 * 256, (d.N + 127) // 128 * 128), torch.float32, name + ".splitk").ptr
    d.splitk_ticket = _plan_tickets(plan, ((d.M + 63) // 64) * ((d.N + 63) // 64))
    if d.lora_down:
        # one pair of T slabs per slice and per COLUMN TILE (each column tile's last slice reduces its own copy); sized for
        # the narrowest tile (64 columns) unless the tile is fixed
        ni = (d.tile & 15) if (d.tile and not tuning) else 1
        tiles_n = (d.N + 64 * ni - 1) // (64 * ni)
        d.splitk_t32 = plan.arena.alloc((tiles_n * 2 * slabs, d.M, d.ld_t), torch.float32, name + ".splitk_T").ptr


PREFETCH_MIN_BYTES = int(os.environ.get("SLIDERS_PREFETCH_MIN", str(6 << 20)))
PREFETCH_MAX_BYTES = 96 << 20

TOUCH_WINDOW = 6      # ops a weight touch may ride ahead of the product that needs the weights


def attach_weight_touch(prog: "lib.Program") -> "lib.Program":
    """The product path's weight prefetch (slh_gemm_desc.pf_*): every GEMM whose packed weights are 6-96 MB (GEGLU.proj 26 MB,
    ff.net.2 13 MB, q|k|v 10 MB at the 1280-channel level) has those bytes touched by the idle workgroup slots of an EARLIER
    launch that leaves >= 64 CUs free - the 160-tile products on the 128 x 128 ring tile (attention out-projections, attn2.to_q,
    ff.net.2 itself) - or, round 6, of the key-split self-attention launch (slh_attn_desc.pf_*: three workgroups per CU, 640 of 768
    slots taken) - the nearest such carrier within TOUCH_WINDOW ops that does not carry a touch yet.  No extra launch, no
    second stream.  SLIDERS_NO_WEIGHT_TOUCH=1 returns the program unchanged."""
    if os.environ.get("SLIDERS_NO_WEIGHT_TOUCH") is not None:
        return prog
    ops = list(prog.ops)

    def carrier(o, d):
        if o == lib.OP_ATTN_FWD:          # round 6: the key-split self-attention of the 32 x 32 level (its workgroups leave a slot per CU)
            return os.environ.get("SLIDERS_NO_ATTN_TOUCH") is None and lib.attn_carries_touch(d)
        if o != lib.OP_GEMM or d.mode != 0 or (d.tile & 0xFFFFFF) != 0x4412:
            return False
        return ((d.M + 127) // 128) * ((d.N + 127) // 128) <= 192
    used = set()
    for i, (o, d) in enumerate(ops):
        if o != lib.OP_GEMM or d.w_layout != 1:
            continue
        nb = (d.N + 63) // 64 * 64 * d.K * 2
        if not (PREFETCH_MIN_BYTES <= nb <= PREFETCH_MAX_BYTES):
            continue
        cand = reversed(range(max(0, i - TOUCH_WINDOW), i))     # nearest first (measured: 24.95 ms against 25.05 farthest-first, 25.27 without)
        if os.environ.get("SLIDERS_TOUCH_FARTHEST"):
            cand = range(max(0, i - TOUCH_WINDOW), i)
        for j in cand:
            if j not in used and carrier(*ops[j]):
                ops[j][1].pf_ptr, ops[j][1].pf_bytes = d.w, nb
                used.add(j)
                break
    if not used:
        return prog
    out = lib.Program()
    for (o, d), nm in zip(ops, prog.op_names):
        out.add(o, d, nm)
    return out


def _src_parts(x: Src):
    if isinstance(x, tuple):
        return x[0], x[1]
    return x, None


class UNetPlan:
    def __init__(self, cfg: UNetConfig, weights: WeightStore, arena: Arena, zarena: Arena, B: int, H: int, W: int,
                 ctx_len: int = 77, lora: Optional[LoraStore] = None, mode: str = "off",
                 lora_scale_ptr: int = 0, io: Optional[dict] = None):
        assert mode in ("off", "on", "train")
        assert mode == "off" or lora is not None
        self.cfg, self.w, self.arena, self.zarena = cfg, weights, arena, zarena
        self.B, self.H, self.W, self.ctx_len = B, H, W, ctx_len
        self.lora = lora if mode != "off" else None
        self.mode = mode
        self.train = mode == "train"
        self.lora_scale_ptr = lora_scale_ptr
        self.prog = lib.Program()
        self.tape: List[dict] = []
        self.lnfold_items: List[tuple] = []       # adapter sides of LayerNorm folds (slh_lora_lnfold_item as int64 words)
        self.zmark = zarena.mark()
        self._build_io(io)
        self._forward()
        self.zend = zarena.mark()
        # zero the fp32 accumulators (GroupNorm statistics ...) used by this program first
        head = lib.Program()
        if self.zend > self.zmark:
            p, n = zarena.region(self.zmark, self.zend)
            head.memset(p, n, 0, "zero_stats")
        if self.lnfold_items:
            self._lnfold_table = torch.tensor(self.lnfold_items, dtype=torch.int64, device=self.lora.params.device)
            head.add(lib.OP_LORA_LN_FOLD, lib.LoraLnFoldDesc(items=self._lnfold_table.data_ptr(), n=len(self.lnfold_items)),
                     "lora_ln_fold")
        head.extend(self.prog)
        self.prog = attach_weight_touch(head)
        # The text K/V of every cross-attention block depend only on the prompt embeddings (and frozen weights): inside a
        # denoise loop (train_util.py:263-294: same embeddings for every timestep) steps 2.. replay the pass without
        # the batched K/V projection and its head transpose.  Valid only while no other plan ran in between (plans
        # share the arena) and io["ctx"] is unchanged - the caller's responsibility (trainer / sampler loops).
        self.prog_text_cached = None
        if self.kv_all is not None:
            skip = {"attn2_kv_all", "attn2_vt_all", "lora_ln_fold"}     # (the adapters do not change inside a denoise loop either)
            pc = lib.Program()
            for (op, d), nm in zip(self.prog.ops, self.prog.op_names):
                if nm not in skip:
                    pc.add(op, d, nm)
            self.prog_text_cached = pc

    # ------------------------------------------------------------------------------------------------
    # buffers
    # ------------------------------------------------------------------------------------------------
    def _build_io(self, io):
        cfg, B = self.cfg, self.B
        if io is not None:
            self.io = io
            return
        a = self.arena
        self.io = {
            "sample": a.alloc((B, cfg.in_channels, self.H, self.W), torch.bfloat16, "in.sample"),
            "t": a.alloc((B, 1), torch.float32, "in.t"),
            "ctx": a.alloc((B, self.ctx_len, cfg.cross_attention_dim), torch.bfloat16, "in.ctx"),
            "eps": a.alloc((B, cfg.out_channels, self.H, self.W), torch.bfloat16, "out.eps"),
        }
        if cfg.is_xl:
            self.io["time_ids"] = a.alloc((B, 6), torch.float32, "in.time_ids")
            self.io["add_in"] = a.alloc((B, cfg.projection_class_embeddings_input_dim), torch.bfloat16, "in.add_in")

    def act(self, B, H, W, C, name="") -> Act:
        buf = self.arena.alloc((B * H * W, C), torch.bfloat16, name)
        return Act(buf.ptr, B, H, W, C, C, buf, name)

    def key_act(self, B, H, W, C, name="") -> Act:
        """An activation that is never written or read - a tape KEY for the gradient chain (the backward keys gradient buffers on
        the forward buffer's address): one aligned granule of the arena instead of a B*H*W x C tensor."""
        buf = self.arena.alloc((128,), torch.bfloat16, name)
        return Act(buf.ptr, B, H, W, C, C, buf, name)

    def f32(self, shape, name="", zero=False) -> Buf:
        return (self.zarena if zero else self.arena).alloc(shape, torch.float32, name)

    # ------------------------------------------------------------------------------------------------
    # op emitters
    # ------------------------------------------------------------------------------------------------
    def _lora_group(self, paths: List[str]) -> Optional[List[LoraEntry]]:
        if self.lora is None:
            return None
        return self.lora.fused_group(paths)

    def _conv_fields(self, d, src: Act, conv: dict, Ho: int, Wo: int):
        d.mode = 1
        d.batch, d.hs, d.ws = src.B, src.H, src.W
        d.src_xform = conv.get("xform", 0)
        d.stride = conv.get("stride", 1)
        d.ho, d.wo = Ho, Wo

    def skinny(self, x: Src, w_ptr: int, R: int, K: int, conv: Optional[dict], Mout: int, Ho: int, Wo: int,
               name: str, out: Optional[Buf] = None, bias_ptr: int = 0, out_kind: int = 0) -> Buf:
        x0, x1 = _src_parts(x)
        if out is None:
            out = self.f32((Mout, R), name + ".T")
        d = lib.SkinnyDesc(a0=x0.ptr, a1=x1.ptr if x1 else 0, w=w_ptr, bias=bias_ptr, out=out.ptr,
                           lda0=x0.ld, lda1=x1.ld if x1 else 0, ca0=x0.C, ca1=x1.C if x1 else 0,
                           M=Mout, R=R, K=K, ldo=R, out_kind=out_kind)
        if conv is not None:
            self._conv_fields(d, x0, conv, Ho, Wo)
        else:
            d.stride = 1
        self.prog.add(lib.OP_SKINNY, d, name)
        return out

    def gemm(self, x: Src, wname: str, N: int, name: str, bias: bool = True, conv: Optional[dict] = None,
             rowbias: Optional[Tuple[int, int]] = None, residual: Optional[Act] = None,
             lora_paths: Optional[List[str]] = None, geglu: bool = False, out: Optional[Act] = None,
             w_ptr: Optional[int] = None, bias_ptr: Optional[int] = None, vt_heads: Optional[int] = None,
             ln_stats: bool = False, ln_fold: Optional[Act] = None, geglu_pre: Optional[Act] = None,
             ln_mr: Optional[Buf] = None, tape_x: Optional[Act] = None, geglu16: bool = False,
             xattn: Optional[dict] = None, ln_norm: Optional[str] = None) -> Optional[Act]:
        """y = x . W^T (+bias)(+rowbias per sample)(+LoRA)(+residual).  conv: {'stride','xform'} for 3x3.
        vt_heads: the product is a fused q|k|v projection of that many heads; where the kernel supports it (no-grad
        passes, head_dim % 64 == 0) its V third is written head-transposed for slh_attn_fwd straight from the epilogue
        and self.last_vt = (pointer, 0) names it - one launch and one HBM round trip of V less per self-attention.
        ln_stats: also leave per-row statistics of the result for a LayerNorm folded into the NEXT product (out.ln, set only
        when the tile that will run supports it).  ln_fold = the un-normalised activation x whose producer left such
        statistics: this product computes Linear(LayerNorm(x)) from x itself with the gamma-scaled copy of the weights
        (weights.py _put_ln_folded); returns None - nothing emitted - when the tile that would run cannot (split-K).
        Training passes: geglu_pre receives proj(x) itself next to the GEGLU output (the backward's pre-activation; the tape
        records this product with it as its output), ln_mr the rows' (mean, rstd) of a folded LayerNorm, and tape_x stands
        in for the normalised tensor that was never written (a key for the gradient chain: nothing reads its memory).
        xattn = {k, vt_ptr, vt_heads, Tk, Tq, scale}: the product is attn2.to_q and, when the tile that runs is the 128 x 128 ring
        tile, the cross-attention behind it happens in its epilogue (slh_gemm_desc.xa_*): the returned activation is then the
        attention output and self.xattn_done says so.
        ln_fold together with lora_paths (ln_norm = the LayerNorm's weight name): the adapter's down-projection is folded as well
        (slh_gemm_desc.ln_lora_*; A . gamma and its row sums / offsets are rebuilt from the live parameters by ONE
        slh_lora_ln_fold launch at the head of the program) - only on the ping-pong tiles; None when another tile would run."""
        self.xattn_done = False
        x0, x1 = _src_parts(x)
        cin = x0.C + (x1.C if x1 else 0)
        B = x0.B
        if conv is not None:
            sh = 1 if conv.get("xform", 0) else 0
            HL, WL = x0.H << sh, x0.W << sh
            st = conv.get("stride", 1)
            Ho, Wo = (HL - 1) // st + 1, (WL - 1) // st + 1   # 3x3, pad 1
            K = 9 * cin
        else:
            Ho, Wo, K = x0.H, x0.W, cin
        M = B * Ho * Wo
        Nout = N // 2 if geglu else N
        if out is None:
            out = self.act(B, Ho, Wo, Nout, name)
        grp = self._lora_group(lora_paths) if lora_paths else None
        T = None
        fused = grp is not None and os.environ.get("SLIDERS_LORA_UNFUSED") is None
        if grp is not None:
            R = sum(e.target.rank for e in grp)
            if fused:
                # lora_down rides inside the GEMM (third operand tile); T is only written out for the backward
                T = self.f32((M, R), name + ".T") if self.train else None
            else:
                T = self.skinny(x, self.lora.down_ptr(grp[0]), R, K, conv, M, Ho, Wo, name + ".lora_down")
        sfx = "16" if geglu16 else ""      # GEGLU.proj in the 16 | 16 block order (slh_gemm_desc.geglu = 3)
        assert not geglu16 or (geglu and geglu_pre is None and not lora_paths)
        d = lib.GemmDesc(a0=x0.ptr, a1=x1.ptr if x1 else 0,
                         w=w_ptr if w_ptr is not None else self.w.ptr(wname + ".w" + sfx),
                         bias=(bias_ptr if bias_ptr is not None else (self.w.ptr(wname + ".b" + sfx) if bias else 0)),
                         rowbias=rowbias[0] if rowbias else 0,
                         lora_t=T.ptr if (T and not fused) else 0, lora_up=self.lora.up_ptr(grp[0]) if grp else 0,
                         lora_down=self.lora.down_ptr(grp[0]) if fused else 0,
                         lora_t_out=T.ptr if (T and fused and self.train) else 0,
                         lora_rank=(4 * len(grp)) if fused else 0,
                         lora_scale=self.lora_scale_ptr if grp else 0,
                         residual=residual.ptr if residual else 0, c=out.ptr,
                         lda0=x0.ld, lda1=x1.ld if x1 else 0, ca0=x0.C, ca1=x1.C if x1 else 0,
                         mode=0, stride=1, ldw=K, M=M, N=N, K=K,
                         ld_rowbias=rowbias[1] if rowbias else 0, rows_per_sample=Ho * Wo,
                         ld_t=(4 * len(grp)) if grp else 0, lora_groups=len(grp) if grp else 0,
                         ld_res=residual.ld if residual else 0, ldc=out.ld, geglu=(3 if geglu16 else 1) if geglu else 0, tile=0,
                         w_layout=1 if (w_ptr is None and self.w.packed) else 0)
        if conv is not None:
            self._conv_fields(d, x0, conv, Ho, Wo)
        # the tuned table keys the two sides of a folded LayerNorm apart (",ni" consumer, ",no" producer: the producer needs a
        # 128-column tile): the descriptor carries the fold BEFORE the lookup (the producer's pointer is a placeholder until the
        # tile that will run is known to support it)
        train_fold = self.train and os.environ.get("SLIDERS_TRAIN_NO_LN_FOLD") is None
        want_ln_out = bool(ln_stats and (not self.train or train_fold) and not geglu and N % 64 == 0)
        if ln_fold is not None:
            d.w, d.bias = self.w.ptr(wname + ".lnw" + sfx), 0
            d.ln_in, d.ln_in_chunks, d.ln_eps = ln_fold.ln[0].ptr, ln_fold.ln[1], 1e-5
            d.ln_s, d.ln_b = self.w.ptr(wname + ".lns" + sfx), self.w.ptr(wname + ".lnb" + sfx)
        if want_ln_out:
            d.ln_out = 8
        d.tile = tuned_tile(d)
        d.ln_out = 0
        if not d.tile and M <= 192 and N >= 4096:
            d.tile = 0x12        # few rows, very wide: 64-row tiles waste the least of the short M
        if xattn is not None and (d.tile == 0x4412 or os.environ.get("SLIDERS_XATTN_ALL")):
            # (where the table prefers another tile for the query projection - the 640-channel level - the two launches stay)
            d.tile = 0x4412
            k = xattn["k"]
            d.xa_k, d.xa_vt, d.xa_tk, d.xa_tq = k.ptr, xattn["vt_ptr"], xattn["Tk"], xattn["Tq"]
            d.xa_ldk, d.xa_ldvt, d.xa_vt_heads, d.xa_scale = k.ld, xattn["ldvt"], xattn["vt_heads"], xattn["scale"]
            self.xattn_done = True
        if ln_fold is not None:
            if ((d.tile >> 16) & 15) > 1 or (not d.tile and splitk_wanted(d)):
                return None
            if grp is not None:
                fam = (d.tile >> 12) & 15
                fits = (fam == 8 and (d.tile >> 4) & 15 == 1 and (d.tile & 15) <= 4) or (fam == 7 and (d.tile & 15) == 8)
                if not fused or ln_norm is None or not fits:
                    return None                   # (the 128-register tiles of gemm.hip have no room for the second fold)
                R = 4 * len(grp)
                a2 = self.arena.alloc((R, K), torch.bfloat16, name + ".lnA")
                sc = self.f32((2, 16), name + ".lnA_sc")
                d.lora_down, d.ln_lora_s, d.ln_lora_c = a2.ptr, sc.ptr, sc.ptr + 64
                self.lnfold_items.append((self.lora.down_ptr(grp[0]), self.w.ptr(ln_norm + ".g"), self.w.ptr(ln_norm + ".b"),
                                          a2.ptr, sc.ptr, sc.ptr + 64, R | (K << 32)))
            if ln_mr is not None:
                d.ln_mr_out = ln_mr.ptr
        else:
            provision_splitk(self, d, name)
        if geglu_pre is not None:
            assert geglu and geglu_pre.C == N
            d.geglu_pre, d.ld_pre = geglu_pre.ptr, geglu_pre.ld
        if want_ln_out and (d.tile >> 12) & 15 == 5:
            st = self.f32((N // 80, M, 2), name + ".ln_chunks")      # the 64 x 160 tile leaves 80-column chunks (a wave's columns)
            d.ln_out = st.ptr
            out.ln = (st, N // 80)
        elif want_ln_out and (d.tile >> 12) & 15 == 7:
            cw = 80 if (d.tile & 15) % 5 == 0 else 64                # the four-wave tiles: 80-column chunks where a wave owns 80 / 160 columns
            st = self.f32((N // cw, M, 2), name + ".ln_chunks")
            d.ln_out = st.ptr
            out.ln = (st, N // cw)
        elif want_ln_out and not d.splitk_c32 and (lib.gemm_variant(d) >> 4) & 15 == 2 and \
                ((d.tile >> 12) & 15 != 8 or (d.tile >> 4) & 15 == 4):
            st = self.f32((N // 64, M, 2), name + ".ln_chunks")
            d.ln_out = st.ptr
            out.ln = (st, N // 64)
        self.last_vt = None
        if vt_heads and not geglu and conv is None and os.environ.get("SLIDERS_NO_FUSED_VT") is None and \
                (not self.train or os.environ.get("SLIDERS_TRAIN_NO_FUSED_VT") is None):
            Cq = N // 3
            Dh, Tk = Cq // vt_heads, Ho * Wo
            pp_ok = (d.tile >> 12) & 15 != 8 or (d.tile & 0xFF) in (0x42, 0x14)     # ping-pong tiles: 256 x 256 and 128 x 256 only
            if (d.tile >> 12) & 15 == 7:                                            # four-wave tiles: whole tiles per sample, V block on a wave boundary
                pp_ok = Tk % 128 == 0 and (2 * Cq) % (16 * (d.tile & 15)) == 0
            if Dh % 64 == 0 and (2 * Cq) % 128 == 0 and Tk % 64 == 0 and not (d.tile >> 16) & 15 and pp_ok:
                vt = self.arena.alloc((B, vt_heads, Dh, Tk), torch.bfloat16, name + ".vt")
                d.vt_out, d.vt_col0, d.vt_D, d.vt_heads, d.vt_tokens, d.vt_ld = vt.ptr, 2 * Cq, Dh, vt_heads, Tk, Tk
                d.vt_also_c = 1 if self.train else 0      # the backward reads V row-major
                self.last_vt = (vt.ptr, 0)
        settle_tile(d)
        self.prog.add(lib.OP_GEMM, d, name)
        if self.train:
            self.tape.append(dict(op="gemm", x=tape_x if tape_x is not None else x,
                                  out=geglu_pre if geglu_pre is not None else out, wname=wname, N=N, K=K, conv=conv, grp=grp,
                                  T=T, residual=residual, rowbias=rowbias, name=name, Ho=Ho, Wo=Wo))
        return out

    def groupnorm(self, x: Src, wname: str, eps: float, act: int, name: str) -> Act:
        x0, x1 = _src_parts(x)
        C = x0.C + (x1.C if x1 else 0)
        B, H, W = x0.B, x0.H, x0.W
        G = self.cfg.norm_num_groups
        # statistics are reduced in a fixed order (bit-reproducible pass): per-workgroup partials + arrival tickets; only
        # the tickets need the zeroed arena
        stats = self.f32((B, G, 2), name + ".stats")
        prow, ntick = lib.gn_workspace(C, H * W, G)
        part = self.f32((B, prow, G, 2), name + ".partial")
        ticket = self.f32((B, ntick), name + ".ticket", zero=True)
        y = self.act(B, H, W, C, name)
        d = lib.GnDesc(x0=x0.ptr, x1=x1.ptr if x1 else 0, gamma=self.w.ptr(wname + ".g"), beta=self.w.ptr(wname + ".b"),
                       stats=stats.ptr, y=y.ptr, ldx0=x0.ld, ldx1=x1.ld if x1 else 0, c0=x0.C, c1=x1.C if x1 else 0,
                       batch=B, hw=H * W, groups=G, ldy=y.ld, eps=eps, act=act, partial=part.ptr, ticket=ticket.ptr)
        one = lib.gn_fused_ok(C, H * W, G)      # 1: tiny tensor, one workgroup per group set; 2: cache-resident slab, sibling workgroups
        if one == 2 and os.environ.get("SLIDERS_GN_ONE", "1") == "0":
            one = 0
        if one and os.environ.get("SLIDERS_GN_TWO_LAUNCH") is None:
            self.prog.add(lib.OP_GN_FUSED, d, name + ".fused")
        else:
            self.prog.add(lib.OP_GN_STATS, d, name + ".stats")
            self.prog.add(lib.OP_GN_APPLY, d, name + ".apply")
        if self.train:
            self.tape.append(dict(op="gn", x=x, out=y, wname=wname, eps=eps, act=act, stats=stats, name=name))
        return y

    def layernorm(self, x: Act, wname: str, name: str) -> Act:
        y = self.act(x.B, x.H, x.W, x.C, name)
        mr = self.f32((x.M, 2), name + ".mean_rstd") if self.train else None
        d = lib.LnDesc(x=x.ptr, gamma=self.w.ptr(wname + ".g"), beta=self.w.ptr(wname + ".b"), y=y.ptr,
                       mean_rstd=mr.ptr if mr else 0, M=x.M, C=x.C, ldx=x.ld, ldy=y.ld, eps=1e-5)
        self.prog.add(lib.OP_LAYERNORM, d, name)
        if self.train:
            self.tape.append(dict(op="ln", x=x, out=y, wname=wname, mr=mr, name=name))
        return y

    def ln_gemm(self, h: Act, norm: str, wname: str, N: int, bias: bool = True, lora_paths: Optional[List[str]] = None,
                geglu: bool = False, vt_heads: Optional[int] = None, geglu_pre: Optional[Act] = None, geglu16: bool = False,
                xattn: Optional[dict] = None) -> Act:
        """Linear(LayerNorm(h)): folded into one product when h's producer left row statistics, the consumer carries no adapter
        and the pass keeps no tape; the LayerNorm launch + the plain product otherwise."""
        grp = self._lora_group(lora_paths) if lora_paths else None
        foldable = h.ln is not None and getattr(self.w, "ln_fold", False) and \
            self.w.has(wname + ".lnw") and h.C % 64 == 0 and h.C <= 1280 and h.ld == h.C
        if grp is not None and foldable and not self.train and not geglu and os.environ.get("SLIDERS_NO_LORA_LN_FOLD") is None:
            # adapter-carrying consumer (q|k|v under noxattn): the fold covers the adapter's down-projection too
            amark, nallocs, nitems = self.arena.mark(), len(self.arena.allocs), len(self.lnfold_items)
            pmark = self.prog.mark()
            y = self.gemm(h, wname, N, wname, bias=False, lora_paths=lora_paths, vt_heads=vt_heads, ln_fold=h, ln_norm=norm)
            if y is not None:
                return y
            self.arena.reset(amark)
            del self.arena.allocs[nallocs:]
            del self.lnfold_items[nitems:]
            self.prog.truncate(pmark)             # (an unfused adapter's down-projection may have been recorded ahead of the refusal)
        if grp is None and foldable:
            if not self.train:
                amark, nallocs, pmark = self.arena.mark(), len(self.arena.allocs), self.prog.mark()
                y = self.gemm(h, wname, N, wname, bias=False, geglu=geglu, vt_heads=vt_heads, ln_fold=h,
                              geglu16=geglu16 and self.w.has(wname + ".lnw16"), xattn=xattn)
                if y is not None:
                    return y
                self.arena.reset(amark)           # fold refused: the output it had reserved goes back
                del self.arena.allocs[nallocs:]
                self.prog.truncate(pmark)
            elif os.environ.get("SLIDERS_TRAIN_NO_LN_FOLD") is None:
                # training pass: the same fold; the product also leaves (mean, rstd) per row for the LayerNorm backward, and the
                # tape keeps the LayerNorm and the product as two records around a stand-in for the normalised tensor
                amark, nallocs, pmark = self.arena.mark(), len(self.arena.allocs), self.prog.mark()
                mr = self.f32((h.M, 2), norm + ".mean_rstd")
                stand_in = self.key_act(h.B, h.H, h.W, h.C, norm + ".unwritten")
                mark = len(self.tape)
                self.tape.append(dict(op="ln", x=h, out=stand_in, wname=norm, mr=mr, name=norm))
                y = self.gemm(h, wname, N, wname, bias=False, geglu=geglu, ln_fold=h, ln_mr=mr, tape_x=stand_in,
                              geglu_pre=geglu_pre)
                if y is not None:
                    return y
                del self.tape[mark:]              # fold refused (split-K tile): give the arena back and drop anything recorded
                self.arena.reset(amark)
                del self.arena.allocs[nallocs:]
                self.prog.truncate(pmark)
        n = self.layernorm(h, norm, norm)
        return self.gemm(n, wname, N, wname, bias=bias, lora_paths=lora_paths, geglu=geglu, vt_heads=vt_heads,
                         geglu_pre=geglu_pre, geglu16=geglu16 and geglu_pre is None and self.w.has(wname + ".w16"), xattn=xattn)

    def attention(self, q: Act, k: Act, v: Act, Tk: int, heads: int, name: str, vt_pre=None) -> Act:
        """q [B*Tq][C] view, k/v [B*Tk][C] views (any ld); returns [B*Tq][C].  vt_pre = (pointer to this layer's first
        head inside an already transposed [B][heads_total][Dp][ldt] array, heads_total)."""
        B, Tq, C = q.B, q.HW, q.C
        D = C // heads
        assert D * heads == C and D % 8 == 0 and D <= 192, f"unsupported head_dim {D}"
        Dp = (D + 63) // 64 * 64
        ldt = (Tk + 63) // 64 * 64
        if vt_pre is None:
            vt = self.arena.alloc((B, heads, Dp, ldt), torch.bfloat16, name + ".vt")
            self.prog.add(lib.OP_TRANSPOSE_HEADS, lib.TransposeDesc(src=v.ptr, dst=vt.ptr, B=B, H=heads, T=Tk, ld=v.ld,
                                                                    ldt=ldt, D=D), name + ".vt")
            vt_ptr, vt_heads = vt.ptr, 0
        else:
            vt_ptr, vt_heads = vt_pre
        o = self.act(q.B, q.H, q.W, C, name)
        lse = self.f32((B * heads * Tq + 64,), name + ".lse") if self.train else None   # padded: bwd reads by 64s
        d = lib.AttnDesc(q=q.ptr, k=k.ptr, vt=vt_ptr, o=o.ptr, lse=lse.ptr if lse else 0, B=B, H=heads, Tq=Tq, Tk=Tk,
                         ldq=q.ld, ldk=k.ld, ldvt=ldt, ldo=o.ld, scale=D ** -0.5, D=D, vt_batch_heads=vt_heads)
        self.prog.add(lib.OP_ATTN_FWD, d, name)
        if self.train:
            self.tape.append(dict(op="attn", q=q, k=k, v=v, o=o, lse=lse, Tk=Tk, heads=heads, name=name))
        return o

    # ------------------------------------------------------------------------------------------------
    # network pieces
    # ------------------------------------------------------------------------------------------------
    def _embeddings(self):
        cfg, B, w = self.cfg, self.B, self.w
        ted = cfg.time_embed_dim
        c0 = cfg.block_out_channels[0]
        a = self.arena
        t_in = a.alloc((B, c0), torch.bfloat16, "temb.sin")
        self.prog.add(lib.OP_TEMBED, lib.TembedDesc(vals=self.io["t"].ptr, out=t_in.ptr, nb=B, n_vals=1, dim=c0,
                                                    ldo=c0, col0=0), "time_proj")
        h1 = a.alloc((B, ted), torch.bfloat16, "temb.h1")
        self._gemv(t_in.ptr, c0, "time_embedding.linear_1", ted, c0, h1.ptr, ted, 0, name="time_embedding.linear_1")
        emb = a.alloc((B, ted), torch.bfloat16, "temb.emb")
        self._gemv(h1.ptr, ted, "time_embedding.linear_2", ted, ted, emb.ptr, ted, 1, name="time_embedding.linear_2")
        if cfg.is_xl:
            add_in = self.io["add_in"]
            pin = cfg.projection_class_embeddings_input_dim
            self.prog.add(lib.OP_TEMBED, lib.TembedDesc(vals=self.io["time_ids"].ptr, out=add_in.ptr, nb=B, n_vals=6,
                                                        dim=cfg.addition_time_embed_dim, ldo=pin,
                                                        col0=cfg.pooled_dim), "add_time_proj")
            g1 = a.alloc((B, ted), torch.bfloat16, "aemb.h1")
            self._gemv(add_in.ptr, pin, "add_embedding.linear_1", ted, pin, g1.ptr, ted, 0, name="add_embedding.linear_1")
            emb2 = a.alloc((B, ted), torch.bfloat16, "temb.emb_sum")
            self._gemv(g1.ptr, ted, "add_embedding.linear_2", ted, ted, emb2.ptr, ted, 1, addend=(emb.ptr, ted),
                       name="add_embedding.linear_2")
            emb = emb2
        self.emb = emb
        # every ResnetBlock2D.time_emb_proj(silu(emb)) in one launch
        tot = w.temb_total
        self.temb_all = a.alloc((B, tot), torch.bfloat16, "temb.proj_all")
        lt = None
        if self.lora is not None and self.lora.temb_entries:
            L = len(self.lora.temb_entries)
            if L != len(w.resnet_paths):
                raise NotImplementedError("time_emb_proj adapters must cover every resnet or none")
            lt = self.f32((B, 4 * L), "temb.lora_T")
            d = lib.GemvDesc(x=emb.ptr, w=self.lora.params.data_ptr() + 2 * self.lora.temb_down_off, y=lt.ptr,
                             nb=B, N=4 * L, K=ted, ldx=ted, ldy=4 * L, in_act=1, out_f32=1)
            self.prog.add(lib.OP_GEMV, d, "temb.lora_down")
            if not hasattr(self.lora, "temb_tcol"):
                tcol = torch.empty(tot, dtype=torch.int32)
                for i, p in enumerate(w.resnet_paths):
                    e = self.lora.temb_entries[i]
                    assert e.target.module_path == p + ".time_emb_proj"
                    o = w.temb_offsets[p]
                    tcol[o:o + e.target.out_dim] = 4 * i
                self.lora.temb_tcol = tcol.to(self.lora.params.device)
        d = lib.GemvDesc(x=emb.ptr, w=w.ptr("temb_proj.w"), bias=w.ptr("temb_proj.b"), y=self.temb_all.ptr,
                         nb=B, N=tot, K=ted, ldx=ted, ldy=tot, in_act=1, out_f32=0)
        if lt is not None:
            d.lora_t = lt.ptr
            d.ld_t = lt.shape[1]
            d.lora_tcol = self.lora.temb_tcol.data_ptr()
            d.lora_up = self.lora.params.data_ptr() + 2 * self.lora.temb_up_off
            d.lora_scale = self.lora_scale_ptr
        self.prog.add(lib.OP_GEMV, d, "time_emb_proj_all")
        self.temb_lora_T = lt

    def _gemv(self, x_ptr, ldx, wname, N, K, y_ptr, ldy, in_act, addend=None, name=""):
        d = lib.GemvDesc(x=x_ptr, w=self.w.ptr(wname + ".w"), bias=self.w.ptr(wname + ".b"), y=y_ptr,
                         addend=addend[0] if addend else 0, ld_add=addend[1] if addend else 0,
                         nb=self.B, N=N, K=K, ldx=ldx, ldy=ldy, in_act=in_act, out_f32=0)
        self.prog.add(lib.OP_GEMV, d, name)

    def _resnet(self, x: Src, path: str, cout: int) -> Act:
        x0, x1 = _src_parts(x)
        cin = x0.C + (x1.C if x1 else 0)
        eps = self.cfg.norm_eps
        a1 = self.groupnorm(x, path + ".norm1", eps, 1, path + ".norm1")
        rb = (self.temb_all.ptr + 2 * self.w.temb_offsets[path], self.w.temb_total)
        h = self.gemm(a1, path + ".conv1", cout, path + ".conv1", conv={}, rowbias=rb, lora_paths=[path + ".conv1"])
        if self.train:
            self.tape[-1]["temb_path"] = path
        a2 = self.groupnorm(h, path + ".norm2", eps, 1, path + ".norm2")
        if cin != cout:
            sc = self.gemm(x, path + ".conv_shortcut", cout, path + ".conv_shortcut",
                           lora_paths=[path + ".conv_shortcut"])
        else:
            assert x1 is None
            sc = x0
        return self.gemm(a2, path + ".conv2", cout, path + ".conv2", conv={}, residual=sc,
                         lora_paths=[path + ".conv2"])

    def _tblock(self, h: Act, path: str, heads: int, ctx: Act) -> Act:
        C = h.C
        a1, a2 = path + ".attn1", path + ".attn2"
        # BasicTransformerBlock.norm1/2/3: in the no-grad passes a LayerNorm whose consumer carries no adapter is folded
        # into that product (ln_gemm); its producer leaves the row statistics (ln_stats)
        qkv = self.ln_gemm(h, path + ".norm1", a1 + ".qkv", 3 * C, bias=False,
                           lora_paths=[a1 + ".to_q", a1 + ".to_k", a1 + ".to_v"], vt_heads=heads)
        T = h.HW
        o1 = self.attention(qkv.cols(0, C), qkv.cols(C, C), qkv.cols(2 * C, C), T, heads, a1 + ".sdpa", vt_pre=self.last_vt)
        h1 = self.gemm(o1, a1 + ".out", C, a1 + ".out", residual=h, lora_paths=[a1 + ".to_out.0"], ln_stats=True)
        vt_pre = None
        if self.kv_all is not None:
            k_off, v_off = self.w.kv_all_offset[a2]
            k2, v2, kv = self.kv_all.cols(k_off, C), self.kv_all.cols(v_off, C), self.kv_all
            if self.vt_all is not None:
                D = C // heads
                Dp, ldt = (D + 63) // 64 * 64, (self.ctx_len + 63) // 64 * 64
                vt_pre = (self.vt_all.ptr + 2 * ((v_off - self.w.kv_all_vbase) // D) * Dp * ldt, self.vt_all_heads)
        else:
            kv = self.gemm(ctx, a2 + ".kv", 2 * C, a2 + ".kv", bias=False, lora_paths=[a2 + ".to_k", a2 + ".to_v"])
            k2, v2 = kv.cols(0, C), kv.cols(C, C)
        if self._lora_group([a2 + ".to_k", a2 + ".to_v"]) is None:
            self.nograd_kv.add(kv.buf.ptr)      # text K/V carry no gradient unless they are adapted
        # no-grad passes, head dim 64, text keys, no adapter on to_q: the attention runs in the epilogue of the query projection
        xa = None
        D2 = C // heads
        # (the gate restates slh_gemm's own checks for xa_*: head dim 64 with whole 64-column heads, 65..96 keys - two 64-key V^T
        # tiles are always staged, so xa_ldvt = roundup(ctx_len, 64) must reach 128 - and whole 128-row query tiles per sample)
        if not self.train and vt_pre is not None and D2 == 64 and C % 64 == 0 and 64 < self.ctx_len <= 96 and h.HW % 128 == 0 and \
                self._lora_group([a2 + ".to_q"]) is None and os.environ.get("SLIDERS_NO_FUSED_XATTN") is None:
            xa = dict(k=k2, vt_ptr=vt_pre[0], vt_heads=vt_pre[1], Tk=self.ctx_len, Tq=h.HW, scale=D2 ** -0.5,
                      ldvt=(self.ctx_len + 63) // 64 * 64)
        q2 = self.ln_gemm(h1, path + ".norm2", a2 + ".q", C, bias=False, lora_paths=[a2 + ".to_q"], xattn=xa)
        if xa is not None and self.xattn_done:
            o2 = q2
        else:
            o2 = self.attention(q2, k2, v2, self.ctx_len, heads, a2 + ".sdpa", vt_pre=vt_pre)
        h2 = self.gemm(o2, a2 + ".out", C, a2 + ".out", residual=h1, lora_paths=[a2 + ".to_out.0"], ln_stats=True)
        if self.train and (self._lora_group([path + ".ff.net.0.proj"]) is not None or
                           os.environ.get("SLIDERS_TRAIN_UNFUSED_GEGLU") is not None):
            n3 = self.layernorm(h2, path + ".norm3", path + ".norm3")
            pre = self.gemm(n3, path + ".ff1", 8 * C, path + ".ff1", lora_paths=[path + ".ff.net.0.proj"])
            ff = self.act(h.B, h.H, h.W, 4 * C, path + ".geglu")
            self.prog.add(lib.OP_ELEMENTWISE, lib.EwDesc(a=pre.ptr, out=ff.ptr, M=pre.M, C=4 * C, lda=pre.ld, ldo=ff.ld,
                                                         op=lib.EW_GEGLU_FWD), path + ".geglu")
            self.tape.append(dict(op="geglu", pre=pre, out=ff, name=path + ".geglu"))
        elif self.train:
            # one launch: the GEGLU epilogue also stores proj(x) for the backward (and norm3 folds into it when h2's producer
            # left row statistics)
            pre = self.act(h.B, h.H, h.W, 8 * C, path + ".ff1")
            ff = self.ln_gemm(h2, path + ".norm3", path + ".ff1", 8 * C, geglu=True, geglu_pre=pre)
            self.tape.append(dict(op="geglu", pre=pre, out=ff, name=path + ".geglu"))
        else:
            grp = self._lora_group([path + ".ff.net.0.proj"])
            if grp is not None:
                raise NotImplementedError("LoRA on GEGLU.proj is not a reference target")
            ff = self.ln_gemm(h2, path + ".norm3", path + ".ff1", 8 * C, geglu=True, geglu16=getattr(self.w, "geglu16", False))
        return self.gemm(ff, path + ".ff2", C, path + ".ff2", residual=h2, lora_paths=[path + ".ff.net.2"], ln_stats=True)

    def _transformer(self, x: Act, path: str, layers: int, heads: int) -> Act:
        g = self.groupnorm(x, path + ".norm", 1e-6, 0, path + ".norm")
        h = self.gemm(g, path + ".proj_in", x.C, path + ".proj_in", lora_paths=[path + ".proj_in"], ln_stats=True)
        for k in range(layers):
            h = self._tblock(h, f"{path}.transformer_blocks.{k}", heads, self.ctx)
        return self.gemm(h, path + ".proj_out", x.C, path + ".proj_out", residual=x, lora_paths=[path + ".proj_out"])

    def _forward(self):
        cfg, B, H, W = self.cfg, self.B, self.H, self.W
        boc = cfg.block_out_channels
        self._embeddings()
        ctxb = self.io["ctx"]
        self.ctx = Act(ctxb.ptr, B, 1, self.ctx_len, cfg.cross_attention_dim, cfg.cross_attention_dim, ctxb, "ctx")
        # every transformer block projects the same text embeddings to K/V: when those projections carry no adapter they
        # run as ONE GEMM over the concatenated weights at the head of the pass - the training forward included (rounds 1-2
        # kept per-block projections there because the batched layout exposed an LDS-DMA publish race in the attention
        # kernels: profiles/r02_lds_dma_race_fix.txt; fixed, and green on the GPU suite x2 + 0/16 on the reproducer,
        # profiles/r03_first_call.txt.  SLIDERS_TRAIN_KV_PER_BLOCK=1 restores the old form for A/B runs.)
        self.kv_all = self.vt_all = None
        kvo = getattr(self.w, "kv_all_offset", None)
        if kvo and not (self.train and os.environ.get("SLIDERS_TRAIN_KV_PER_BLOCK")) and \
                all(self._lora_group([a + ".to_k", a + ".to_v"]) is None for a in kvo):
            n_all = self.w.gemm_shape["attn2_kv_all.w"][0]
            self.kv_all = self.gemm(self.ctx, "attn2_kv_all", n_all, "attn2_kv_all", bias=False)
            # one head dim everywhere (SDXL: 64) -> the V halves of all blocks are one [B*77][sum(C)] matrix whose
            # heads transpose in ONE launch; each block's attention then points at its own heads of that array
            dims = {boc[i] // cfg.attention_head_dim[i] for i, t in enumerate(cfg.down_block_types) if t != "DownBlock2D"}
            dims.add(boc[-1] // cfg.attention_head_dim[-1])
            if len(dims) == 1:
                D = dims.pop()
                vb = self.w.kv_all_vbase
                self.vt_all_heads = vb // D
                Dp, ldt = (D + 63) // 64 * 64, (self.ctx_len + 63) // 64 * 64
                self.vt_all = self.arena.alloc((B, self.vt_all_heads, Dp, ldt), torch.bfloat16, "attn2_vt_all")
                self.prog.add(lib.OP_TRANSPOSE_HEADS, lib.TransposeDesc(
                    src=self.kv_all.ptr + 2 * vb, dst=self.vt_all.ptr, B=B, H=self.vt_all_heads, T=self.ctx_len,
                    ld=self.kv_all.ld, ldt=ldt, D=D), "attn2_vt_all")
        h = self.act(B, H, W, boc[0], "conv_in")
        self.nograd = {ctxb.ptr, h.buf.ptr}     # nothing trainable upstream of these
        self.nograd_kv = set()
        self.prog.add(lib.OP_CONV_IN, lib.ConvInDesc(x=self.io["sample"].ptr, w=self.w.ptr("conv_in.w"),
                                                     bias=self.w.ptr("conv_in.b"), y=h.ptr, batch=B,
                                                     cin=cfg.in_channels, h=H, wd=W, cout=boc[0], ldy=h.ld), "conv_in")
        skips = [h]
        # block name -> the activation diffusers' block returns (after its down / upsampler): read by the parity tests' per-block
        # error ledger after a pass (no buffer is reused inside a pass, so they are all still there); nothing else uses it
        self.block_out: Dict[str, Act] = {}
        for i, t in enumerate(cfg.down_block_types):
            path = f"down_blocks.{i}"
            cout = boc[i]
            final = i == len(boc) - 1
            for j in range(cfg.layers_per_block):
                h = self._resnet(h, f"{path}.resnets.{j}", cout)
                if t != "DownBlock2D":
                    h = self._transformer(h, f"{path}.attentions.{j}", cfg.transformer_layers_per_block[i],
                                          cfg.attention_head_dim[i])
                skips.append(h)
            if not final:
                p = f"{path}.downsamplers.0.conv"
                h = self.gemm(h, p, cout, p, conv={"stride": 2}, lora_paths=[p])
                skips.append(h)
            self.block_out[path] = h
        mp = "mid_block"
        h = self._resnet(h, mp + ".resnets.0", boc[-1])
        h = self._transformer(h, mp + ".attentions.0", cfg.transformer_layers_per_block[-1], cfg.attention_head_dim[-1])
        h = self._resnet(h, mp + ".resnets.1", boc[-1])
        self.block_out[mp] = h
        rboc = tuple(reversed(boc))
        rheads = tuple(reversed(cfg.attention_head_dim))
        rtl = tuple(reversed(cfg.transformer_layers_per_block))
        for i, t in enumerate(cfg.up_block_types):
            path = f"up_blocks.{i}"
            cout = rboc[i]
            final = i == len(boc) - 1
            for j in range(cfg.layers_per_block + 1):
                skip = skips.pop()
                h = self._resnet((h, skip), f"{path}.resnets.{j}", cout)
                if t != "UpBlock2D":
                    h = self._transformer(h, f"{path}.attentions.{j}", rtl[i], rheads[i])
            if not final:
                p = f"{path}.upsamplers.0.conv"
                h = self.gemm(h, p, cout, p, conv={"xform": 1}, lora_paths=[p])
            self.block_out[path] = h
        assert not skips
        g = self.groupnorm(h, "conv_norm_out", cfg.norm_eps, 1, "conv_norm_out")
        self.skinny(g, self.w.ptr("conv_out.w"), cfg.out_channels, 9 * g.C, {}, g.M, g.H, g.W, "conv_out",
                    out=self.io["eps"], bias_ptr=self.w.ptr("conv_out.b"), out_kind=1)
        if self.train:
            self.tape.append(dict(op="conv_out", x=g, name="conv_out"))


# ======================================================================================================
# backward through the frozen net into the LoRA adapters (loss.backward(), train_lora_xl.py:345)
# ======================================================================================================
class BackwardPlan:
    """Reverse walk over a train-mode UNetPlan's tape.  Gradients flow only for samples [b0, b0+nb): in the
    reference's CFG pair the unconditional half receives an exactly-zero gradient (d/du of u + 1*(t-u) is
    1 - 1 = 0 in bf16, train_util.py:250-253 with guidance_scale=1), so its backward is skipped.

    Every backward-data product reuses slh_gemm with the pre-transposed frozen weights; LoRA gradients are
    skinny reductions (slh_skinny with the up matrix as a [K][4] down-projection of dY, slh_lora_wgrad)."""

    def __init__(self, fwd: UNetPlan, b0: int, nb: int, one_ptr: int):
        assert fwd.train and fwd.lora is not None
        self.f = fwd
        self.cfg, self.w, self.arena, self.zarena, self.lora = fwd.cfg, fwd.w, fwd.arena, fwd.zarena, fwd.lora
        self.b0, self.nb = b0, nb
        self.one_ptr = one_ptr
        self.scale_ptr = fwd.lora_scale_ptr
        self.prog = lib.Program()
        self._g: Dict[int, Act] = {}          # base buffer ptr -> full-width gradient buffer (nb samples)
        self._written = set()
        zmark = self.zarena.mark()
        H, W = fwd.H, fwd.W
        self.deps_pix = self.arena.alloc((nb * H * W, 4), torch.float32, "bwd.deps_pix")
        # launches that do not sit on the dependency chain are collected and issued as ONE batched launch each
        # (slh_batch_desc): the head transposes of FORWARD activations (K^T, Q^T of every attention layer) at the head of the
        # program, and the adapters' weight gradients at its tail - except the up-gradient (dB) of a module whose output
        # gradient buffer doubles as its residual input's (alias_grad): later accumulations would change dY under it
        # dO^T of a self-attention (for dK / dV) comes out of the backward-data product that produces dO (vt_out + vt_also_c)
        # instead of a transpose launch: attention output buffer -> (heads, Tq), filled from the tape; -> (buffer, ldt) once made
        self._wants_dot: Dict[int, Tuple[int, int]] = {}
        self._dot_made: Dict[int, Tuple] = {}
        for rec in fwd.tape:
            if rec["op"] == "attn" and rec["k"].buf.ptr not in fwd.nograd_kv and os.environ.get("SLIDERS_BWD_DOT_LAUNCH") is None:
                self._wants_dot[rec["o"].buf.ptr] = (rec["heads"], rec["q"].HW)
        # GEGLU outputs -> their tape record: the backward-data product of the Linear that consumes one writes d(proj) itself
        self._geglu_of: Dict[int, dict] = {}
        if os.environ.get("SLIDERS_BWD_UNFUSED_GEGLU") is None:
            for rec in fwd.tape:
                if rec["op"] == "geglu":
                    self._geglu_of[rec["out"].buf.ptr] = rec
        self._tr_batch: List = []
        self._wg_batch: Dict[int, List] = {4: [], 12: []}
        self._keep: List = []            # device tables of the batched launches
        self.batched = os.environ.get("SLIDERS_BWD_UNBATCHED") is None
        # weight gradients reduced over their M splits in a fixed order (slabs + tickets) instead of fp32 atomics: the LoRA
        # gradient buffer is bit-reproducible.  SLIDERS_WGRAD_ATOMIC=1: the old form (A/B)
        self.wgrad_fixed_order = os.environ.get("SLIDERS_WGRAD_ATOMIC") is None
        self.fuse_u = os.environ.get("SLIDERS_BWD_UNFUSED_U") is None
        self._uses_up_t = False
        self._walk()
        dev = None if self.arena.virtual else fwd.lora.params.device
        for R, descs in self._wg_batch.items():
            if descs:
                if self.wgrad_fixed_order:
                    for d in descs:
                        d.slabs = 8                 # marker: slab geometry (the workspace is the batch's)
                bd, keep = lib.batch_table(lib.OP_WGRAD_BATCH, descs, dev, arg=R)
                if self.wgrad_fixed_order:
                    # one slab of 256 R floats and one ticket per workgroup: the M splits of a column block meet in a fixed order
                    bd.slabs = self.arena.alloc((bd.total, 256 * R), torch.float32, f"bwd.wgrad_slabs_R{R}").ptr
                    bd.tickets = self.zarena.alloc((bd.total,), torch.int32, f"bwd.wgrad_tickets_R{R}").ptr
                self._keep.append(keep)
                self.prog.add(lib.OP_WGRAD_BATCH, bd, f"bwd.wgrad_batch_R{R}")
        zend = self.zarena.mark()
        head = lib.Program()
        if zend > zmark:
            p, n = self.zarena.region(zmark, zend)
            head.memset(p, n, 0, "zero_bwd_stats")
        if self._uses_up_t:
            if self.arena.virtual:
                gd = lib.Gather16Desc(src=0x1000, idx=0x1000, out=0x1000, n=1)
            else:
                gd = lib.Gather16Desc(src=fwd.lora.params.data_ptr(), idx=fwd.lora.up_t_index.data_ptr(),
                                      out=fwd.lora.up_t.data_ptr(), n=fwd.lora.up_t.numel())
            head.add(lib.OP_GATHER16, gd, "bwd.up_t")
        if self._tr_batch:
            bd, keep = lib.batch_table(lib.OP_TRANSPOSE_BATCH, self._tr_batch, dev)
            self._keep.append(keep)
            head.add(lib.OP_TRANSPOSE_BATCH, bd, "bwd.kt_qt_batch")
        head.extend(self.prog)
        self.prog = head

    # ---- gradient buffers ------------------------------------------------------------------------
    def _sl(self, a: Act) -> Act:
        return a.sample(self.b0, self.nb)

    def grad(self, a: Act, write: bool = True):
        """(gradient view matching `a` for the selected samples, accumulate?)"""
        base = a.buf.ptr
        g = self._g.get(base)
        if g is None:
            rows = self.nb * a.HW
            buf = self.arena.alloc((rows, a.ld), torch.bfloat16, "g:" + a.name)
            g = Act(buf.ptr, self.nb, a.H, a.W, a.ld, a.ld, buf, "g:" + a.name)
            self._g[base] = g
        c0 = ((a.ptr - a.buf.ptr) // 2) % a.ld
        view = Act(g.ptr + 2 * c0, self.nb, a.H, a.W, a.C, g.ld, g.buf, g.name)
        key = (base, c0, a.C)
        acc = key in self._written
        if write:
            self._written.add(key)
        return view, acc

    def has_grad(self, a: Act) -> bool:
        """True once any column range of a's buffer has received a gradient."""
        return any(k[0] == a.buf.ptr for k in self._written)

    def alias_grad(self, a: Act, g: Act):
        self._g[a.buf.ptr] = Act(g.ptr, g.B, g.H, g.W, a.ld, g.ld, g.buf, g.name)
        self._written.add((a.buf.ptr, 0, a.C))

    # ---- op emitters -----------------------------------------------------------------------------
    def _ew(self, op, a: Act, out: Act, b: Optional[Act] = None, name="", C=None, iarg=0, iarg2=0, M=None):
        d = lib.EwDesc(a=a.ptr, b=b.ptr if b else 0, out=out.ptr, M=M if M is not None else a.M, C=C or a.C, lda=a.ld,
                       ldb=b.ld if b else 0, ldo=out.ld, op=op, iarg=iarg, iarg2=iarg2)
        self.prog.add(lib.OP_ELEMENTWISE, d, name)

    def add_into(self, dst_fwd: Act, src: Act, name: str):
        g, acc = self.grad(dst_fwd)
        if acc:
            self._ew(lib.EW_ADD, g, g, src, name + ".add")
        else:
            self._ew(lib.EW_COPY, src, g, name=name + ".copy")

    def _walk(self):
        f = self.f
        for rec in reversed(f.tape):
            getattr(self, "_b_" + rec["op"])(rec)

    def _b_conv_out(self, rec):
        x = rec["x"]
        gx, acc = self.grad(x)
        d = lib.LoraCdgradDesc(u=self.deps_pix.ptr, a_down=self.w.ptr("conv_out.w"), scale=self.one_ptr, gx=gx.ptr,
                               batch=self.nb, hl=x.H, wl=x.W, ho=x.H, wo=x.W, stride=1, cin=x.C, ldu=4, ldgx=gx.ld,
                               accumulate=1 if acc else 0)
        self.prog.add(lib.OP_LORA_CONV_DGRAD, d, "bwd.conv_out")

    def _b_gn(self, rec):
        x0, x1 = _src_parts(rec["x"])
        y = rec["out"]
        if not self.has_grad(y):
            return
        need0 = x0.buf.ptr not in self.f.nograd
        need1 = x1 is not None and x1.buf.ptr not in self.f.nograd
        if not (need0 or need1):
            return
        gy, _ = self.grad(y, write=False)
        G = self.cfg.norm_num_groups
        bst = self.arena.alloc((self.nb, G, 2), torch.float32, "bwd." + rec["name"] + ".bstats")
        prow, ntick = lib.gn_workspace(x0.C + (x1.C if x1 is not None else 0), x0.HW, G)
        bpart = self.arena.alloc((self.nb, prow, G, 2), torch.float32, "bwd." + rec["name"] + ".bpartial")
        btick = self.zarena.alloc((self.nb, ntick), torch.float32, "bwd." + rec["name"] + ".bticket")
        x0s = self._sl(x0)
        x1s = self._sl(x1) if x1 is not None else None
        g0, a0 = self.grad(x0) if need0 else (None, False)
        g1, a1 = self.grad(x1) if need1 else (None, False)
        st = rec["stats"].ptr + 4 * self.b0 * G * 2
        d = lib.GnBwdDesc(x0=x0s.ptr, x1=x1s.ptr if x1s else 0, gamma=self.w.ptr(rec["wname"] + ".g"),
                          beta=self.w.ptr(rec["wname"] + ".b"), stats=st, bstats=bst.ptr, dy=gy.ptr,
                          dx0=g0.ptr if g0 else 0, dx1=g1.ptr if g1 else 0, ldx0=x0s.ld, ldx1=x1s.ld if x1s else 0,
                          c0=x0.C, c1=x1.C if x1 is not None else 0, batch=self.nb, hw=x0.HW, groups=G, lddy=gy.ld,
                          lddx0=g0.ld if g0 else 0, lddx1=g1.ld if g1 else 0, eps=rec["eps"], act=rec["act"],
                          accumulate0=1 if a0 else 0, accumulate1=1 if a1 else 0, bpartial=bpart.ptr, bticket=btick.ptr)
        self.prog.add(lib.OP_GN_BWD_STATS, d, "bwd." + rec["name"] + ".stats")
        self.prog.add(lib.OP_GN_BWD_APPLY, d, "bwd." + rec["name"] + ".apply")

    def _b_ln(self, rec):
        x, y = rec["x"], rec["out"]
        if not self.has_grad(y):
            return
        gy, _ = self.grad(y, write=False)
        gx, acc = self.grad(x)
        xs = self._sl(x)
        mr = rec["mr"].ptr + 4 * 2 * self.b0 * x.HW
        d = lib.LnBwdDesc(x=xs.ptr, gamma=self.w.ptr(rec["wname"] + ".g"), dy=gy.ptr, mean_rstd=mr, dx=gx.ptr,
                          M=xs.M, C=x.C, ldx=xs.ld, lddy=gy.ld, lddx=gx.ld, accumulate=1 if acc else 0)
        self.prog.add(lib.OP_LAYERNORM_BWD, d, "bwd." + rec["name"])

    def _b_geglu(self, rec):
        pre, out = rec["pre"], rec["out"]
        if not self.has_grad(out):
            return
        go, _ = self.grad(out, write=False)
        gp, acc = self.grad(pre)
        assert not acc
        ps = self._sl(pre)
        self._ew(lib.EW_GEGLU_BWD, ps, gp, go, "bwd." + rec["name"], C=out.C)

    def _transpose(self, src: Act, heads: int, T: int, name: str, forward_data: bool = False):
        """forward_data: src is an activation of the forward pass (final before the backward starts): the transpose joins
        the one batched launch at the head of the program instead of sitting in the chain."""
        D = src.C // heads
        ldt = (T + 63) // 64 * 64
        t = self.arena.alloc((self.nb, heads, (D + 63) // 64 * 64, ldt), torch.bfloat16, name)
        d = lib.TransposeDesc(src=src.ptr, dst=t.ptr, B=self.nb, H=heads, T=T, ld=src.ld, ldt=ldt, D=D)
        if forward_data and self.batched:
            self._tr_batch.append(d)
        else:
            self.prog.add(lib.OP_TRANSPOSE_HEADS, d, name)
        return t, ldt

    def _b_attn(self, rec):
        q, k, v, o = rec["q"], rec["k"], rec["v"], rec["o"]
        if not self.has_grad(o):
            return
        heads, Tk, Tq = rec["heads"], rec["Tk"], q.HW
        go, _ = self.grad(o, write=False)
        need_dkv = 1 if (k.buf.ptr not in self.f.nograd_kv) else 0
        qs, ks, vs, os_ = self._sl(q), self._sl(k), self._sl(v), self._sl(o)
        kt, ldkt = self._transpose(ks, heads, Tk, "bwd." + rec["name"] + ".kt", forward_data=True)
        gq, aq = self.grad(q)
        assert not aq
        delta = self.arena.alloc((self.nb * heads * Tq + 64,), torch.float32, "bwd." + rec["name"] + ".delta")
        d = lib.AttnBwdDesc(q=qs.ptr, k=ks.ptr, v=vs.ptr, o=os_.ptr, d_o=go.ptr, kt=kt.ptr,
                            lse=rec["lse"].ptr + 4 * self.b0 * heads * Tq, delta=delta.ptr, dq=gq.ptr,
                            B=self.nb, H=heads, Tq=Tq, Tk=Tk, ldq=qs.ld, ldk=ks.ld, ldv=vs.ld, ldo=os_.ld, lddo=go.ld,
                            ldkt=ldkt, lddq=gq.ld, scale=(q.C // heads) ** -0.5, need_dkv=need_dkv, D=q.C // heads)
        if need_dkv:
            qt, ldqt = self._transpose(qs, heads, Tq, "bwd." + rec["name"] + ".qt", forward_data=True)
            made = self._dot_made.get(o.buf.ptr)
            if made is not None:
                dot = made[0]                 # written head-transposed by the product that produced dO
                assert made[1] == ldqt
            else:
                dot, _ = self._transpose(go, heads, Tq, "bwd." + rec["name"] + ".dot")
            gk, ak = self.grad(k)
            gv, av = self.grad(v)
            assert not ak and not av
            d.qt, d.dot, d.ldqt, d.dk, d.dv, d.lddk, d.lddv = qt.ptr, dot.ptr, ldqt, gk.ptr, gv.ptr, gk.ld, gv.ld
        self.prog.add(lib.OP_ATTN_BWD, d, "bwd." + rec["name"])

    def _splitk(self, d, name):
        d.tile = tuned_tile(d)
        provision_splitk(self, d, name)

    def _wgrad(self, d, name: str, defer: bool):
        if defer and self.batched:
            self._wg_batch[d.R].append(d)
        else:
            if self.wgrad_fixed_order:
                nb = lib.wgrad_single_blocks(d)
                d.slabs = self.arena.alloc((nb, 256 * d.R), torch.float32, name + ".slabs").ptr
                d.tickets = self.zarena.alloc((nb,), torch.int32, name + ".tickets").ptr
            self.prog.add(lib.OP_WGRAD, d, name)

    def _b_gemm(self, rec):
        y = rec["out"]
        if not self.has_grad(y):
            return
        f = self.f
        name = "bwd." + rec["name"]
        gy, _ = self.grad(y, write=False)
        x0, x1 = _src_parts(rec["x"])
        conv, grp, N, K = rec["conv"], rec["grp"], rec["N"], rec["K"]
        Ho, Wo = rec["Ho"], rec["Wo"]
        Ms = self.nb * Ho * Wo
        # residual branch: d(out)/d(residual) = identity
        r = rec["residual"]
        if r is not None and r.buf.ptr not in f.nograd:
            if r.buf.ptr not in self._g and r.ld == r.C and gy.ld == gy.C and r.C == gy.C:
                self.alias_grad(r, gy)
            else:
                self.add_into(r, gy, name + ".res")
        # time-embedding add: gradient w.r.t. the per-sample bias feeds the time_emb_proj adapter
        if rec.get("temb_path") and self.lora.temb_entries:
            path = rec["temb_path"]
            i = f.w.resnet_paths.index(path)
            e = self.lora.temb_entries[i]
            gsum = self.zarena.alloc((self.nb, N), torch.float32, name + ".gtemb")
            self._ew(lib.EW_COLSUM, gy, Act(gsum.ptr, self.nb, 1, 1, N, N, gsum), name=name + ".colsum", iarg2=Ho * Wo)
            ted = self.cfg.time_embed_dim
            L4 = f.temb_lora_T.shape[1]
            for s in range(self.nb):
                d = lib.TembLoraBwdDesc(g=gsum.ptr + 4 * s * N, t=f.temb_lora_T.ptr + 4 * ((self.b0 + s) * L4 + 4 * i),
                                        up=self.lora.up_ptr(e), emb=f.emb.ptr + 2 * (self.b0 + s) * ted,
                                        d_up=self.lora.gup_ptr(e), d_down=self.lora.gdown_ptr(e), scale=self.scale_ptr,
                                        C=N, ted=ted)
                self.prog.add(lib.OP_TEMB_LORA_BWD, d, name + ".temb_lora")
        # LoRA: U = dY . B_up per fused member; dB = s dY^T T ; dA = s U^T X
        U = None
        fused_u = False
        dA_after = []
        need0 = x0.buf.ptr not in f.nograd
        need1 = x1 is not None and x1.buf.ptr not in f.nograd
        if grp is not None:
            ng = len(grp)
            Ng = N // ng
            U = self.arena.alloc((Ms, 4 * ng), torch.float32, name + ".U")
            T = rec["T"]
            Ts = T.ptr + 4 * self.b0 * (Ho * Wo) * 4 * ng
            # U = dY . B (the up matrices as a rank-4 down-projection of the output gradient): inside the backward-data GEMM
            # of a dense module - third operand tile = the k-major copy of B (LoraStore.up_t), written out through lora_t_out
            # for the down-gradient - unless that product does not exist (no gradient needed upstream)
            up_t_off = self.lora.up_t_offset(grp) if (self.fuse_u and conv is None and x1 is None and (need0 or need1)) else None
            if up_t_off is not None:
                fused_u = True          # (split-K included: the slice that arrives last reduces T as well)
            if not fused_u:
                for g_i, e in enumerate(grp):
                    d = lib.SkinnyDesc(a0=gy.ptr + 2 * g_i * Ng, w=self.lora.up_ptr(e), out=U.ptr + 4 * 4 * g_i, lda0=gy.ld,
                                       ca0=Ng, mode=0, stride=1, M=Ms, R=4, K=Ng, ldo=4 * ng, w_kmajor=1)
                    self.prog.add(lib.OP_SKINNY, d, name + f".U{g_i}")
            d = lib.WgradDesc(z0=gy.ptr, v=Ts, out=self.lora.gup_ptr(grp[0]), scale=self.scale_ptr, ldz0=gy.ld, c0=N,
                              mode=0, stride=1, M=Ms, R=4, ldv=4 * ng, ldo=4, out_rmajor=0,
                              vgroup_cols=Ng if ng > 1 else 0)
            # dY of a module with a residual input may be the residual's gradient buffer too (alias_grad above): it keeps
            # accumulating after this point, so its up-gradient cannot wait for the batched launch at the tail
            self._wgrad(d, name + ".dB", defer=r is None or r.buf.ptr in f.nograd)
            x0s = self._sl(x0)
            x1s = self._sl(x1) if x1 is not None else None
            # the down matrices of a fused q|k|v group are adjacent ([12][K]) and so are their U columns: one R = 12 launch
            # reads X once instead of three times
            one = ng == 3 and conv is None
            for g_i, e in enumerate(grp[:1] if one else grp):
                d = lib.WgradDesc(z0=x0s.ptr, z1=x1s.ptr if x1s else 0, v=U.ptr + 4 * 4 * g_i, out=self.lora.gdown_ptr(e),
                                  scale=self.scale_ptr, ldz0=x0s.ld, ldz1=x1s.ld if x1s else 0, c0=x0.C,
                                  c1=x1.C if x1 is not None else 0, mode=0, stride=1, M=Ms, R=12 if one else 4, ldv=4 * ng,
                                  ldo=K, out_rmajor=1, vgroup_cols=0)
                if conv is not None:
                    d.mode, d.batch, d.hs, d.ws = 1, self.nb, x0.H, x0.W
                    d.src_xform, d.stride, d.ho, d.wo = conv.get("xform", 0), conv.get("stride", 1), Ho, Wo
                if fused_u:
                    dA_after.append((d, name + f".dA{g_i}"))            # U is written by the dgrad launch below
                else:
                    self._wgrad(d, name + f".dA{g_i}", defer=True)      # forward activations and U: final
        # backward data
        if not (need0 or need1):
            return
        cin = x0.C + (x1.C if x1 is not None else 0)
        wT = self.w.ptr(rec["wname"] + ".wT")
        gyimg = Act(gy.ptr, self.nb, Ho, Wo, N, gy.ld, gy.buf, gy.name)
        if conv is None:
            geglu_rec = self._geglu_of.get(x0.buf.ptr) if (x1 is None and grp is None and not self.has_grad(x0)) else None
            if geglu_rec is not None and x0.C % 32 == 0:
                # the Linear behind a GEGLU (ff.net.2): its backward-data product writes d(proj) directly (GEGLU backward in
                # the epilogue, slh_gemm_desc.geglu = 2) - no d(ff) tensor, no elementwise launch
                pre = geglu_rec["pre"]
                gp, pacc = self.grad(pre)
                assert not pacc
                ps = self._sl(pre)
                d = lib.GemmDesc(a0=gy.ptr, w=wT, c=gp.ptr, lda0=gy.ld, ca0=N, mode=0, stride=1, ldw=N, M=Ms, N=cin, K=N,
                                 ldc=gp.ld, rows_per_sample=Ho * Wo, w_layout=1 if self.w.packed else 0)
                self._splitk(d, name)
                d.geglu, d.geglu_pre, d.ld_pre = 2, ps.ptr, ps.ld
                settle_tile(d)
                self.prog.add(lib.OP_GEMM, d, name + ".dgrad")
                return
            if x1 is None:
                gx, acc = self.grad(x0)
                tgt, tacc = gx, acc
            else:
                tb = self.arena.alloc((Ms, cin), torch.bfloat16, name + ".gxcat")
                tgt, tacc = Act(tb.ptr, self.nb, x0.H, x0.W, cin, cin, tb), False
            d = lib.GemmDesc(a0=gy.ptr, w=wT, c=tgt.ptr, residual=tgt.ptr if tacc else 0, lda0=gy.ld, ca0=N, mode=0,
                             stride=1, ldw=N, M=Ms, N=cin, K=N, ld_res=tgt.ld, ldc=tgt.ld, rows_per_sample=Ho * Wo,
                             w_layout=1 if self.w.packed else 0)
            if grp is not None:
                d.ld_t, d.lora_up, d.lora_scale = 4 * len(grp), self.lora.down_ptr(grp[0]), self.scale_ptr
                d.lora_groups, d.lora_rank, d.lora_up_rmajor = 1, 4 * len(grp), 1
                if fused_u:
                    self._uses_up_t = True
                    base = 0x2000 if self.arena.virtual else self.lora.up_t.data_ptr()
                    d.lora_down, d.lora_t_out = base + 2 * up_t_off, U.ptr
                else:
                    d.lora_t = U.ptr
            self._splitk(d, name)
            want = self._wants_dot.get(x0.buf.ptr) if (x1 is None and not tacc and tgt.ptr == gx.ptr) else None
            if want is not None:
                heads, Tq = want
                Dh = cin // heads
                if Dh % 64 == 0 and Tq % 64 == 0 and Ms == self.nb * Tq and cin == x0.C and tgt.ld % 8 == 0:
                    dot = self.arena.alloc((self.nb, heads, Dh, Tq), torch.bfloat16, name + ".dot")
                    d.vt_out, d.vt_col0, d.vt_D, d.vt_heads, d.vt_tokens, d.vt_ld = dot.ptr, 0, Dh, heads, Tq, Tq
                    d.vt_also_c = 1
                    self._dot_made[x0.buf.ptr] = (dot, Tq)
            settle_tile(d)
            self.prog.add(lib.OP_GEMM, d, name + ".dgrad")
            for dA, nm in dA_after:
                self._wgrad(dA, nm, defer=True)
            if x1 is not None:
                if need0:
                    self.add_into(x0, tgt.cols(0, x0.C), name + ".gx0")
                if need1:
                    self.add_into(x1, tgt.cols(x0.C, x1.C), name + ".gx1")
        else:
            assert x1 is None
            xform, stride = conv.get("xform", 0), conv.get("stride", 1)
            if xform == 1:     # forward read a nearest-2x upsampled image: dgrad lands on the 2h x 2w grid first
                HL, WL = 2 * x0.H, 2 * x0.W
                tb = self.arena.alloc((self.nb * HL * WL, cin), torch.bfloat16, name + ".gx_up")
                tgt, tacc = Act(tb.ptr, self.nb, HL, WL, cin, cin, tb), False
            else:
                HL, WL = x0.H, x0.W
                tgt, tacc = self.grad(x0)
            d = lib.GemmDesc(a0=gy.ptr, w=wT, c=tgt.ptr, residual=tgt.ptr if tacc else 0, lda0=gy.ld, ca0=N, mode=1,
                             batch=self.nb, hs=Ho, ws=Wo, src_xform=2 if stride == 2 else 0, stride=1, ho=HL, wo=WL,
                             ldw=9 * N, M=self.nb * HL * WL, N=cin, K=9 * N, ld_res=tgt.ld, ldc=tgt.ld,
                             rows_per_sample=HL * WL, w_layout=1 if self.w.packed else 0)
            self._splitk(d, name)
            settle_tile(d)
            self.prog.add(lib.OP_GEMM, d, name + ".dgrad")
            if grp is not None:
                d2 = lib.LoraCdgradDesc(u=U.ptr, a_down=self.lora.down_ptr(grp[0]), scale=self.scale_ptr, gx=tgt.ptr,
                                        batch=self.nb, hl=HL, wl=WL, ho=Ho, wo=Wo, stride=stride, cin=cin, ldu=4,
                                        ldgx=tgt.ld, accumulate=1)
                self.prog.add(lib.OP_LORA_CONV_DGRAD, d2, name + ".lora_dgrad")
            if xform == 1:
                gx, acc = self.grad(x0)
                if acc:
                    t2b = self.arena.alloc((self.nb * x0.HW, cin), torch.bfloat16, name + ".gx_dn")
                    t2 = Act(t2b.ptr, self.nb, x0.H, x0.W, cin, cin, t2b)
                    self._ew(lib.EW_UPSAMPLE_BWD, tgt, t2, name=name + ".upsample_bwd", iarg=x0.W, iarg2=x0.HW, M=t2.M)
                    self._ew(lib.EW_ADD, gx, gx, t2, name + ".add")
                else:
                    self._ew(lib.EW_UPSAMPLE_BWD, tgt, gx, name=name + ".upsample_bwd", iarg=x0.W, iarg2=x0.HW, M=gx.M)
