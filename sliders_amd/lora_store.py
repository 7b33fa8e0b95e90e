"""Packed storage of the LoRA adapter parameters (the only trainable state of the hot path).

Reference: `LoRAModule` keeps `lora_down` / `lora_up` as separate nn.Linear / nn.Conv2d per target
(trainscripts/textsliders/lora.py:55-101), 346 modules = 692 tiny bf16 tensors for SDXL.  Here all of them
live in ONE flat bf16 buffer (plus one flat fp32 gradient buffer and two flat bf16 AdamW moment buffers), so
the optimizer step is one launch and the data-parallel exchange is one RCCL all-reduce of one buffer.

Kernel-side layouts inside the flat buffer
  down, Linear : [r][in]                 (= lora_down.weight)
  down, Conv3x3: [r][tap][Cin]           (lora_down.weight (r,Cin,3,3) permuted; implicit-GEMM K order)
  up           : [out][r]                (= lora_up.weight, conv (out,r,1,1) flattened)
Modules that the planner fuses into one GEMM are adjacent: attn1 (to_q,to_k,to_v) downs form [3r][C] and
their ups [3C][r]; attn2 (to_k,to_v) likewise; every ResnetBlock2D.time_emb_proj down/up is concatenated
into one [r*L][temb] / [sum(Cout)][r] pair (one GEMV per UNet pass).

`state_dict()` / `load_state_dict()` convert to / from the reference checkpoint layout (key names and shapes
of lora.py:231-248) so files are interchangeable with the reference's inference notebooks.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .config import UNetConfig
from .modules import LoraTarget, lora_targets, lora_visits


@dataclass
class LoraEntry:
    target: LoraTarget
    down_off: int = -1
    up_off: int = -1
    trainable: bool = True

    @property
    def name(self) -> str:
        return self.target.lora_name

    @property
    def k_dim(self) -> int:  # contraction length of the down projection
        t = self.target
        return t.in_dim * (9 if t.kind == "conv3" else 1)

    @property
    def down_numel(self) -> int:
        return self.target.rank * self.k_dim

    @property
    def up_numel(self) -> int:
        return self.target.out_dim * self.target.rank


class LoraStore:
    def __init__(self, cfg: UNetConfig, rank: int = 4, alpha: float = 1.0, train_method: str = "noxattn",
                 network_type: str = "c3lier", device="cpu", init: str = "reference", kaiming_a: float = 1.0,
                 state_dtype: torch.dtype = torch.bfloat16):
        """state_dtype: dtype of the adapter parameters and optimizer moments - the reference's weight_dtype as it is applied to the
        network (train_lora_xl.py:60-61, 84-90: network.to(device, dtype=weight_dtype)).  bfloat16 (the reference's default): one flat
        bf16 buffer is both the optimizer's state and what the kernels read.  float32: `master` (fp32 parameters) and fp32 moments are
        the optimizer's state; `params` stays the bf16 buffer the UNet kernels read, rewritten from the master by the optimizer kernel
        in the same launch."""
        if state_dtype not in (torch.bfloat16, torch.float32):
            raise NotImplementedError(f"adapter state dtype {state_dtype}: bfloat16 or float32")
        self.state_dtype = state_dtype
        self.cfg = cfg
        self.rank = rank
        self.alpha = rank if alpha is None or alpha == 0 else alpha
        self.train_method = train_method
        self.network_type = network_type
        self.device = device
        targets = lora_targets(cfg, train_method, rank, network_type)
        for t in targets:
            if t.rank != rank:
                raise NotImplementedError(f"{t.lora_name}: rank clipped to {t.rank}; uniform rank required")
        self.entries: List[LoraEntry] = [LoraEntry(t) for t in targets]
        self.by_path: Dict[str, LoraEntry] = {e.target.module_path: e for e in self.entries}
        self.by_name: Dict[str, LoraEntry] = {e.name: e for e in self.entries}
        self.scale = self.alpha / self.rank
        self._layout()
        n = self.numel
        self.params = torch.zeros(n, dtype=torch.bfloat16, device=device)
        self.grads = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(n, dtype=state_dtype, device=device)
        self.exp_avg_sq = torch.zeros(n, dtype=state_dtype, device=device)
        self.master = torch.zeros(n, dtype=torch.float32, device=device) if state_dtype == torch.float32 else None
        self.opt_step = 0
        self._build_up_t()
        if init == "reference":
            self.init_reference(kaiming_a)

    # ---- layout -------------------------------------------------------------------------------------
    def _layout(self):
        off = 0
        placed = set()
        self._up_t_slots: Dict[tuple, tuple] = {}
        self._up_t_numel = 0
        self.temb_entries: List[LoraEntry] = [e for e in self.entries if e.target.module_path.endswith("time_emb_proj")]
        by_path = self.by_path

        def place_group(group: List[LoraEntry]):
            nonlocal off
            for e in group:  # downs first (contiguous), then ups
                e.down_off = off
                off += e.down_numel
            for e in group:
                e.up_off = off
                off += e.up_numel
            for e in group:
                placed.add(e.name)
            if all(e.target.kind in ("linear", "conv1") for e in group):
                ntot = sum(e.target.out_dim for e in group)
                self._up_t_slots[tuple(e.name for e in group)] = (self._up_t_numel, group, ntot)
                self._up_t_numel += (4 * len(group) * ntot + 7) // 8 * 8

        for e in self.entries:
            if e.name in placed or e in self.temb_entries:
                continue
            p = e.target.module_path
            if p.endswith(".to_q") and ".attn1." in p + ".":
                base = p[: -len(".to_q")]
                grp = [by_path.get(base + s) for s in (".to_q", ".to_k", ".to_v")]
                if all(g is not None for g in grp):
                    place_group(grp)
                    continue
            if p.endswith(".to_k") and ".attn2." in p + ".":
                base = p[: -len(".to_k")]
                grp = [by_path.get(base + s) for s in (".to_k", ".to_v")]
                if all(g is not None for g in grp):
                    place_group(grp)
                    continue
            place_group([e])
        # all time_emb_proj adapters: one contiguous [r*L][temb] down block and one [sum Cout][r] up block
        self.temb_down_off = off
        for e in self.temb_entries:
            e.down_off = off
            off += e.down_numel
        self.temb_up_off = off
        for e in self.temb_entries:
            e.up_off = off
            off += e.up_numel
        # keep every block 16-byte aligned: all numels are multiples of 8 bf16 here (rank 4 * dims % 2 == 0)
        for e in self.entries:
            assert e.down_off % 8 == 0 and e.up_off % 4 == 0, e.name
        self.numel = off

    # ---- k-major copies of the up matrices (backward-data products with the adapter fused in) ---------
    def up_t_offset(self, grp: List[LoraEntry]) -> Optional[int]:
        """Element offset, inside self.up_t, of the [4 * len(grp)][sum out_dim] matrix whose row 4g + r holds column r of
        member g's up matrix over that member's own output columns and zeros elsewhere (block-diagonal for a fused q|k|v
        group): U = dY . B, a down-projection of the output gradient, rides in the backward-data GEMM as its third operand
        (planner.BackwardPlan._b_gemm).  One slot per dense group of the layout; None for convolutions / unknown groups."""
        slot = self._up_t_slots.get(tuple(e.name for e in grp))
        return None if slot is None else slot[0]

    def _build_up_t(self):
        """up_t (bf16, refreshed from the live parameters by one slh_gather16 at the head of every backward) and its int32
        gather index into self.params (-1 = structural zero)."""
        idx = torch.full((max(self._up_t_numel, 1),), -1, dtype=torch.int32)
        for off, grp, ntot in self._up_t_slots.values():
            col = 0
            for g, e in enumerate(grp):
                n = e.target.out_dim
                src = e.up_off + torch.arange(n, dtype=torch.int32)[None, :] * 4 + torch.arange(4, dtype=torch.int32)[:, None]
                rows = off + (4 * g + torch.arange(4)[:, None]) * ntot + col + torch.arange(n)[None, :]
                idx[rows.reshape(-1).long()] = src.reshape(-1)
                col += n
        self.up_t_index = idx.to(self.device)
        self.up_t = torch.zeros(max(self._up_t_numel, 1), dtype=torch.bfloat16, device=self.device)

    def fused_group(self, paths: List[str]) -> Optional[List[LoraEntry]]:
        """Entries for `paths` if they are all adapted AND stored adjacently (downs then ups), else None."""
        grp = [self.by_path.get(p) for p in paths]
        if any(g is None for g in grp):
            return None
        for a, b in zip(grp, grp[1:]):
            if b.down_off != a.down_off + a.down_numel or b.up_off != a.up_off + a.up_numel:
                return None
        return grp

    # ---- pointers ------------------------------------------------------------------------------------
    def down_ptr(self, e: LoraEntry) -> int:
        return self.params.data_ptr() + 2 * e.down_off

    def up_ptr(self, e: LoraEntry) -> int:
        return self.params.data_ptr() + 2 * e.up_off

    def gdown_ptr(self, e: LoraEntry) -> int:
        return self.grads.data_ptr() + 4 * e.down_off

    def gup_ptr(self, e: LoraEntry) -> int:
        return self.grads.data_ptr() + 4 * e.up_off

    # ---- init / (de)serialisation ------------------------------------------------------------------
    def _down_to_kernel(self, e: LoraEntry, w: torch.Tensor) -> torch.Tensor:
        if e.target.kind == "conv3":
            return w.permute(0, 2, 3, 1).reshape(-1)
        return w.reshape(-1)

    def _down_from_kernel(self, e: LoraEntry, flat: torch.Tensor) -> torch.Tensor:
        t = e.target
        if t.kind == "conv3":
            return flat.view(t.rank, 3, 3, t.in_dim).permute(0, 3, 1, 2).contiguous()
        if t.kind == "conv1":
            return flat.view(t.rank, t.in_dim, 1, 1).contiguous()
        return flat.view(t.rank, t.in_dim).contiguous()

    def _up_shape(self, e: LoraEntry):
        t = e.target
        return (t.out_dim, t.rank) if t.kind == "linear" else (t.out_dim, t.rank, 1, 1)

    def init_reference(self, kaiming_a: float = 1.0):
        """Same RNG draws, in the same order, as the reference's network construction so that torch.manual_seed(s) gives
        the reference's initial adapter weights: LoRAModule.__init__ (lora.py:68-97) lets the nn.Linear / nn.Conv2d
        constructors draw their default init first, then kaiming_uniform_(down, a) and zeros_(up) - and create_modules
        (lora.py:206-216) does so for the duplicate visits of the conv leaves as well, before dropping them by name.
        Pinned by tests/golden/lora_init.json, produced by the reference's own LoRANetwork."""
        host = torch.zeros(self.numel, dtype=torch.float32)
        for t, dup in lora_visits(self.cfg, self.train_method, self.rank, self.network_type):
            if t.kind == "linear":
                down = nn.Linear(t.in_dim, t.rank, bias=False)
                up = nn.Linear(t.rank, t.out_dim, bias=False)
            else:
                k = 3 if t.kind == "conv3" else 1
                down = nn.Conv2d(t.in_dim, t.rank, (k, k), (t.stride, t.stride), (k // 2, k // 2), bias=False)
                up = nn.Conv2d(t.rank, t.out_dim, (1, 1), (1, 1), bias=False)
            nn.init.kaiming_uniform_(down.weight, a=kaiming_a)
            nn.init.zeros_(up.weight)
            if dup:
                continue                       # constructed (RNG advanced) and discarded, like the reference
            e = self.by_name[t.lora_name]
            host[e.down_off:e.down_off + e.down_numel] = self._down_to_kernel(e, down.weight.detach())
        self._set_params(host)

    def _set_params(self, host_f32: torch.Tensor):
        """New parameter values (fp32, host or device): the fp32 master when there is one, and the bf16 buffer the kernels read."""
        if self.master is not None:
            self.master.copy_(host_f32)
        self.params.copy_(host_f32.to(torch.bfloat16))

    def sync_master_from_params(self):
        """After writing `params` directly (tests, tools): make the fp32 master agree with it."""
        if self.master is not None:
            self.master.copy_(self.params.float())

    def state_dict(self, dtype: Optional[torch.dtype] = None) -> "OrderedDict[str, torch.Tensor]":
        """Reference key order per module: `<name>.alpha`, `<name>.lora_down.weight`, `<name>.lora_up.weight`."""
        host = (self.master if self.master is not None else self.params).detach().to("cpu")
        sd = OrderedDict()
        for e in self.entries:
            if not e.trainable:
                continue
            down = self._down_from_kernel(e, host[e.down_off:e.down_off + e.down_numel].clone())
            up = host[e.up_off:e.up_off + e.up_numel].clone().view(self._up_shape(e))
            alpha = torch.tensor(self.alpha)
            if dtype is not None:
                down, up, alpha = down.to(dtype), up.to(dtype), alpha.to(dtype)
            sd[f"{e.name}.alpha"] = alpha
            sd[f"{e.name}.lora_down.weight"] = down
            sd[f"{e.name}.lora_up.weight"] = up
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        host = (self.master if self.master is not None else self.params).detach().to("cpu").to(torch.float32)
        seen = set()
        for e in self.entries:
            kd, ku = f"{e.name}.lora_down.weight", f"{e.name}.lora_up.weight"
            if kd not in sd or ku not in sd:
                if strict:
                    raise KeyError(f"missing key(s) for {e.name}")
                continue
            host[e.down_off:e.down_off + e.down_numel] = self._down_to_kernel(e, sd[kd].to(torch.float32))
            host[e.up_off:e.up_off + e.up_numel] = sd[ku].to(torch.float32).reshape(-1)
            seen.update((kd, ku, f"{e.name}.alpha"))
        if strict:
            extra = set(sd.keys()) - seen
            if extra:
                raise KeyError(f"unexpected key(s): {sorted(extra)[:4]} ...")
        self._set_params(host)

    def n_trainable(self) -> int:
        return sum(e.down_numel + e.up_numel for e in self.entries if e.trainable)
