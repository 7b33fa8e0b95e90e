"""Measured tile choices for slh_gemm (written by scripts/tune_gemm.py on an MI355X).

The C library has a fill-the-chip heuristic; the planner overrides it with the measured best tile for every
(shape, addressing mode) it has an entry for.  Tables live in sliders_amd/tuning/*.json and are merged.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Optional

_TABLE: Optional[Dict[str, int]] = None


def gemm_key(d, with_lora: bool = False) -> str:
    k = f"{d.M},{d.N},{d.K},m{d.mode},s{d.stride},x{d.src_xform},g{d.geglu}"
    if with_lora:     # the fused adapter changes the LDS footprint and the MFMA count per k-step: tuned separately
        k += f",l{1 if d.lora_down else 0}"
        # folded LayerNorm: the producer side needs a 128-column tile, the consumer side carries a prologue - both tuned
        # apart from the plain product of the same shape
        if getattr(d, "ln_out", None):
            k += ",no"
        if getattr(d, "ln_in", None):
            k += ",ni"
    return k


def tile_ok(d, tile: int) -> bool:
    """Can slh_gemm run descriptor d with this tile code?  (The constraints slh_gemm itself checks: used by the tuner to skip
    candidates and by the planner to drop a table entry that no longer fits the launch it is looked up for.)"""
    mi, ni, wm = (tile >> 4) & 15, tile & 15, (tile >> 12) & 15
    if getattr(d, "xa_k", None):                  # cross-attention in the epilogue: the 128 x 128 ring tile only
        return tile == 0x4412
    if tile >> 20:                                # bits 20+ are reserved (round 4's stream-K form lived there; removed)
        return False
    if wm == 7:                                   # the four-wave tiles (csrc/gemm7.hip): the library's own rule, asked with this tile code
        from . import lib
        keep, d.tile = d.tile, tile
        try:
            return lib.gemm7_ok(d)
        finally:
            d.tile = keep
    if wm == 5:                                   # the 64 x 160 tile (csrc/gemm5.hip): the library's own rule
        from . import lib
        return not (tile >> 16) & 15 and lib.gemm5_ok(d)
    if getattr(d, "ln_in", None) and d.lora_down:     # LayerNorm fold + fused adapter: ping-pong 128 x 192 / 128 x 256, no split-K
        return wm == 8 and mi == 1 and ni in (3, 4) and not (tile >> 16) & 15 and \
            (not d.vt_out or (ni == 4 and d.mode == 0))
    if getattr(d, "ln_in", None) and (tile >> 16) & 15 > 1:      # a folded LayerNorm excludes split-K
        return False
    if wm == 8 and (tile >> 16) & 15:             # split-K: the slabs of these tiles must fit the workspace contract
        bm, bn = (256, 256) if mi == 4 else (128 * mi, 64 * ni)
        r = lambda v, q: (v + q - 1) // q * q
        if r(d.M, bm) * r(d.N, bn) > r(d.M, 256) * r(d.N, 128):
            return False
    if wm == 8:                                   # ping-pong K loops (csrc/gemm8p.hip)
        if (mi, ni) == (4, 2):                    # 256 x 256: no fused adapter
            return not d.lora_down and d.geglu in (0, 1, 2, 3)
        if mi != 1 or ni < 3 or ni > 5:
            return False
        if d.geglu in (1, 2) or d.ln_out:                 # 32 | 32 GEGLU blocks and the chunk statistics assume NI = 2
            return False
        if d.vt_out and not (ni == 4 and d.mode == 0):    # the V^T store: a wave's columns must not straddle vt_col0 (128 x 256 only)
            return False
        return True
    if not tile:
        return True
    if d.geglu in (1,) and ni != 2:
        return False
    if d.geglu == 3:
        return True
    if d.ln_out and ni != 2:
        return False
    return True


def settle_tile(d) -> int:
    """Last check before a GEMM descriptor is recorded: the table lookup (tuned_tile) happens while the descriptor is still being
    filled in - the planner attaches V^T / dO^T stores, the GEGLU backward and training outputs AFTER it knows the tile - so an entry
    of the 64 x 160 family (which names a shape, not a feature set) can end up on a launch that tile has no epilogue for.  Such a
    launch goes back to the 128 x 128 ring tile the entry replaced (or to the library heuristic).  Returns the tile that will run."""
    t = d.tile
    if t and (t >> 12) & 15 in (5, 7) and not tile_ok(d, t):
        d.tile = 0x4412 if tile_ok(d, 0x4412) else 0
    return d.tile


def table() -> Dict[str, int]:
    global _TABLE
    if _TABLE is None:
        _TABLE = {}
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning")
        paths = sorted(glob.glob(os.path.join(here, "*.json")))
        if os.environ.get("SLIDERS_TUNING_OVERRIDE"):          # same-box A/B of table variants: entries of this file win
            paths.append(os.environ["SLIDERS_TUNING_OVERRIDE"])
        for path in paths:
            with open(path) as f:
                _TABLE.update({k: int(v) for k, v in json.load(f).items()})
    return _TABLE


def tuned_tile(d) -> int:
    """0 = no entry (library heuristic)."""
    if os.environ.get("SLIDERS_NO_TUNING"):
        return 0
    tb = table()
    full = gemm_key(d, True)
    base = full.replace(",no", "").replace(",ni", "")          # entries measured before the LayerNorm folding existed
    t = tb.get(full, tb.get(base, tb.get(gemm_key(d), 0)))
    if not t and d.geglu == 3:     # GEGLU in 16 | 16 blocks takes any tile: unmeasured shapes run the tile measured for the 32 | 32 form
        g1 = lambda k: k.replace(",g3", ",g1")
        t = tb.get(g1(full), tb.get(g1(base), tb.get(g1(gemm_key(d)), 0)))
    if not t and d.lora_down:      # adapter fused in but only the plain product was measured (backward-data GEMMs): same tile
        t = tb.get(base[:-1] + "0", 0)
    if t and not tile_ok(d, t):
        # an entry for the 64 x 160 tile names a shape, not a feature set (the key does not see row bias, V^T stores, training outputs):
        # where the launch needs more than that tile's epilogue offers, the 128 x 128 ring tile it replaced runs
        t = 0x4412 if (t >> 12) & 15 in (5, 7) and tile_ok(d, 0x4412) else 0
    force = os.environ.get("SLIDERS_FORCE_STAGES")     # experiment knob: 2 or 3 for every non-128x128 tile
    if force and t and (t & 0xFF) != 0x22:
        t = (t & 0xFF) | (int(force) << 8 if force == "3" else 0)
    return t
