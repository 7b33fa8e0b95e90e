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


def table() -> Dict[str, int]:
    global _TABLE
    if _TABLE is None:
        _TABLE = {}
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning")
        for path in sorted(glob.glob(os.path.join(here, "*.json"))):
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
    if not t and d.lora_down:      # adapter fused in but only the plain product was measured (backward-data GEMMs): same tile
        t = tb.get(base[:-1] + "0", 0)
    force = os.environ.get("SLIDERS_FORCE_STAGES")     # experiment knob: 2 or 3 for every non-128x128 tile
    if force and t and (t & 0xFF) != 0x22:
        t = (t & 0xFF) | (int(force) << 8 if force == "3" else 0)
    return t
