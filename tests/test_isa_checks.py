"""Static checks on the compiled gfx950 code (CPU-only: hipcc cross-compiles, nothing is executed).

LDS-DMA (global_load_lds) completion is tracked by vmcnt, and __syncthreads() does not wait on it; a prefetching loop
that forgets the explicit wait reads tiles that have not landed - a timing-dependent corruption that no parity test
catches reliably.  scripts/check_lds_dma_waits.py proves on the control-flow graph of every LDS-DMA kernel that each
s_barrier is preceded by a vmcnt wait on all paths."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import check_lds_dma_waits as chk  # noqa: E402

CSRC = os.path.join(ROOT, "sliders_amd", "csrc")
_ASM = {}


def _asm(src):
    if src not in _ASM:
        _ASM[src] = chk.device_asm(os.path.join(CSRC, src))
    return _ASM[src]

RACY = """
    s_load_dwordx2 s[0:1], s[4:5], 0x0
    global_load_lds_dwordx4 v[2:3], off
    s_waitcnt vmcnt(0)
.LBB0_1:
    s_waitcnt lgkmcnt(0)
    s_barrier
    global_load_lds_dwordx4 v[2:3], off
    ds_read_b128 v[4:7], v8
    s_cbranch_scc1 .LBB0_3
    s_branch .LBB0_1
.LBB0_3:
    s_endpgm
"""


def test_checker_flags_a_loop_carried_unwaited_dma():
    hits = chk.check_kernel(RACY)
    assert len(hits) == 1 and hits[0][1].startswith("s_barrier")
    fixed = RACY.replace("    s_waitcnt lgkmcnt(0)\n    s_barrier", "    s_waitcnt vmcnt(0)\n    s_barrier")
    assert chk.check_kernel(fixed) == []


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src", ["attention.hip", "attention_bwd.hip", "gemm.hip"])
def test_every_barrier_waits_for_lds_dma(src):
    asm = _asm(src)
    checked = 0
    for name, body in chk.kernels(asm):
        if "global_load_lds" not in body or chk.is_counted_ring(name):
            continue
        checked += 1
        assert chk.check_kernel(body) == [], name
    assert checked >= 6


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src", sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")))
def test_no_kernel_spills_to_scratch(src):
    """every kernel of the library fits its registers: a spill in a hot loop is a silent 2-10x (seen once this round:
    an epilogue preload pushed one GEMM instantiation into scratch), and the launch-bounds / tile choices assume it"""
    res = chk.kernel_resources(_asm(src))
    assert res, src
    for name, r in res.items():
        assert r["scratch"] == 0, (name, r)
        assert r["lds"] <= 160 * 1024, (name, r)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_row_kernels_issue_their_loads_together():
    """LayerNorm forward / backward read a row as up to three 16-byte chunks per lane (+ gamma / beta / dy): as loads behind
    `if (c < C)` hipcc emitted load / s_waitcnt vmcnt(0) three times in a row - three serial round trips per row.  The kernels
    load unconditionally at clamped columns; this guards the compiled code against the pattern coming back."""
    asm = _asm("norm.hip")
    seen = 0
    for name, body in chk.kernels(asm):
        if "layernorm" not in name:
            continue
        seen += 1
        sites, loads = chk.serial_load_sites(body)
        assert loads >= 9 and sites == 0, (name, sites, loads)
    assert seen == 2


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src", sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")))
def test_no_kernel_waits_on_another_workgroup(src):
    """The rule since round 5 (round 4's stream-K tile polled flags of other workgroups and could hang two concurrent launches): no
    kernel of the library waits for another workgroup.  Every cross-workgroup reduction is the last-arriver form - one relaxed
    fetch-add on a ticket, whoever draws the last ticket does the combining, nobody polls.  Two cheap guards against the pattern
    coming back: the compiled code of every kernel is free of `s_sleep` (what a polite polling loop is made of), and no source
    loop has an atomic load in its condition."""
    import re
    for name, body in chk.kernels(_asm(src)):
        assert "s_sleep" not in body, f"{name}: s_sleep (a polling wait?)"
    text = open(os.path.join(CSRC, src)).read() + "".join(open(os.path.join(CSRC, h)).read() for h in ("common.h", "gemm_common.h"))
    text = re.sub(r"//[^\n]*", "", text)
    assert not re.search(r"\bwhile\s*\([^;{]*__hip_atomic_load", text), f"{src}: a loop polls an atomic"
    assert "__builtin_amdgcn_s_sleep" not in text
