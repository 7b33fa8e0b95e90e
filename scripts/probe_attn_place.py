"""(Round 6: the SLH_ATTN_TRACE branches this script reads were removed from csrc/attention.hip; build the variant from
`git show 46d9a74:sliders_amd/csrc/attention.hip` to re-run it.)  Where the dispatcher puts the attention workgroups (development aid; GPU box): run with the SLH_ATTN_TRACE build
(SLIDERS_HIP_LIB=.../libsliders_hip_trace.so), prints waves per SIMD / workgroups per CU histograms and the launch time.
Knob of the launcher: SLH_ATTN_NW2=1 (64-query workgroups everywhere).  (SLH_ATTN_SPREAD, an LDS cap on workgroups per CU,
was removed after this probe showed the placement is already even: profiles/r03_attn_variants.txt.)"""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd import lib

dev = torch.device("cuda:0")
lib.load()
s = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
D = 64
trace = "trace" in os.environ.get("SLIDERS_HIP_LIB", "")
print(os.environ.get("SLIDERS_HIP_LIB", "default lib"), "spread", os.environ.get("SLH_ATTN_SPREAD"), "nw2", os.environ.get("SLH_ATTN_NW2"))
for T, BH in ((4096, 16), (4096, 20), (4096, 24), (1024, 40), (1024, 64), (1024, 96)):
    B, H = 1, BH
    C = H * D
    q = torch.randn(B * T, C, device=dev).to(torch.bfloat16)
    k = torch.randn(B * T, C, device=dev).to(torch.bfloat16)
    vt = torch.randn(B, H, D, T, device=dev).to(torch.bfloat16)
    o = torch.empty(B * T, C, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B * H * T, device=dev, dtype=torch.float32)
    d = lib.AttnDesc(q=p(q), k=p(k), vt=p(vt), o=p(o), lse=p(lse), B=B, H=H, Tq=T, Tk=T, ldq=C, ldk=C, ldvt=T, ldo=C,
                     scale=D ** -0.5, D=D)
    for _ in range(3):
        lib.call(lib.OP_ATTN_FWD, d, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.call(lib.OP_ATTN_FWD, d, s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    line = f"T {T} BH {BH}: {us:7.1f} us"
    if trace:
        nw = 2 if os.environ.get("SLH_ATTN_NW2") or BH * T // 128 < 256 else 4
        nwaves = BH * T // 32
        tr = lse.view(torch.int32)[: nwaves * 16].view(nwaves, 16).cpu().numpy().astype("int64") & 0xFFFFFFFF
        ph = tr[:, 4:9].mean(axis=0) / (T // 64)
        line += "  cycles per tile [barrier, stage, QK, softmax, PV] = " + " ".join(f"{x:.0f}" for x in ph) + f" (sum {ph.sum():.0f})"
        hw, xcc, t0, dt = tr[:, 0], tr[:, 1] & 15, tr[:, 2], tr[:, 3]
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        per_simd = Counter(zip(cukey.tolist(), simd.tolist()))
        per_cu = Counter(cukey.tolist())
        # only waves of the first round (started before the first wave ended) count as co-resident
        first_end = (t0 + dt).min()
        resident = t0 < first_end
        per_simd_res = Counter(zip(cukey[resident].tolist(), simd[resident].tolist()))
        line += (f"  CUs used {len(per_cu)}  waves/CU hist {sorted(Counter(per_cu.values()).items())}  waves/SIMD hist "
                 f"{sorted(Counter(per_simd.values()).items())}  co-resident at start {sorted(Counter(per_simd_res.values()).items())}"
                 f"  wave time min/med/max {dt.min() / 100:.0f}/{sorted(dt)[len(dt) // 2] / 100:.0f}/{dt.max() / 100:.0f} us")
    print(line, flush=True)
