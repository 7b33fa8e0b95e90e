"""Attention forward: time against the number of resident waves per SIMD (development aid; GPU box).  T = 4096 and 1024 at
D = 64 with the (sample, head) count chosen so the grid is 1, 1.25, 2, 2.5, 3, 4, 5 workgroups of 4 waves per CU.
SLIDERS_HIP_LIB selects the library build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd import lib

dev = torch.device("cuda:0")
lib.load()
s = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
D = 64
print(os.environ.get("SLIDERS_HIP_LIB", "default lib"))
for T, bhs in ((4096, (8, 16, 20, 24, 32, 40)), (1024, (32, 40, 64, 80, 96, 128, 160)), (256, (128, 256))):
    for BH in bhs:
        B, H = 1, BH
        C = H * D
        nb = 3
        qs = [torch.randn(B * T, C, device=dev).to(torch.bfloat16) for _ in range(nb)]
        ks = [torch.randn(B * T, C, device=dev).to(torch.bfloat16) for _ in range(nb)]
        vts = [torch.randn(B, H, D, T, device=dev).to(torch.bfloat16) for _ in range(nb)]
        o = torch.empty(B * T, C, device=dev, dtype=torch.bfloat16)
        descs = [lib.AttnDesc(q=p(q), k=p(k), vt=p(vt), o=p(o), B=B, H=H, Tq=T, Tk=T, ldq=C, ldk=C, ldvt=T, ldo=C,
                              scale=D ** -0.5, D=D) for q, k, vt in zip(qs, ks, vts)]
        for d in descs:
            lib.call(lib.OP_ATTN_FWD, d, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        e0.record()
        for rep in range(8):
            for d in descs:
                lib.call(lib.OP_ATTN_FWD, d, s)
                n += 1
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        wgs = BH * T // 128
        fl = 4.0 * BH * T * T * D
        tiles = T // 64
        print(f"T {T:5d} BH {BH:4d}  wgs/CU {wgs / 256:5.2f}  {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s   "
              f"{us * 2400 / tiles:7.0f} cyc per key tile step", flush=True)
