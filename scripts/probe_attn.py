"""Attention forward kernel times on the shapes of the SDXL 1024 / 512 and SD-1.x 512 passes (development aid; GPU box).
SLIDERS_HIP_LIB selects the library build (A/B of kernel variants in one call)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd import lib

dev = torch.device("cuda:0")
lib.load()
s = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
shapes = [("sdxl1024 self 64^2", 2, 10, 4096, 4096, 64), ("sdxl1024 self 32^2", 2, 20, 1024, 1024, 64),
          ("sdxl1024 cross 64^2", 2, 10, 4096, 77, 64), ("sdxl1024 cross 32^2", 2, 20, 1024, 77, 64),
          ("sdxl512 self 32^2", 2, 10, 1024, 1024, 64), ("sdxl512 self 16^2", 2, 20, 256, 256, 64),
          ("sd1 self 64^2 D40", 2, 8, 4096, 4096, 40), ("sd1 self 32^2 D80", 2, 8, 1024, 1024, 80),
          ("sd1 self 16^2 D160", 2, 8, 256, 256, 160), ("sd1 cross 64^2 D40", 2, 8, 4096, 77, 40)]
tot = 0.0
for name, B, H, Tq, Tk, D in shapes:
    C = H * D
    Dp, ldt = (D + 63) // 64 * 64, (Tk + 63) // 64 * 64
    nb = 4
    qs = [torch.randn(B * Tq, C, device=dev).to(torch.bfloat16) for _ in range(nb)]
    ks = [torch.randn(B * Tk, C, device=dev).to(torch.bfloat16) for _ in range(nb)]
    vts = [torch.randn(B, H, Dp, ldt, device=dev).to(torch.bfloat16) for _ in range(nb)]
    o = torch.empty(B * Tq, C, device=dev, dtype=torch.bfloat16)
    descs = [lib.AttnDesc(q=p(q), k=p(k), vt=p(vt), o=p(o), B=B, H=H, Tq=Tq, Tk=Tk, ldq=C, ldk=C, ldvt=ldt, ldo=C,
                          scale=D ** -0.5, D=D) for q, k, vt in zip(qs, ks, vts)]
    for d in descs:
        lib.call(lib.OP_ATTN_FWD, d, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    e0.record()
    for rep in range(10):
        for d in descs:
            lib.call(lib.OP_ATTN_FWD, d, s)
            n += 1
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 4.0 * B * H * Tq * Tk * D
    tot += us
    print(f"{name:24s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
print(f"sum {tot:.1f} us  ({os.environ.get('SLIDERS_HIP_LIB', 'default lib')})")
