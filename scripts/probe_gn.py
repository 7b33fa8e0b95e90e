"""GroupNorm stats / apply kernel times for the shapes of an SDXL 1024x1024 pass (development aid; GPU box).
Each shape cycles through enough distinct buffers that the inputs come from HBM, as in the real pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd import lib

dev = torch.device("cuda:0")
lib.load()
s = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
B, G = 2, 32
shapes = [(16384, 320), (16384, 640), (16384, 960), (4096, 320), (4096, 640), (4096, 1280), (4096, 1920), (1024, 640),
          (1024, 1280), (1024, 2560), (1024, 1920), (1024, 320), (1024, 960), (256, 640), (256, 1280), (256, 1920), (256, 2560),
          (64, 1280), (64, 2560)]
tot = 0.0
for hw, C in shapes:
    nbuf = max(2, int(600e6 / (B * hw * C * 2)))
    xs = [torch.randn(B * hw, C, device=dev).to(torch.bfloat16) for _ in range(nbuf)]
    y = torch.empty(B * hw, C, device=dev, dtype=torch.bfloat16)
    g, bt = torch.ones(C, device=dev, dtype=torch.bfloat16), torch.zeros(C, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(B, G, 2, device=dev)
    prow, ntick = lib.gn_workspace(C, hw, G)
    part = torch.zeros(B, prow, G, 2, device=dev)
    ticket = torch.zeros(B, ntick, dtype=torch.int32, device=dev)
    descs = [lib.GnDesc(x0=p(x), gamma=p(g), beta=p(bt), stats=p(stats), y=p(y), ldx0=C, c0=C, batch=B, hw=hw, groups=G, ldy=C,
                        eps=1e-5, act=1, partial=p(part), ticket=p(ticket)) for x in xs]
    res = {}
    ops = [(lib.OP_GN_STATS, "stats"), (lib.OP_GN_APPLY, "apply")]
    if lib.gn_fused_ok(C, hw, G):
        ops.append((lib.OP_GN_FUSED, "fused"))
    for op, nm in ops:
        for d in descs[:2]:
            lib.call(op, d, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        e0.record()
        for rep in range(3):
            for d in descs:
                lib.call(op, d, s)
                n += 1
        e1.record()
        torch.cuda.synchronize()
        res[nm] = e0.elapsed_time(e1) * 1e3 / n
    mb = B * hw * C * 2 / 1e6
    tot += res["stats"] + res["apply"]
    fz = f"  fused {res['fused']:6.1f} us" if "fused" in res else ""
    print(f"hw {hw:6d} C {C:5d} ({mb:6.1f} MB, {prow:4d} partial rows): stats {res['stats']:6.1f} us "
          f"({mb / res['stats'] * 1e3 / 1e3:6.0f} GB/s)  apply {res['apply']:6.1f} us ({2 * mb / res['apply']:6.0f} GB/s){fz}", flush=True)
print(f"sum over shapes {tot:.1f} us")
