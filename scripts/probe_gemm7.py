"""Isolated timing of slh_gemm tiles (development aid): cold weights (rotated through enough copies to leave the 256 MB Infinity Cache)
and warm weights (one copy), plain product, per shape and tile.  K sweeps give the per-K-tile rate and the fixed cost of a launch."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd import lib
from sliders_amd.weights import pack_gemm_w

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="2048x3840x256,2048x3840x640,2048x3840x1280,2048x3840x2560")
ap.add_argument("--tiles", default="8014,7648")
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--epi", default="none", choices=["none", "bias_res", "geglu", "geglu_ln"])
a = ap.parse_args()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
s = stream.cuda_stream


def time_it(descs, reps):
    for d in descs[:3]:
        lib.call(lib.OP_GEMM, d, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(reps):
        lib.call(lib.OP_GEMM, descs[i % len(descs)], s)
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for shp in a.shapes.split(","):
    M, N, K = (int(v) for v in shp.split("x"))
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    res = torch.randn(M, N, device=dev).bfloat16()
    nc = max(2, min(48, int(600e6 // (N * K * 2)) + 1))
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    wps = [pack_gemm_w(w) for _ in range(nc)]
    geglu = a.epi.startswith("geglu")
    c = torch.zeros(M, N // 2 if geglu else N, device=dev, dtype=torch.bfloat16)
    ref = x.float() @ w.float().t() + ((bias.float() + res.float()) if a.epi == "bias_res" else 0)
    lnkw = {}
    if geglu:      # timing only (16 | 16 block order taken as given); the LayerNorm-fold variant gets unit statistics
        ref = None
        if a.epi == "geglu_ln":
            ch = torch.zeros(K // 64, M, 2, device=dev); ch[..., 1] = 64.0
            sv, bp = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
            lnkw = dict(ln_in=ch.data_ptr(), ln_in_chunks=K // 64, ln_s=sv.data_ptr(), ln_b=bp.data_ptr(), ln_eps=1e-5)
    row = []
    for t in a.tiles.split(","):
        tile = int(t, 16)
        mk = lambda wp: lib.GemmDesc(a0=x.data_ptr(), w=wp.data_ptr(), bias=bias.data_ptr() if a.epi in ("bias_res", "geglu") else 0,
                                     residual=res.data_ptr() if a.epi == "bias_res" else 0, c=c.data_ptr(), lda0=K, ca0=K, mode=0, stride=1,
                                     ldw=0, M=M, N=N, K=K, ld_res=N, ldc=c.shape[1], rows_per_sample=M, tile=tile, w_layout=1,
                                     geglu=3 if geglu else 0, **lnkw)
        descs = [mk(wp) for wp in wps]
        c.zero_()
        try:
            lib.call(lib.OP_GEMM, descs[0], s)
        except lib.SlidersHipError:
            row.append(f"{tile:x}: refused")
            continue
        torch.cuda.synchronize()
        e = ((c.float() - ref).norm() / ref.norm()).item() if ref is not None else 0.0
        cold = time_it(descs, a.reps)
        warm = time_it(descs[:1], a.reps)
        row.append(f"{tile:x}: cold {cold:6.1f} warm {warm:6.1f} us ({2.0 * M * N * K / warm / 1e6:5.0f} TF){'' if e < 5e-3 else ' WRONG %.2e' % e}")
    print(f"{shp:18s} " + " | ".join(row), flush=True)
