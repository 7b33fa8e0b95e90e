"""GEGLU.proj of the 1280-channel level (2048 x 10240 x 1280, geglu = 3, bias) by tile, cold weights (development aid; GPU box)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd import lib
from sliders_amd.weights import _geglu_perm16, pack_gemm_w

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
s = stream.cuda_stream
tiles = [int(t, 16) for t in (sys.argv[1] if len(sys.argv) > 1 else "8015,4012").split(",")]
for M, N, K in ((2048, 10240, 1280), (8192, 5120, 640)):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    wps = [pack_gemm_w(_geglu_perm16(w)).clone() for _ in range(12)]
    bp = _geglu_perm16(b).contiguous()
    c = torch.zeros(M, N // 2, device=dev, dtype=torch.bfloat16)
    proj = (x.float() @ w.float().t() + b.float()).bfloat16().float()
    ref = proj[:, :N // 2] * torch.nn.functional.gelu(proj[:, N // 2:]).bfloat16().float()
    out = []
    for tile in tiles:
        ds = [lib.GemmDesc(a0=x.data_ptr(), w=wp.data_ptr(), bias=bp.data_ptr(), c=c.data_ptr(), lda0=K, ca0=K, mode=0, stride=1, ldw=0, M=M, N=N,
                           K=K, ldc=N // 2, geglu=3, rows_per_sample=M, tile=tile, w_layout=1) for wp in wps]
        c.zero_()
        lib.call(lib.OP_GEMM, ds[0], s)
        torch.cuda.synchronize()
        err = ((c.float() - ref).norm() / ref.norm()).item()
        for i in range(3):
            lib.call(lib.OP_GEMM, ds[i % len(ds)], s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(36):
            lib.call(lib.OP_GEMM, ds[i % len(ds)], s)
        e1.record(stream)
        e1.synchronize()
        us = e0.elapsed_time(e1) / 36 * 1e3
        out.append(f"{tile:x}: {us:6.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF" + ("" if err < 6e-3 else f" ERR{err:.1e}"))
    print(f"{M}x{N}x{K} geglu16: " + " | ".join(out), flush=True)
