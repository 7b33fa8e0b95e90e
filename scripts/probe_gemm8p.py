"""Timing + correctness of slh_gemm tiles over the dominant shapes of the SDXL 1024^2 pass (development aid).
Weights are rotated through enough copies to be HBM-cold on every launch (as in the pass: 5 GB of weights stream through
the 256 MB Infinity Cache); the activations stay warm (the previous kernel of a pass has just written them)."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from sliders_amd import lib
from sliders_amd.weights import pack_gemm_w

DENSE = "2048x10240x1280,2048x1280x5120,2048x3840x1280,2048x1280x1280,8192x5120x640,8192x640x2560,8192x1920x640,8192x640x640,4096x4096x4096"
CONV = "2x128x128x320x320,2x64x64x640x640,2x32x32x1280x1280,2x64x64x1280x640,2x128x128x640x320"

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default=DENSE)
ap.add_argument("--convs", default=CONV, help="BxHxWxCinxCout 3x3 stride 1")
ap.add_argument("--tiles", default="4012,4412,8042")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--check", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
s = stream.cuda_stream
tiles = [int(t, 16) for t in a.tiles.split(",")]


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


def copies_for(nbytes):
    return max(2, min(48, int(600e6 // nbytes) + 1))


for shp in [v for v in a.shapes.split(",") if v]:
    M, N, K = (int(v) for v in shp.split("x"))
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    res = torch.randn(M, N, device=dev).bfloat16()
    nc = copies_for(N * K * 2)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    wps = [pack_gemm_w(w) for _ in range(nc)]
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ref = (x.float() @ w.float().t() + bias.float() + res.float()) if a.check else None
    row = []
    for tile in tiles:
        descs = [lib.GemmDesc(a0=x.data_ptr(), w=wp.data_ptr(), bias=bias.data_ptr(), residual=res.data_ptr(), c=c.data_ptr(),
                              lda0=K, ca0=K, mode=0, stride=1, ldw=0, M=M, N=N, K=K, ld_res=N, ldc=N, rows_per_sample=M, tile=tile,
                              w_layout=1) for wp in wps]
        if tile >> 16:     # split-K (bits 16-19) / stream-K (bits 20-21): slab workspace + zeroed tickets
            nsl = 8
            ws = torch.empty(nsl * ((M + 255) // 256 * 256) * ((N + 127) // 128 * 128), device=dev, dtype=torch.float32)
            tk = torch.zeros(4096, device=dev, dtype=torch.int64)
            for d in descs:
                d.splitk_c32, d.splitk_ticket, d.splitk_slabs = ws.data_ptr(), tk.data_ptr(), nsl
        err = ""
        if a.check:
            c.zero_()
            lib.call(lib.OP_GEMM, descs[0], s)
            torch.cuda.synchronize()
            e = ((c.float() - ref).norm() / ref.norm()).item()
            err = f" (rel {e:.1e})" if e < 5e-3 else f" (REL {e:.2e} !!!)"
        us = time_it(descs, a.reps)
        row.append(f"{tile:x}: {us:6.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF{err}")
    print(f"{shp:20s} " + " | ".join(row), flush=True)

for shp in [v for v in a.convs.split(",") if v]:
    B, H, W, Ci, Co = (int(v) for v in shp.split("x"))
    torch.manual_seed(H + Ci + Co)
    img = torch.randn(B, Ci, H, W, device=dev).bfloat16()
    w4 = (torch.randn(Co, Ci, 3, 3, device=dev) / math.sqrt(9 * Ci)).bfloat16()
    bias = torch.randn(Co, device=dev).bfloat16()
    xp = img.permute(0, 2, 3, 1).reshape(B * H * W, Ci).contiguous()
    wflat = w4.permute(0, 2, 3, 1).reshape(Co, -1).contiguous()
    M, N, K = B * H * W, Co, 9 * Ci
    nc = copies_for(N * K * 2)
    wps = [pack_gemm_w(wflat) for _ in range(nc)]
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ref = None
    if a.check:
        ref = (F.conv2d(img.float(), w4.float(), bias.float(), padding=1)).permute(0, 2, 3, 1).reshape(M, N)
    row = []
    for tile in tiles:
        descs = [lib.GemmDesc(a0=xp.data_ptr(), w=wp.data_ptr(), bias=bias.data_ptr(), c=c.data_ptr(), lda0=Ci, ca0=Ci, mode=1, batch=B,
                              hs=H, ws=W, stride=1, ho=H, wo=W, ldw=0, M=M, N=N, K=K, ldc=N, rows_per_sample=H * W, tile=tile,
                              w_layout=1) for wp in wps]
        err = ""
        if a.check:
            c.zero_()
            lib.call(lib.OP_GEMM, descs[0], s)
            torch.cuda.synchronize()
            e = ((c.float() - ref).norm() / ref.norm()).item()
            err = f" (rel {e:.1e})" if e < 5e-3 else f" (REL {e:.2e} !!!)"
        us = time_it(descs, a.reps)
        row.append(f"{tile:x}: {us:6.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF{err}")
    print(f"conv {shp:20s} M{M} N{N} K{K} " + " | ".join(row), flush=True)
