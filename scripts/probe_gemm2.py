"""Cold/warm timing + correctness of slh_gemm tile variants on the SDXL shapes (development aid).
cold = consecutive launches rotate through enough weight copies to defeat the 256 MB Infinity Cache (as in a real
pass, where every weight is read once); warm = the same weights every launch."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd import lib
from sliders_amd.weights import pack_gemm_w

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="2048x1280x1280,2048x1280x5120,2048x3840x1280,2048x10240x1280,8192x640x640,8192x5120x640,8192x640x2560,32768x320x640")
ap.add_argument("--tiles", default="11,22,4012,422,4412,4322,322,412,421")
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--lora", type=int, default=0)
ap.add_argument("--res", type=int, default=0, help="1: with bias and residual (timing of the epilogue's operand loads)")
a = ap.parse_args()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
s = stream.cuda_stream
for shp in a.shapes.split(","):
    M, N, K = (int(v) for v in shp.split("x"))
    x = torch.randn(M, K, device=dev).bfloat16()
    wbytes = 2 * N * K
    ncopy = max(2, min(160, int(math.ceil(420e6 / wbytes))))
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(2)]
    wps = [pack_gemm_w(ws[i % 2]).clone() for i in range(ncopy)]
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ref = x.float() @ ws[0].float().t()
    A = (torch.randn(4, K, device=dev) / math.sqrt(K)).bfloat16()
    up = torch.randn(N, 4, device=dev).bfloat16()
    scale = torch.tensor([0.25], device=dev)
    if a.lora:
        ref = ref + 0.25 * (x.float() @ A.float().t()) @ up.float().t()
    bias = torch.randn(N, device=dev).bfloat16()
    resid = torch.randn(M, N, device=dev).bfloat16()
    if a.res:
        ref = ref + bias.float() + resid.float()
    fl = 2.0 * M * N * K
    out = []
    for tile in (int(t, 16) for t in a.tiles.split(",")):
        def desc(wp):
            d = lib.GemmDesc(a0=x.data_ptr(), w=wp.data_ptr(), c=c.data_ptr(), lda0=K, ca0=K, mode=0, stride=1, ldw=0,
                             M=M, N=N, K=K, ldc=N, rows_per_sample=M, tile=tile, w_layout=1)
            if a.res:
                d.bias, d.residual, d.ld_res = bias.data_ptr(), resid.data_ptr(), N
            if a.lora:
                d.lora_down, d.lora_up, d.lora_scale = A.data_ptr(), up.data_ptr(), scale.data_ptr()
                d.ld_t, d.lora_groups, d.lora_rank = 4, 1, 4
            return d
        ds = [desc(wp) for wp in wps]
        c.zero_()
        lib.call(lib.OP_GEMM, ds[0], s)
        torch.cuda.synchronize()
        err = ((c.float() - ref).norm() / ref.norm()).item()
        res = {}
        for mode in ("cold", "warm"):
            for i in range(3):
                lib.call(lib.OP_GEMM, ds[i % ncopy if mode == "cold" else 0], s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for i in range(a.reps):
                lib.call(lib.OP_GEMM, ds[(i + 3) % ncopy if mode == "cold" else 0], s)
            e1.record(stream)
            e1.synchronize()
            res[mode] = e0.elapsed_time(e1) / a.reps * 1e3
        out.append(f"{tile:x}: {res['cold']:6.1f}/{res['warm']:6.1f}us {fl / res['cold'] / 1e6:5.0f}TF" + ("" if err < 6e-3 else f" ERR{err:.1e}"))
    print(f"{shp:18s} lora{a.lora} " + " | ".join(out), flush=True)
