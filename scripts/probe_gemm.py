"""Ablation timing of one slh_gemm shape: which part of the K loop the time goes to (development aid).
(Round 6: the -DSLH_GEMM_PROBE branches were removed from csrc/gemm.hip - the shipped library ignores reserved_; build the probe
variant from `git show 46d9a74:sliders_amd/csrc/gemm.hip` to re-run the ablations recorded in profiles/r0[1-5]_*.txt.)"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd import lib
from sliders_amd.weights import pack_gemm_w

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="2048x10240x1280,2048x1280x1280,2048x1280x5120,8192x640x640")
ap.add_argument("--tiles", default="22,11,4012")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--probes", default="0,4,1,2,3,5,6,7,15,16",
                help="reserved_ values; 32 / 64 (no W / no X refills) exist only in a -DSLH_GEMM_PROBE_W build (SLIDERS_HIP_LIB)")
a = ap.parse_args()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
s = stream.cuda_stream
for shp in a.shapes.split(","):
    M, N, K = (int(v) for v in shp.split("x"))
    x = (torch.randn(M, K, device=dev)).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    wp = pack_gemm_w(w)
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    for tile in (int(t, 16) for t in a.tiles.split(",")):
        row = []
        for probe in (int(v) for v in a.probes.split(",")):
            d = lib.GemmDesc(a0=x.data_ptr(), w=wp.data_ptr(), c=c.data_ptr(), lda0=K, ca0=K, mode=0, stride=1, ldw=0,
                             M=M, N=N, K=K, ldc=N, rows_per_sample=M, tile=tile, w_layout=1, reserved_=probe)
            for _ in range(3):
                lib.call(lib.OP_GEMM, d, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(a.reps):
                lib.call(lib.OP_GEMM, d, s)
            e1.record(stream)
            e1.synchronize()
            row.append((probe, e0.elapsed_time(e1) / a.reps * 1e3))
        fl = 2.0 * M * N * K
        print(f"{shp:22s} tile {tile:5x}: " + "  ".join(f"p{pr}:{us:6.1f}" for pr, us in row) +
              f"   full {fl / row[0][1] / 1e6:5.0f} TF/s   [p1 no refill, p2 no mfma/ds_read, p4 no epilogue, p8 no first fill, p16 empty, p32 no W refill, p64 no X refill]", flush=True)
