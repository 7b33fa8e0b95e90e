"""How much of a small product's time is the latency of its HBM-cold weights?  Same launch timed (a) cold: weights rotated through
600 MB of copies, (b) warm: one copy, (c) cold but touched just before by a streaming read (lands in the Infinity Cache and in
whatever L2s the touching workgroups ran on).  Events bracket the slh_gemm launch only."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sliders_amd import lib
from sliders_amd.weights import pack_gemm_w

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
s = stream.cuda_stream
SHAPES = [(2048, 1280, 1280, 0x4412), (2048, 1280, 5120, 0x4412), (2048, 3840, 1280, 0x4012), (2048, 10240, 1280, 0x4012),
          (8192, 640, 640, 0x12), (8192, 640, 2560, 0x4012), (8192, 5120, 640, 0x4012)]
for M, N, K, tile in SHAPES:
    x = torch.randn(M, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    res = torch.randn(M, N, device=dev).bfloat16()
    nc = max(2, min(64, int(600e6 // (N * K * 2)) + 1))
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    wps = [pack_gemm_w(w) for _ in range(nc)]
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    descs = [lib.GemmDesc(a0=x.data_ptr(), w=wp.data_ptr(), bias=bias.data_ptr(), residual=res.data_ptr(), c=c.data_ptr(), lda0=K, ca0=K,
                          mode=0, stride=1, ldw=0, M=M, N=N, K=K, ld_res=N, ldc=N, rows_per_sample=M, tile=tile, w_layout=1) for wp in wps]

    def run(mode, reps=24):
        tot = 0.0
        for i in range(reps + 4):
            j = 0 if mode == "warm" else i % nc
            if mode == "touch":
                wps[j].view(torch.int32).sum()
                x.add_(0)                      # the activations as the previous kernel of a pass leaves them: just written
            elif mode == "cold":
                x.add_(0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            lib.call(lib.OP_GEMM, descs[j], s)
            e1.record(stream)
            e1.synchronize()
            if i >= 4:
                tot += e0.elapsed_time(e1) * 1e3
        return tot / reps

    print(f"{M}x{N}x{K} tile {tile:x}: cold {run('cold'):6.1f} us  touched {run('touch'):6.1f} us  warm {run('warm'):6.1f} us", flush=True)
