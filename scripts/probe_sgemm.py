"""Timing probe of slh_sgemm (development aid): exact vs split, with the split kernel's ablation bits."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sliders_amd import lib
dev = torch.device("cuda:0")
s = torch.cuda.current_stream()
for (B, H, Ci, Co) in ((1, 512, 128, 128), (1, 256, 256, 256), (1, 128, 512, 512), (1, 64, 512, 512)):
    x = torch.randn(B * H * H, Ci, device=dev)
    w = torch.randn(Co, 9 * Ci, device=dev) / math.sqrt(9 * Ci)
    c = torch.zeros(B * H * H, Co, device=dev)
    out = []
    for split in (0, 1, 3, 5, 9, 15):
        d = lib.SgemmDesc(x=x.data_ptr(), w=w.data_ptr(), c=c.data_ptr(), ldx=Ci, ldw=9 * Ci, ldc=Co, M=B * H * H, N=Co, K=9 * Ci,
                          mode=1, cin=Ci, batch=B, hs=H, ws=H, ho=H, wo=H, stride=1, pad=1, alpha=1.0, split_bf16=split)
        for _ in range(2):
            lib.call(lib.OP_SGEMM, d, s.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            lib.call(lib.OP_SGEMM, d, s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        out.append(f"split{split}: {us:7.1f} us ({2.0 * B * H * H * Co * 9 * Ci / us * 1e-6:5.0f} TF/s)")
    print(f"conv {H}x{H} {Ci}->{Co}: " + " | ".join(out), flush=True)
print("bits: 1 split kernel, +2 no LDS stores/cvt, +4 no MFMA, +8 no global loads after the prologue")
