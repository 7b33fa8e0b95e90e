"""Development aid (never used by the product path or the tests): what does the vendor library (hipBLASLt through
torch.matmul) reach on the pass's dominant dense shapes?  A ceiling estimate for slh_gemm's tile schedules; run under
`rocprofv3 --kernel-trace --stats` to see which macro-tiles it picks."""
import torch

dev = torch.device("cuda:0")
shapes = [(2048, 10240, 1280), (2048, 1280, 5120), (2048, 1280, 1280), (2048, 3840, 1280), (2048, 1280, 11520),
          (8192, 640, 5760), (8192, 5120, 640), (8192, 640, 640), (32768, 320, 2880), (8192, 640, 2560)]
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    # 24 distinct weights cycled so that the weight is as cold as in the replayed pass
    ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) for _ in range(24)]
    for w in ws[:3]:
        torch.matmul(x, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(2):
        for w in ws:
            torch.matmul(x, w.t())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 48
    e0.record()
    for r in range(48):
        torch.matmul(x, ws[0].t())
    e1.record()
    torch.cuda.synchronize()
    us_w = e0.elapsed_time(e1) * 1e3 / 48
    print(f"{M}x{N}x{K}: cold-weights {us:7.1f} us ({2.0 * M * N * K / us * 1e-6:6.0f} TF/s)   warm {us_w:7.1f} us "
          f"({2.0 * M * N * K / us_w * 1e-6:6.0f} TF/s)", flush=True)
