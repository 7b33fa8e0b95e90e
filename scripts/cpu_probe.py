import os, time
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except OSError as e: print(p, "n/a")
print("loadavg", open("/proc/loadavg").read().strip())
import torch
print("torch threads default", torch.get_num_threads())
x = torch.randn(2, 320, 32, 32); w = torch.randn(320, 320, 3, 3)
for nt in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    torch.nn.functional.conv2d(x, w, padding=1)
    t0 = time.time()
    for _ in range(20): torch.nn.functional.conv2d(x, w, padding=1)
    t1 = time.time()
    a = torch.randn(2048, 1280); b = torch.randn(1280, 1280)
    a @ b
    t2 = time.time()
    for _ in range(5): a @ b
    t3 = time.time()
    print(f"threads {nt}: conv {1e3*(t1-t0)/20:.2f} ms  gemm {1e3*(t3-t2)/5:.2f} ms", flush=True)
