"""Bit-identity of the pipelined attention forward forms against the shipped dispatch (development aid; GPU box): run once per
SLH_ATTN_PL value with the same seed, each run writes the outputs' bytes (O and lse) to a file; `--compare a b` says whether
two runs agree."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = 0
    for k in a:
        same = torch.equal(a[k], b[k])
        bad += 0 if same else 1
        if not same:
            d = (a[k].float() - b[k].float()).abs()
            print(f"{k}: DIFFERENT ({int((a[k] != b[k]).sum())} of {a[k].numel()} elements, max |d| {d.max().item():.3e})")
    print(f"{sys.argv[2]} vs {sys.argv[3]}: {'identical' if bad == 0 else f'{bad} tensors differ'} ({len(a)} tensors)")
    sys.exit(1 if bad else 0)

from sliders_amd import lib

dev = torch.device("cuda:0")
lib.load()
s = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
out = {}
g = torch.Generator(device="cpu").manual_seed(7)
for name, B, H, Tq, Tk in (("self64", 2, 10, 4096, 4096), ("self32", 2, 20, 1024, 1024), ("self16", 2, 20, 256, 256),
                           ("b1", 1, 5, 512, 128), ("cross_like", 1, 3, 128, 256)):
    C = H * 64
    q = (torch.randn(B * Tq, C, generator=g) * 1.5).to(dev, torch.bfloat16)
    k = (torch.randn(B * Tk, C, generator=g) * 1.5).to(dev, torch.bfloat16)
    vt = torch.randn(B, H, 64, Tk, generator=g).to(dev, torch.bfloat16)
    o = torch.zeros(B * Tq, C, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B * H * Tq + 64, device=dev, dtype=torch.float32)
    d = lib.AttnDesc(q=p(q), k=p(k), vt=p(vt), o=p(o), lse=p(lse), B=B, H=H, Tq=Tq, Tk=Tk, ldq=C, ldk=C, ldvt=Tk, ldo=C, scale=0.125, D=64)
    lib.call(lib.OP_ATTN_FWD, d, s)
    torch.cuda.synchronize()
    out[name + ".o"] = o.cpu()
    out[name + ".lse"] = lse.cpu()
    # sanity against fp32 math
    qf = q.float().view(B, Tq, H, 64).transpose(1, 2)
    kf = k.float().view(B, Tk, H, 64).transpose(1, 2)
    ref = torch.softmax(qf @ kf.transpose(-1, -2) * 0.125, -1) @ vt.float().transpose(-1, -2)
    err = (ref.transpose(1, 2).reshape(B * Tq, C) - o.float()).abs().max().item()
    print(f"{name}: max |o - fp32 reference| = {err:.3e}")
torch.save(out, sys.argv[1])
