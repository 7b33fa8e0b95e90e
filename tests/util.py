"""Shared helpers for the GPU parity tests (the HIP path is always called through the C ABI)."""
import torch

from sliders_amd import lib


def stream():
    return torch.cuda.current_stream().cuda_stream


def p(t):
    return t.data_ptr() if t is not None else 0


def bf(t):
    return t.to(torch.bfloat16)


def rel_err(a, b):
    a = a.float()
    b = b.float()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def max_err(a, b):
    return (a.float() - b.float()).abs().max().item()


def report(name, got, ref, tol_rel):
    r = rel_err(got, ref)
    m = max_err(got, ref)
    print(f"[parity] {name}: rel_l2={r:.3e} max_abs={m:.3e} ref_rms={ref.float().pow(2).mean().sqrt().item():.3e}")
    assert torch.isfinite(got.float()).all(), f"{name}: non-finite output"
    assert r < tol_rel, f"{name}: rel_l2 {r:.3e} >= {tol_rel:.1e} (max_abs {m:.3e})"
