"""Hashes of everything that decides which kernels a pass launches and how they are built: the HIP sources and the tuned tile
tables.  Counter passes (profiles/r*_pmc_traffic.json) record them; bench.py attaches their HBM-traffic figures to a bench line only
when the tree it runs from hashes the same - as a whole, or (file_hashes) in every file the reported kernel is built from."""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def _files():
    return sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                  glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) + glob.glob(os.path.join(_HERE, "tuning", "*.json")) +
                  [os.path.join(os.path.dirname(_HERE), "include", "sliders_hip.h")])


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for f in _files():
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def file_hashes() -> dict:
    """{basename: sha256[:16]} of the same file set, one entry per file."""
    out = {}
    for f in _files():
        with open(f, "rb") as fh:
            out[os.path.basename(f)] = hashlib.sha256(fh.read()).hexdigest()[:16]
    return out


def kernel_files(kernel_name: str, all_files) -> list:
    """The files a kernel's code and its launches depend on: its translation unit, the headers it includes, the public header
    (descriptor layout) and every tuned tile table (they decide which launches run the kernel)."""
    unit = {"gemm_kernel": ["gemm.hip", "gemm_common.h"], "gemm8p": ["gemm8p.hip", "gemm_common.h"],
            "gemm5": ["gemm5.hip", "gemm.hip", "gemm_common.h"], "gemm7": ["gemm7.hip", "gemm.hip", "gemm_common.h"],
            "attn_fwd": ["attention.hip"], "attn_bwd": ["attention_bwd.hip"]}
    picked = None
    for prefix, files in unit.items():
        if kernel_name.startswith(prefix):
            picked = list(files)
    if picked is None:
        return sorted(all_files)                      # unknown kernel: the whole set must match
    return sorted(set(picked + ["common.h", "sliders_hip.h"] + [f for f in all_files if f.endswith(".json")]))
