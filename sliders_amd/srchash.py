"""Hash of everything that decides which kernels a pass launches and how they are built: the HIP sources and the tuned tile
tables.  Counter passes (profiles/r*_pmc_traffic.json) record it; bench.py attaches their HBM-traffic figures to a bench line only
when the tree it runs from hashes the same."""
import glob
import hashlib
import os


def kernel_source_hash() -> str:
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")) +
                   glob.glob(os.path.join(here, "csrc", "*.cpp")) + glob.glob(os.path.join(here, "tuning", "*.json")) +
                   [os.path.join(os.path.dirname(here), "include", "sliders_hip.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
