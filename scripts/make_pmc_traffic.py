"""profiles/rNN_pmc_traffic.json from the two PMC summaries of scripts/measure_round2.sh (scripts/pmc_summary.py CSVs):
per kernel, fabric-side bytes per launch.  FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B,
MI355X_MICROARCH.md section HBM), WRITE_SIZE is taken as reported (uncalibrated); both counters are in KB.
usage: make_pmc_traffic.py fetch_summary.csv write_summary.csv out.json"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_amd.srchash import file_hashes, kernel_source_hash


def rows(path):
    with open(path) as f:
        lines = [l for l in f if "," in l]
    return list(csv.DictReader(lines))


fetch, write, out = sys.argv[1:4]
tree_head = sys.argv[4] if len(sys.argv) > 4 else None      # git HEAD of the tree the passes were taken on
kern = {}
for r in rows(fetch):
    n = max(int(r["dispatches"]), 1)
    kern[r["kernel"]] = {"launches": n, "fetch_bytes_per_launch": int(2 * 1024 * float(r["FETCH_SIZE"]) / n)}
for r in rows(write):
    n = max(int(r["dispatches"]), 1)
    k = kern.setdefault(r["kernel"], {"launches": n})
    k["write_bytes_per_launch"] = int(1024 * float(r["WRITE_SIZE"]) / n)
    hit, miss = float(r["TCC_HIT_sum"]), float(r["TCC_MISS_sum"])
    k["l2_hit_rate"] = round(hit / (hit + miss), 3) if hit + miss > 0 else None
res = {"tree_head": tree_head, "kernel_source_hash": kernel_source_hash(), "file_hashes": file_hashes(), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes, --kernel-trace only) over "
                 "LoRA-on B=2 UNet passes of the configuration in the file name (scripts/bench_forward.py --lora --warm 1 --iters 1: first call, "
                 "warm-up, one replay - the first call's dispatches dropped per kernel, scripts/pmc_summary.py --drop-first-third); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 64 B per 128-B request); "
                 "WRITE_SIZE uncalibrated; KB -> bytes",
       "kernels": {k: v for k, v in kern.items() if "fetch_bytes_per_launch" in v and "write_bytes_per_launch" in v}}
with open(out, "w") as f:
    json.dump(res, f, indent=1)
print(f"{len(res['kernels'])} kernels -> {out}")
