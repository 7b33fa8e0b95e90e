"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name (sum of counters, dispatch count)."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
DROP_FIRST_THIRD = "--drop-first-third" in sys.argv      # three passes were profiled (first call + warm-up + replay): the first call's
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0][:70].replace(",", ";")
        agg[(k, int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(int(r["Dispatch_Id"]))
# ... dispatches (cold weights, first-touch misses) are dropped per kernel name when asked to
per_disp, agg = agg, defaultdict(lambda: defaultdict(float))
for k in list(cnt):
    ids = sorted(cnt[k])
    keep = set(ids[len(ids) // 3:]) if (DROP_FIRST_THIRD and len(ids) >= 3) else set(ids)
    cnt[k] = keep
    for i in keep:
        for c, v in per_disp[(k, i)].items():
            agg[k][c] += v
names = sorted({c for v in agg.values() for c in v})
print("kernel,dispatches," + ",".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("GRBM_GUI_ACTIVE", 0))):
    print(f"{k},{len(cnt[k])}," + ",".join(f"{v.get(n, 0):.4g}" for n in names))
