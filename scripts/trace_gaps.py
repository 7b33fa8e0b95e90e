"""Idle time between kernels of replayed UNet passes, from a rocprofv3 --kernel-trace CSV (development aid).
usage: trace_gaps.py <dir with *kernel_trace.csv> [min_gap_us_to_split_passes]"""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
ev = []
for f in files:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
ev.sort()
split = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 300e3
runs, cur = [], [ev[0]]
for a, b in zip(ev, ev[1:]):
    if b[0] - a[1] > split:
        runs.append(cur); cur = []
    cur.append(b)
runs.append(cur)
print(f"{len(ev)} kernels, {len(runs)} bursts (split at gaps > {split / 1e3:.0f} us)")
for r in sorted(runs, key=len, reverse=True)[:4]:
    span = r[-1][1] - r[0][0]
    busy = sum(e - s for s, e, _ in r)
    gaps = [b[0] - a[1] for a, b in zip(r, r[1:])]
    pos = [g for g in gaps if g > 0]
    print(f"  burst of {len(r)} kernels: span {span / 1e6:.3f} ms, sum of kernel durations {busy / 1e6:.3f} ms, "
          f"idle {100.0 * (span - busy) / span:.1f} %, mean positive gap {sum(pos) / max(len(pos), 1) / 1e3:.2f} us, "
          f"overlapping successors {sum(1 for g in gaps if g <= 0)}")
