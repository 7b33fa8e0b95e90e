"""Where the GPU is idle inside training iterations, from a rocprofv3 --kernel-trace CSV of a bench.py run (development aid).
Merges kernel intervals over all streams, takes the last `window_ms` of the trace (the timed iterations), and lists the largest
gaps with the kernels on either side.   usage: trace_idle.py <dir with *kernel_trace.csv> [window_ms] [top]"""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1500e6
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
ev = []
for f in files:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
ev.sort()
t_end = max(e for _, e, _ in ev)
ev = [x for x in ev if x[0] >= t_end - window]
busy, gaps = 0, []
cs, ce, last = ev[0][0], ev[0][1], ev[0][2]
for s, e, n in ev[1:]:
    if s > ce:
        gaps.append((s - ce, ce - ev[0][0], last, n))
        busy += ce - cs
        cs, ce, last = s, e, n
    elif e > ce:
        ce, last = e, n
busy += ce - cs
span = ce - ev[0][0]
print(f"{len(ev)} kernels in the last {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms, idle {(span - busy) / 1e6:.2f} ms = {100.0 * (span - busy) / span:.2f} %")
hist = {}
for g, *_ in gaps:
    b = "<2us" if g < 2e3 else "<10us" if g < 1e4 else "<100us" if g < 1e5 else "<1ms" if g < 1e6 else ">=1ms"
    hist.setdefault(b, [0, 0]); hist[b][0] += 1; hist[b][1] += g
print("  gaps by size: " + "  ".join(f"{k}: {v[0]} = {v[1] / 1e6:.2f} ms" for k, v in hist.items()))
for g, at, a, b in sorted(gaps, reverse=True)[:top]:
    print(f"  {g / 1e3:9.1f} us at +{at / 1e6:8.2f} ms   after {a}   before {b}")
