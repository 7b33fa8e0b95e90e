"""Merge scripts/tune_insitu.py --incremental logs into a tuning table: a key is shared by the passes of an iteration (LoRA-on
denoise pass x ~25, frozen B=3 pass, training forward, backward: one each), so a candidate tile replaces the table entry only when
the launch-weighted time over ALL passes that contain the key improves by more than 2 %.  usage: merge_tune_logs.py <log> <table.json>"""
import json
import re
import sys
from collections import defaultdict

WEIGHT = {"on": 25.0, "train": 1.0, "backward": 1.0, "off3": 1.0}
log, table_path = sys.argv[1], sys.argv[2]
cost = defaultdict(lambda: defaultdict(dict))        # key -> tile (-1 = table) -> {program: us * count * weight}
prog = None
for line in open(log):
    m = re.match(r"== (\w+):", line)
    if m:
        prog = m.group(1)
        continue
    m = re.match(r"\s+(\S+)\s+x\s*(\d+)\s+table\s+([\d.]+)us \| (.*)", line)
    if not m or prog is None:
        continue
    key, cnt, ttab, rest = m.group(1), int(m.group(2)), float(m.group(3)), m.group(4)
    w = WEIGHT.get(prog, 1.0) * cnt
    cost[key][-1][prog] = ttab * w
    for tok in rest.split():
        t, v = tok.split(":")
        cost[key][int(t, 16)][prog] = float(v) * w
table = json.load(open(table_path))
changed = 0
for key, per_tile in cost.items():
    progs = set(per_tile[-1])
    base = sum(per_tile[-1].values())
    best, best_cost = None, base
    for t, c in per_tile.items():
        if t == -1 or set(c) != progs:
            continue                                  # not measured (or not among the top 5) in every pass that has the key
        tot = sum(c.values())
        if tot < 0.98 * best_cost:
            best, best_cost = t, tot
    if best is not None and table.get(key) != best:
        print(f"{key:48s} {table.get(key, 0):6x} -> {best:6x}   {base / 1e3:8.2f} -> {best_cost / 1e3:8.2f} (weighted ms)")
        table[key] = best
        changed += 1
json.dump(table, open(table_path, "w"), indent=0, sort_keys=True)
print(f"{changed} entries changed in {table_path}")
