#!/bin/bash
out=gpurun_out/r03_vae; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae -o vae -- python $R/scripts/time_image_iter.py > $R/$out/prof.log 2>&1
cp $(find /tmp/prof_vae -name "*kernel_stats.csv" | head -1) $R/$out/kernel_stats.csv
cp $(find /tmp/prof_vae -name "*kernel_trace.csv" | head -1) /tmp/kt.csv
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('/tmp/kt.csv')))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if not any(t in n for t in ('sgemm','gn32','vae_','softmax32')): continue
    key=(n.split('(')[0][-40:], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X'))
    agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
out=open('/root/repo/gpurun_out/r03_vae/vae_by_grid.txt','w')
tot=sum(sum(v) for v in agg.values())
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    out.write(f"{k[0]:42s} grid {k[1]:>9s} wg {k[2]:>4s}  n {len(v):4d}  avg {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:7.2f} ms ({100*sum(v)/tot:4.1f} %)\n")
PY
