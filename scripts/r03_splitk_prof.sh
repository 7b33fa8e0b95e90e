#!/bin/bash
out=gpurun_out/r03_splitk; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fwd -o fwd -- python $R/scripts/bench_forward.py --lora --warm 1 --iters 3 > $R/$out/prof_fwd.log 2>&1
cp $(find /tmp/prof_fwd -name "*kernel_stats.csv" | head -1) $R/$out/fwd_kernel_stats.csv
cp $(find /tmp/prof_fwd -name "*kernel_trace.csv" | head -1) /tmp/kt.csv
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('/tmp/kt.csv')))
# per (kernel, grid) stats for gemm kernels
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'gemm_kernel' not in n: continue
    key=(n.split('gemm_kernel')[1][:28], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('Workgroup_Size_X'))
    agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
out=open('/root/repo/gpurun_out/r03_splitk/gemm_by_grid.txt','w')
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    out.write(f"{k[0]:30s} grid {k[1]:>8s} wg {k[2]:>4s}  n {len(v):4d}  avg {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:7.2f} ms\n")
PY
