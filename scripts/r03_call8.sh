#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c8
mkdir -p $O
cd $R
export TMPDIR=/tmp
T="24412,34412,24012,20422,30422,24322,34322,20412,30412,20012,30012,24011,20011"
SLIDERS_SPLITK_MIN_K=1024 SLIDERS_SPLITK_ALL=1 timeout 700 python scripts/tune_insitu.py --incremental --fwd-only --model sdxl --hw 128 --tiles $T --out $O/sdxl_128_insitu_k1024.json > $O/tune_sdxl128_k1024.log 2>&1; grep -E "^==|total|incremental|table" $O/tune_sdxl128_k1024.log | head -60
timeout 400 python bench.py --no-cpu-baseline > $O/5_bench.json 2> $O/5_bench.err; python - <<'PY'
import json
r=json.load(open('/root/repo/gpurun_out/r03_c8/5_bench.json'))
print(r['value'], r['ms_per_step'], r['roofline']['paths']['unet_pass'], r['roofline']['frac'])
for e in r.get('extra_configs',[]):
    print(e.get('error') or (e['metric'], e['value'], e['roofline']['paths']['unet_pass']))
PY
