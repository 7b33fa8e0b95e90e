#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c4
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -n 6 -p no:cacheprovider -rf > $O/1_suite.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/1_suite.log | head -30
timeout 200 python scripts/probe_gn.py > $O/2_probe_gn.log 2>&1; cat $O/2_probe_gn.log | grep -v amdgpu.ids
timeout 400 python bench.py --no-cpu-baseline > $O/5_bench.json 2> $O/5_bench.err; python - <<'PY'
import json
r=json.load(open('/root/repo/gpurun_out/r03_c4/5_bench.json'))
print(r['value'], r['ms_per_step'], r['roofline']['paths']['unet_pass'], r['roofline']['frac'])
for e in r.get('extra_configs',[]):
    print(e.get('error') or (e['metric'], e['value'], e['roofline']['paths']['unet_pass'], e.get('vae_roofline')))
PY
tail -5 $O/5_bench.err
