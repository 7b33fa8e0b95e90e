#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c12
mkdir -p $O $R/gpurun_out/r03_c10
cd $R
export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline > $O/bench_before.json 2> $O/bench_before.err
bash scripts/r03_call10.sh
cp $R/gpurun_out/r03_c10/sdxl_128_insitu.json sliders_amd/tuning/gfx950_sdxl_128_insitu.json
cp $R/gpurun_out/r03_c10/sdxl_64_insitu.json sliders_amd/tuning/gfx950_sdxl_64_insitu.json
cp $R/gpurun_out/r03_c10/sd1_64_insitu.json sliders_amd/tuning/gfx950_sd1_64_insitu.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_after.json 2> $O/bench_after.err
python - <<'PY'
import json
for n in ('before','after'):
    r=json.load(open(f'/root/repo/gpurun_out/r03_c12/bench_{n}.json'))
    p=r['roofline']['paths']
    print(n, r['value'], 'pass', p['unet_pass']['ms'], p['unet_pass']['launches'], 'gemm', p['all_gemm']['ms_per_pass'], 'frac', r['roofline']['frac'], r['roofline']['kernel'])
    for e in r.get('extra_configs',[]):
        print('   ', e.get('error') or (e['metric'][:40], e['value'], e['roofline']['paths']['unet_pass']['ms']))
PY
timeout 400 python -m pytest tests -m gpu -q -n 6 -p no:cacheprovider -rf > $O/suite.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/suite.log | head
