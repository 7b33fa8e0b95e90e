#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c16
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "groupnorm or attention" > $O/0_new.log 2>&1; tail -2 $O/0_new.log
timeout 200 python scripts/probe_attn.py > $O/attn.log 2>&1; grep -v amdgpu.ids $O/attn.log | tail -4
timeout 600 python -m pytest tests -m gpu -q -n 6 -p no:cacheprovider -rf > $O/1_suite.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/1_suite.log | head -20
summ() { python - "$1" <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
p=r['roofline']['paths']
print(sys.argv[1].split('/')[-1], r['value'], 'pass', p['unet_pass']['ms'], p['unet_pass']['launches'], 'gemm', p['all_gemm']['ms_per_pass'], 'gn', p['groupnorm']['ms_per_pass'], p['groupnorm']['launches'], 'attn', p['attention']['ms_per_pass'])
for e in r.get('extra_configs',[]):
    q=e['roofline']['paths']
    print('   ', e['metric'][:44], e['value'], 'pass', q['unet_pass']['ms'], q['unet_pass']['launches'], 'gn', q['groupnorm']['ms_per_pass'], 'frac', q['unet_pass']['frac_of_mfma_peak'])
PY
}
timeout 400 python bench.py --no-cpu-baseline > $O/5_bench_fused.json 2> $O/5_bench_fused.err; summ $O/5_bench_fused.json
SLIDERS_GN_TWO_LAUNCH=1 timeout 400 python bench.py --no-cpu-baseline > $O/5_bench_two.json 2> $O/5_bench_two.err; summ $O/5_bench_two.json
