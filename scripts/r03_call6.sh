#!/bin/bash
# Round 3, GPU call 6: LayerNorm folded into its consumer GEMMs (kernel parity, suite, same-box A/B of the bench line)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c6
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "layernorm_folded or groupnorm" -s > $O/0_ln.log 2>&1; grep -E "passed|failed|Error" $O/0_ln.log | tail -5
timeout 600 python -m pytest tests -m gpu -q -n 6 -p no:cacheprovider -rf > $O/1_suite.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/1_suite.log | head -30
summ() { python - "$1" <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
p=r['roofline']['paths']
print(sys.argv[1].split('/')[-1], r['value'], 'pass', p['unet_pass']['ms'], p['unet_pass']['launches'], 'gemm', p['all_gemm']['ms_per_pass'], 'ln', p['layernorm']['ms_per_pass'], p['layernorm']['launches'], 'gn', p['groupnorm']['ms_per_pass'], 'attn', p['attention']['ms_per_pass'])
PY
}
for i in 1; do
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 8 > $O/5_bench_fold_$i.json 2> $O/5_bench_fold_$i.err; summ $O/5_bench_fold_$i.json
SLIDERS_NO_LN_FOLD=1 timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 8 > $O/5_bench_nofold_$i.json 2> $O/5_bench_nofold_$i.err; summ $O/5_bench_nofold_$i.json
done
