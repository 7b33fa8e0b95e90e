#!/bin/bash
# Round 3, GPU call 3: suite on the fixed tests, GroupNorm kernel probe, in-situ split-K tuning of the larger products, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c3
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q -n 6 -p no:cacheprovider -rf > $O/1_suite.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/1_suite.log | head -30
timeout 200 python scripts/probe_gn.py > $O/2_probe_gn.log 2>&1; cat $O/2_probe_gn.log | grep -v amdgpu.ids
T="24412,34412,44412,24012,20422,30422,40422,24322,34322,20412,30412,40412"
SLIDERS_SPLITK_ALL=1 timeout 600 python scripts/tune_insitu.py --incremental --fwd-only --tiles $T --out $O/3_sdxl_128_insitu.json > $O/3_tune_sdxl128.log 2>&1; grep -E "total|incremental|table .*\| (2|3|4)[0-9a-f]{4}:" $O/3_tune_sdxl128.log | head -60
timeout 300 python bench.py --no-cpu-baseline > $O/5_bench.json 2> $O/5_bench.err; cut -c1-200 $O/5_bench.json
