#!/bin/bash
# Round 3, GPU call 10: in-situ re-tune of every pass (the backward-data GEMMs now carry the adapter), all three configurations
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c10
mkdir -p $O
cd $R
export TMPDIR=/tmp
T="11,12,21,22,4011,4012,4022,322,422,412,421,4412,4322,4411,24412,34412,44412,24012,20422,30422,40422,24322,34322,44322,54322,20412,30412,40412,60412,80412,20012,30012,40012,20011,40011,80011,f0412"
SLIDERS_SPLITK_ALL=1 timeout 900 python scripts/tune_insitu.py --incremental --model sdxl --hw 128 --tiles $T --out $O/sdxl_128_insitu.json > $O/tune_sdxl128.log 2>&1; grep -E "^==|total|incremental" $O/tune_sdxl128.log
SLIDERS_SPLITK_ALL=1 timeout 700 python scripts/tune_insitu.py --incremental --model sdxl --hw 64 --tiles $T --out $O/sdxl_64_insitu.json > $O/tune_sdxl64.log 2>&1; grep -E "^==|total|incremental" $O/tune_sdxl64.log
SLIDERS_SPLITK_ALL=1 timeout 500 python scripts/tune_insitu.py --incremental --model sd1 --hw 64 --tiles $T --out $O/sd1_64_insitu.json > $O/tune_sd164.log 2>&1; grep -E "^==|total|incremental" $O/tune_sd164.log
