#!/bin/bash
# Round 3, GPU call 7: in-situ re-tune (split-K allowed on the larger products) of the SDXL 512x512 passes (image sliders:
# training forward + backward) and the SD-1.x 512x512 passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c7
mkdir -p $O
cd $R
export TMPDIR=/tmp
T="24412,34412,44412,24012,20422,30422,40422,24322,34322,20412,30412,40412,20012,30012,40012,20011,40011,80011"
SLIDERS_SPLITK_ALL=1 timeout 700 python scripts/tune_insitu.py --incremental --model sdxl --hw 64 --tiles $T --out $O/sdxl_64_insitu.json > $O/tune_sdxl64.log 2>&1; grep -E "^==|total|incremental" $O/tune_sdxl64.log
SLIDERS_SPLITK_ALL=1 timeout 500 python scripts/tune_insitu.py --incremental --model sd1 --hw 64 --tiles $T --out $O/sd1_64_insitu.json > $O/tune_sd164.log 2>&1; grep -E "^==|total|incremental" $O/tune_sd164.log
