#!/bin/bash
mkdir -p gpurun_out/r04_c12
cd /root/repo
O=gpurun_out/r04_c12
T="8015,8014,8013,28015,28014,28013,38015,38014,38013,48015,48014"
SLIDERS_SPLITK_ALL=1 timeout 600 python scripts/tune_insitu.py --incremental --tiles $T --out $O/sdxl_128_insitu.json 2>&1 | grep -v amdgpu.ids > $O/tune_sdxl128.log
SLIDERS_SPLITK_ALL=1 timeout 600 python scripts/tune_insitu.py --incremental --hw 64 --tiles $T,8042 --out $O/sdxl_64_insitu.json 2>&1 | grep -v amdgpu.ids > $O/tune_sdxl64.log
SLIDERS_SPLITK_ALL=1 timeout 600 python scripts/tune_insitu.py --incremental --model sd1 --hw 64 --tiles $T,8042 --out $O/sd1_64_insitu.json 2>&1 | grep -v amdgpu.ids > $O/tune_sd164.log
grep -E "total|replaced" $O/tune_*.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fwd -o fwd -- python /root/repo/scripts/bench_forward.py --lora --warm 1 --iters 3 > /root/repo/$O/prof_fwd.log 2>&1
find /tmp/prof_fwd -name "*kernel_stats.csv" -exec cp {} /root/repo/$O/fwd_lora_on_kernel_stats.csv \;
head -30 /root/repo/$O/fwd_lora_on_kernel_stats.csv | cut -c1-180
