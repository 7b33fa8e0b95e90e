#!/bin/bash
# GEGLU.proj in 16 | 16 blocks (default now): in-situ tile candidates for the g3 keys of the three configurations
export TMPDIR=/tmp
o=gpurun_out/r04_c45; mkdir -p $o
T="8015,8014,8013,8042,4012,4412,22,12,4022"
timeout 400 python scripts/tune_insitu.py --incremental --fwd-only --tiles $T --out $o/sdxl_128.json 2>&1 | grep -v amdgpu.ids > $o/tune_sdxl128.log
timeout 300 python scripts/tune_insitu.py --incremental --fwd-only --hw 64 --tiles $T,412,4011 --out $o/sdxl_64.json 2>&1 | grep -v amdgpu.ids > $o/tune_sdxl64.log
timeout 300 python scripts/tune_insitu.py --incremental --fwd-only --model sd1 --hw 64 --tiles $T,412,4011 --out $o/sd1_64.json 2>&1 | grep -v amdgpu.ids > $o/tune_sd164.log
grep -E "^==|g3|total|replaced" $o/tune_*.log
timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1
