#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04_c46; mkdir -p $o
T="8015,8014,8013,8042,4012,4412,22,12,4022"
timeout 500 python scripts/tune_insitu.py --incremental --fwd-only --tiles $T --out $o/sdxl_128.json 2>&1 | grep -v amdgpu.ids > $o/tune_sdxl128.log
grep -E "^==|g3|total|replaced|Error" $o/tune_sdxl128.log
timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1
timeout 300 python scripts/bench_forward.py --hw 64 --lora --warm 2 --iters 10 2>&1 | tail -1
timeout 300 python scripts/bench_forward.py --model sd1 --hw 64 --lora --warm 2 --iters 10 2>&1 | tail -1
