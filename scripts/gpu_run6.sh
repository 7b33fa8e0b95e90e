#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
SLIDERS_FORCE_STAGES=3 timeout 600 python scripts/bench_forward.py --model sdxl --hw 128 > gpurun_out/t6_fwd_s3.log 2>&1
SLIDERS_FORCE_STAGES=2 timeout 600 python scripts/bench_forward.py --model sdxl --hw 128 > gpurun_out/t6_fwd_s2.log 2>&1
SLIDERS_NO_TUNING=1 timeout 600 python scripts/bench_forward.py --model sdxl --hw 128 > gpurun_out/t6_fwd_notune.log 2>&1
grep "ms /" gpurun_out/t6_fwd_s3.log gpurun_out/t6_fwd_s2.log gpurun_out/t6_fwd_notune.log
