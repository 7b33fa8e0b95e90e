#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "gemm" > gpurun_out/t10_kernels.log 2>&1
timeout 1200 python scripts/tune_gemm.py --model sdxl --hw 128 --fwd-only > gpurun_out/t10_tune.log 2>&1
cp sliders_amd/tuning/*.json gpurun_out/ 2>/dev/null
timeout 600 python scripts/bench_forward.py --model sdxl --hw 128 > gpurun_out/t10_fwd_off.log 2>&1
timeout 600 python scripts/bench_forward.py --model sdxl --hw 128 --lora > gpurun_out/t10_fwd_on.log 2>&1
tail -2 gpurun_out/t10_kernels.log; head -12 gpurun_out/t10_tune.log | cut -c1-330; tail -2 gpurun_out/t10_tune.log; grep "ms /" gpurun_out/t10_fwd_off.log gpurun_out/t10_fwd_on.log
