#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > gpurun_out/t4_smoke.log 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/t4_kernels.log 2>&1
timeout 900 python scripts/tune_gemm.py --model sdxl --hw 128 > gpurun_out/t4_tune.log 2>&1
cp sliders_amd/tuning/*.json gpurun_out/ 2>/dev/null
timeout 600 python scripts/bench_forward.py --model sdxl --hw 128 > gpurun_out/t4_fwd_off.log 2>&1
timeout 600 python scripts/bench_forward.py --model sdxl --hw 128 --lora > gpurun_out/t4_fwd_on.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/t4_bench.log 2>&1
tail -3 gpurun_out/t4_smoke.log; tail -3 gpurun_out/t4_kernels.log; tail -3 gpurun_out/t4_tune.log; tail -1 gpurun_out/t4_fwd_off.log gpurun_out/t4_fwd_on.log; tail -2 gpurun_out/t4_bench.log | cut -c1-600
