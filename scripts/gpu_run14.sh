#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/t14_kernels.log 2>&1; tail -2 gpurun_out/t14_kernels.log
python scripts/bench_forward.py --lora --iters 10 > gpurun_out/t14_fwd_on.log 2>&1; tail -1 gpurun_out/t14_fwd_on.log
SLIDERS_GEMM_GROUPS=1 python scripts/bench_forward.py --lora --iters 10 > gpurun_out/t14_fwd_on_g1.log 2>&1; tail -1 gpurun_out/t14_fwd_on_g1.log
SLIDERS_GEMM_GROUPS=4 python scripts/bench_forward.py --lora --iters 10 > gpurun_out/t14_fwd_on_g4.log 2>&1; tail -1 gpurun_out/t14_fwd_on_g4.log
SLIDERS_GEMM_GROUPS=8 python scripts/bench_forward.py --lora --iters 10 > gpurun_out/t14_fwd_on_g8.log 2>&1; tail -1 gpurun_out/t14_fwd_on_g8.log
cp sliders_amd/tuning/gfx950_sdxl_128.json gpurun_out/gfx950_sdxl_128.json
timeout 900 python scripts/tune_gemm.py --fwd-only --out gpurun_out/gfx950_sdxl_128.json > gpurun_out/t14_tune.log 2>&1; tail -2 gpurun_out/t14_tune.log
cp gpurun_out/gfx950_sdxl_128.json sliders_amd/tuning/gfx950_sdxl_128.json
python scripts/bench_forward.py --lora --iters 10 > gpurun_out/t14_fwd_on_retuned.log 2>&1; tail -1 gpurun_out/t14_fwd_on_retuned.log
