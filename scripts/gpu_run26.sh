#!/bin/bash
mkdir -p gpurun_out
cp sliders_amd/tuning/gfx950_sdxl_128.json gpurun_out/gfx950_sdxl_128.json
timeout 300 python scripts/tune_gemm.py --fwd-only --out gpurun_out/gfx950_sdxl_128.json > gpurun_out/t26_tune.log 2>&1; grep -E "g1|166400|sum over" gpurun_out/t26_tune.log | cut -c1-200
cp gpurun_out/gfx950_sdxl_128.json sliders_amd/tuning/gfx950_sdxl_128.json
python scripts/bench_forward.py --lora --iters 10 > gpurun_out/t26_fwd_on.log 2>&1; tail -1 gpurun_out/t26_fwd_on.log
