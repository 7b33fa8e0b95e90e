#!/bin/bash
mkdir -p gpurun_out
cp sliders_amd/tuning/gfx950_sdxl_128.json gpurun_out/gfx950_sdxl_128.json
timeout 300 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "forward_parity_with_lora or forward_parity_no_lora" > gpurun_out/t13_unet.log 2>&1; tail -3 gpurun_out/t13_unet.log
timeout 900 python scripts/tune_gemm.py --out gpurun_out/gfx950_sdxl_128.json > gpurun_out/t13_tune.log 2>&1; tail -2 gpurun_out/t13_tune.log
cp gpurun_out/gfx950_sdxl_128.json sliders_amd/tuning/gfx950_sdxl_128.json
python scripts/bench_forward.py --lora --iters 10 > gpurun_out/t13_fwd_on_packed.log 2>&1; tail -1 gpurun_out/t13_fwd_on_packed.log
SLIDERS_W_ROWMAJOR=1 python scripts/bench_forward.py --lora --iters 10 > gpurun_out/t13_fwd_on_rowmajor.log 2>&1; tail -1 gpurun_out/t13_fwd_on_rowmajor.log
python scripts/bench_forward.py --iters 10 > gpurun_out/t13_fwd_off_packed.log 2>&1; tail -1 gpurun_out/t13_fwd_off_packed.log
