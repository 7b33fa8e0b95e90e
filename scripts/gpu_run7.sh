#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -p no:cacheprovider -k "fused_lora" > gpurun_out/t7_kernels.log 2>&1
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_backward_gpu.py -m gpu -q -s -p no:cacheprovider -k "with_lora or lora_gradients" > gpurun_out/t7_e2e.log 2>&1
timeout 600 python scripts/bench_forward.py --model sdxl --hw 128 --lora > gpurun_out/t7_fwd_on.log 2>&1
SLIDERS_LORA_UNFUSED=1 timeout 600 python scripts/bench_forward.py --model sdxl --hw 128 --lora > gpurun_out/t7_fwd_on_unfused.log 2>&1
grep -E "passed|failed" gpurun_out/t7_kernels.log gpurun_out/t7_e2e.log; grep "ms /" gpurun_out/t7_fwd_on.log gpurun_out/t7_fwd_on_unfused.log
