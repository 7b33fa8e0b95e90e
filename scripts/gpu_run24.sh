#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_unet_gpu.py tests/test_trainer_gpu.py -x -q -m gpu > gpurun_out/t24_unet.log 2>&1; tail -2 gpurun_out/t24_unet.log
python scripts/bench_forward.py --lora --iters 10 > gpurun_out/t24_fwd_on.log 2>&1; tail -1 gpurun_out/t24_fwd_on.log
python scripts/bench_forward.py --iters 10 > gpurun_out/t24_fwd_off.log 2>&1; tail -1 gpurun_out/t24_fwd_off.log
