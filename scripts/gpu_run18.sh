#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > gpurun_out/t18_kernels.log 2>&1; tail -2 gpurun_out/t18_kernels.log
python scripts/bench_forward.py --lora --iters 10 > gpurun_out/t18_fwd_on.log 2>&1; tail -1 gpurun_out/t18_fwd_on.log
python scripts/bench_forward.py --iters 10 > gpurun_out/t18_fwd_off.log 2>&1; tail -1 gpurun_out/t18_fwd_off.log
