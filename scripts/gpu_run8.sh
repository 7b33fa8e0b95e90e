#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_trainer_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/t8_trainer.log 2>&1
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/t8_bench.log 2>&1
grep -E "parity|passed|failed|Error" gpurun_out/t8_trainer.log | cut -c1-300; tail -2 gpurun_out/t8_bench.log | cut -c1-700
