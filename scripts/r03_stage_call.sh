#!/bin/bash
out=gpurun_out/r03_stage; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $out/suite.txt; grep -E "passed|failed|error" $out/suite.txt
for i in 1 2; do
SLIDERS_HIP_LIB=$PWD/sliders_amd/libsliders_hip_visible.so python scripts/bench_forward.py --lora --warm 3 --iters 12 2>&1 | tail -1
python scripts/bench_forward.py --lora --warm 3 --iters 12 2>&1 | tail -1
done
SLIDERS_HIP_LIB=$PWD/sliders_amd/libsliders_hip_visible.so python scripts/bench_forward.py --model sd1 --hw 64 --lora --warm 3 --iters 12 2>&1 | tail -1
python scripts/bench_forward.py --model sd1 --hw 64 --lora --warm 3 --iters 12 2>&1 | tail -1
SLIDERS_HIP_LIB=$PWD/sliders_amd/libsliders_hip_visible.so python scripts/bench_forward.py --hw 64 --lora --warm 3 --iters 12 2>&1 | tail -1
python scripts/bench_forward.py --hw 64 --lora --warm 3 --iters 12 2>&1 | tail -1
