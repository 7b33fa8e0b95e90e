#!/bin/bash
mkdir -p gpurun_out/r04_c11
cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu -n 6 > gpurun_out/r04_c11/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_c11/pytest_gpu.log
tail -8 gpurun_out/r04_c11/pytest_gpu.log
timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 > gpurun_out/r04_c11/fwd.log 2>&1
tail -3 gpurun_out/r04_c11/fwd.log
timeout 600 python bench.py --no-cpu-baseline --no-extra > gpurun_out/r04_c11/bench.json 2> gpurun_out/r04_c11/bench.err
cut -c1-400 gpurun_out/r04_c11/bench.json
