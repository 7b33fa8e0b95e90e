#!/bin/bash
mkdir -p gpurun_out/r04_c10
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" -n 4 > gpurun_out/r04_c10/pytest_gemm.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_c10/pytest_gemm.log
tail -8 gpurun_out/r04_c10/pytest_gemm.log
SLIDERS_SPLITK_ALL=1 timeout 1200 python scripts/tune_insitu.py --incremental --tiles 8015,8014,8013,8042,28015,28014 --out gpurun_out/r04_c10/sdxl_128_insitu.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_c10/tune.log
tail -170 gpurun_out/r04_c10/tune.log
