#!/bin/bash
mkdir -p gpurun_out/r04_c13
cd /root/repo
O=gpurun_out/r04_c13
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" -n 4 > $O/pytest_gemm.log 2>&1
echo "pytest rc $?" >> $O/pytest_gemm.log
tail -6 $O/pytest_gemm.log
timeout 300 python scripts/probe_gemm8p.py --shapes "2048x10240x1280,3072x10240x1280,8192x5120x640,4096x4096x4096" --convs "" --tiles 4012,8015,8025,8042 2>&1 | grep -v amdgpu.ids > $O/probe.log
cat $O/probe.log
SLIDERS_SPLITK_ALL=1 timeout 900 python scripts/tune_insitu.py --incremental --tiles 4012,22,4412,12,4022,8025,8015,8014,28025 --out $O/sdxl_128_insitu.json 2>&1 | grep -v amdgpu.ids > $O/tune_sdxl128.log
grep -E "g3|total|replaced" $O/tune_sdxl128.log
