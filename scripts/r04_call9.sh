#!/bin/bash
mkdir -p gpurun_out/r04_c9
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" -n 4 > gpurun_out/r04_c9/pytest_gemm.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_c9/pytest_gemm.log
tail -15 gpurun_out/r04_c9/pytest_gemm.log
timeout 600 python scripts/probe_gemm8p.py --shapes "2048x10240x1280,2048x3840x1280,8192x5120x640,8192x1920x640,4096x4096x4096" --convs "2x128x128x320x320,2x64x64x640x640,2x128x128x640x320,2x64x64x1280x640,2x128x128x640x640,2x64x64x1280x1280" --tiles 4012,4412,8042,8015,8014 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_c9/probe.log
cat gpurun_out/r04_c9/probe.log
