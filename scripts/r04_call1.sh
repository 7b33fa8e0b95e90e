#!/bin/bash
# round 4, first GPU call: refactored GEMM epilogue + the new 256x256 ping-pong K loop (tile 0x8042)
mkdir -p gpurun_out/r04_c1
cd /root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" -n 4 > gpurun_out/r04_c1/pytest_gemm.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_c1/pytest_gemm.log
tail -5 gpurun_out/r04_c1/pytest_gemm.log
timeout 600 python scripts/probe_gemm8p.py > gpurun_out/r04_c1/probe.log 2>&1
cat gpurun_out/r04_c1/probe.log
