#!/bin/bash
mkdir -p gpurun_out/r04_c29
cd /root/repo
O=gpurun_out/r04_c29
timeout 900 python -m pytest tests/test_parity_r04_gpu.py -x -q -m gpu -s -k "bf16_arm" > $O/arm.log 2>&1
grep -E "RMS|passed|failed|Assert" $O/arm.log
timeout 1500 python -m pytest tests -x -q -m gpu -n 6 > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
