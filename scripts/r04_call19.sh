#!/bin/bash
mkdir -p gpurun_out/r04_c19
cd /root/repo
timeout 1500 python -m pytest tests/test_parity_r04_gpu.py -x -q -m gpu -s > gpurun_out/r04_c19/pytest.log 2>&1
echo "rc $?" >> gpurun_out/r04_c19/pytest.log
grep -E "parity|passed|failed|Error|error|assert" gpurun_out/r04_c19/pytest.log | head -60
tail -5 gpurun_out/r04_c19/pytest.log
