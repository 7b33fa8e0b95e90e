#!/bin/bash
mkdir -p gpurun_out/r04_c24
cd /root/repo
timeout 900 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu -k "bit_reproducible" -s > gpurun_out/r04_c24/pytest.log 2>&1
echo "rc $?" >> gpurun_out/r04_c24/pytest.log
grep -E "passed|failed|Error|assert|differ" gpurun_out/r04_c24/pytest.log | tail -12
