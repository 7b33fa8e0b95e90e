#!/bin/bash
mkdir -p gpurun_out/r04_c21
cd /root/repo
O=gpurun_out/r04_c21
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wgrad" -s > $O/pytest_wgrad.log 2>&1
echo "rc $?" >> $O/pytest_wgrad.log
grep -E "parity|passed|failed|Error|assert" $O/pytest_wgrad.log | tail -20
timeout 1200 python -m pytest tests -x -q -m gpu -n 6 > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
timeout 300 python scripts/time_train_iter.py --breakdown > $O/iter_fixed.txt 2>&1
SLIDERS_WGRAD_ATOMIC=1 timeout 300 python scripts/time_train_iter.py --breakdown > $O/iter_atomic.txt 2>&1
grep -iE "backward|wgrad|denoise|train forward|frozen" $O/iter_fixed.txt | head -12
echo ---- atomic
grep -iE "backward|wgrad" $O/iter_atomic.txt | head -6
