#!/bin/bash
mkdir -p gpurun_out/r04_c30
cd /root/repo
O=gpurun_out/r04_c30
timeout 900 python -m pytest tests/test_schedulers_gpu.py -x -q -m gpu -k "prodigy" > $O/a.log 2>&1; tail -3 $O/a.log
timeout 900 python -m pytest tests/test_schedulers_gpu.py -x -q -m gpu > $O/b.log 2>&1; tail -3 $O/b.log
timeout 1500 python -m pytest tests -q -m gpu -n 6 > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
