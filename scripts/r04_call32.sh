#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04_c32
timeout 900 python -m pytest tests/test_schedulers_gpu.py -x -q -m gpu -k "prodigy" -s 2>&1 | grep -E "prodigy\]|passed|failed" | tee gpurun_out/r04_c32/a.log
