#!/bin/bash
mkdir -p gpurun_out/r04_c25
cd /root/repo
O=gpurun_out/r04_c25
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groupnorm or gn" -n 4 > $O/pytest_gn.log 2>&1
echo "rc $?" >> $O/pytest_gn.log
tail -4 $O/pytest_gn.log
timeout 200 python scripts/probe_gn.py 2>&1 | grep -v amdgpu > $O/probe_gn.txt
cat $O/probe_gn.txt
timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1
timeout 300 python scripts/bench_forward.py --model sd1 --hw 64 --lora --warm 2 --iters 10 2>&1 | tail -1
