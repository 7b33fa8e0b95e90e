#!/bin/bash
mkdir -p gpurun_out/r04_c28
cd /root/repo
O=gpurun_out/r04_c28
timeout 1500 python -m pytest tests -x -q -m gpu -n 6 > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
