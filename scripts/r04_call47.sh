#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04_c47; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu -n 6 > $o/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $o/pytest_gpu.log
tail -12 $o/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $o/smoke.log
