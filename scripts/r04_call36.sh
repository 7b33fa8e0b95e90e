#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04_c36; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $o/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1
timeout 600 python bench.py > $o/bench.txt 2>&1
tail -5 $o/pytest.txt; tail -2 $o/smoke.txt; tail -1 $o/bench.txt
