#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04_c35; mkdir -p $o
run() { name=$1; shift; echo "=== $name"; env "$@" timeout 300 python scripts/debug_prodigy.py 2>&1 | grep -v amdgpu.ids | tail -10; }
{
run base DBG_ALLOC=1
run sync_after DBG_ALLOC=1 DBG_SYNC=after DBG_OPTS=prodigy
run k5 DBG_ALLOC=1 DBG_K=5 DBG_OPTS=prodigy
timeout 600 python -m pytest tests/test_schedulers_gpu.py -x -q -m gpu 2>&1 | tail -5
} > $o/log.txt 2>&1
cat $o/log.txt
