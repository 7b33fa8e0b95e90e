#!/bin/bash
# graph-replay bug hunt
export TMPDIR=/tmp
o=gpurun_out/r04_c33; mkdir -p $o
run() { name=$1; shift; echo "=== $name"; env "$@" timeout 300 python scripts/debug_prodigy.py 2>&1 | tail -12; }
{
run base DBG_ALLOC=1
run noalloc DBG_X=1
run nographs DBG_ALLOC=1 SLIDERS_HIPGRAPH=0 DBG_OPTS=prodigy
run oldnorm DBG_ALLOC=1 SLIDERS_HIP_LIB=$PWD/sliders_amd/libsliders_hip_oldnorm.so DBG_OPTS=prodigy
run after3 DBG_ALLOC=1 DBG_AFTER=3 DBG_OPTS=prodigy
run k5 DBG_ALLOC=1 DBG_K=5 DBG_OPTS=prodigy
for p in on onc off train bw; do run nog_$p DBG_ALLOC=1 DBG_NOGRAPH=$p DBG_OPTS=prodigy; done
run nog_fwd DBG_ALLOC=1 DBG_NOGRAPH=on,onc,off,train DBG_OPTS=prodigy
} > $o/log.txt 2>&1
cat $o/log.txt
