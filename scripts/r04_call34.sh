#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04_c34; mkdir -p $o
run() { name=$1; shift; echo "=== $name"; env "$@" timeout 300 python scripts/debug_prodigy.py 2>&1 | tail -5; }
{
run nocapture DBG_ALLOC=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DBG_OPTS=prodigy
run stream DBG_ALLOC=1 DBG_STREAM=1 DBG_OPTS=prodigy
run sync_before DBG_ALLOC=1 DBG_SYNC=before DBG_OPTS=prodigy
run sync_after DBG_ALLOC=1 DBG_SYNC=after DBG_OPTS=prodigy
run sync_both DBG_ALLOC=1 DBG_SYNC=both DBG_OPTS=prodigy
run batch1 DBG_ALLOC=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DBG_OPTS=prodigy
} > $o/log.txt 2>&1
cat $o/log.txt
