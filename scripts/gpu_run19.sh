#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o fwd -- python $R/scripts/bench_forward.py --lora --warm 1 --iters 5 > $R/gpurun_out/t19_prof.log 2>&1
find /tmp/prof_f -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/t19_fwd_on_kernel_stats.csv \;
head -14 $R/gpurun_out/t19_fwd_on_kernel_stats.csv | cut -c1-150
