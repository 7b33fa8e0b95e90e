#!/bin/bash
# third GPU pass: smoke, tile tuner, first full bench + rocprof of the bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > gpurun_out/t3_smoke.log 2>&1
timeout 900 python scripts/tune_gemm.py --model sdxl --hw 128 > gpurun_out/t3_tune.log 2>&1
cp sliders_amd/tuning/*.json gpurun_out/ 2>/dev/null
timeout 900 python bench.py --steps 4 --warmup 1 > gpurun_out/t3_bench.log 2>&1
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_bench -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/t3_smoke.log; tail -8 gpurun_out/t3_tune.log; tail -3 gpurun_out/t3_bench.log; tail -2 gpurun_out/prof_bench.log
