#!/bin/bash
# Round measurement on the GPU box (gpurun -- 'bash scripts/measure_round.sh'): the contract bench line and the
# rocprofv3 --kernel-trace --stats summary of the same command; copy gpurun_out/r01_* into profiles/ afterwards.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 420 python bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err
tail -c 500 gpurun_out/bench_r01.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench_r01.log 2>&1
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r01_bench_kernel_stats.csv \;
cd $R; cut -c1-200 gpurun_out/bench_r01.json; grep -o '"roofline".*"timing"' gpurun_out/bench_r01.json | cut -c1-600; grep -o '"cpu_baseline".*' gpurun_out/bench_r01.json | cut -c1-500
head -3 gpurun_out/r01_bench_kernel_stats.csv | cut -c1-200
