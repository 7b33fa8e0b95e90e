#!/bin/bash
# PMC passes (FETCH_SIZE, then WRITE_SIZE + L2 hit/miss; separate runs, --kernel-trace only) over one LoRA-on UNet
# pass, summarised per kernel by scripts/pmc_summary.py; also re-runs the bench line + kernel stats.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err
tail -c 400 gpurun_out/bench_r01.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench_r01.log 2>&1
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r01_bench_kernel_stats.csv \;
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc5 -o p -- python $R/scripts/bench_forward.py --lora --warm 0 --iters 1 > $R/gpurun_out/t20_pmc5.log 2>&1
python $R/scripts/pmc_summary.py /tmp/pmc5 > $R/gpurun_out/pmc_fetch_summary.csv 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc6 -o p -- python $R/scripts/bench_forward.py --lora --warm 0 --iters 1 > $R/gpurun_out/t20_pmc6.log 2>&1
python $R/scripts/pmc_summary.py /tmp/pmc6 > $R/gpurun_out/pmc_write_summary.csv 2>&1
cd $R; cut -c1-300 gpurun_out/bench_r01.json; tail -1 gpurun_out/prof_bench_r01.log | cut -c1-200
