#!/bin/bash
# Round-2 measurement on the GPU box (gpurun -- 'bash scripts/measure_round2.sh'): the contract bench lines, the rocprofv3
# --kernel-trace --stats summary of the same bench command, the inter-kernel idle time of replayed passes, and the PMC
# passes (each in its own run, --kernel-trace only).  Copy gpurun_out/r02_* into profiles/ afterwards.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 500 python bench.py > $O/r02_bench_line.json 2> $O/r02_bench_line.err
timeout 300 python bench.py --model sd1 --res 512 --no-cpu-baseline > $O/r02_bench_line_sd1_512.json 2> $O/r02_bench_sd1.err
timeout 300 python bench.py --workload image --res 512 --no-cpu-baseline > $O/r02_bench_line_image_512.json 2> $O/r02_bench_image.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline > $O/r02_prof_bench.log 2>&1
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/r02_bench_sdxl1024_kernel_stats.csv \;
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fwd -o fwd -- python $R/scripts/bench_forward.py --lora --warm 1 --iters 3 > $O/r02_prof_fwd.log 2>&1
find /tmp/prof_fwd -name "*kernel_stats.csv" -exec cp {} $O/r02_fwd_lora_on_kernel_stats.csv \;
python $R/scripts/trace_gaps.py /tmp/prof_fwd > $O/r02_fwd_kernel_gaps.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc5 -o p -- python $R/scripts/bench_forward.py --lora --warm 0 --iters 1 > $O/r02_pmc5.log 2>&1
python $R/scripts/pmc_summary.py /tmp/pmc5 > $O/r02_pmc_fetch_size_fwd_lora_on.csv 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc6 -o p -- python $R/scripts/bench_forward.py --lora --warm 0 --iters 1 > $O/r02_pmc6.log 2>&1
python $R/scripts/pmc_summary.py /tmp/pmc6 > $O/r02_pmc_write_size_l2hit_fwd_lora_on.csv 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc7 -o p -- python $R/scripts/bench_forward.py --lora --warm 0 --iters 1 > $O/r02_pmc7.log 2>&1
python $R/scripts/pmc_summary.py /tmp/pmc7 > $O/r02_pmc_mfma_busy_fwd_lora_on.csv 2>&1
python $R/scripts/make_pmc_traffic.py $O/r02_pmc_fetch_size_fwd_lora_on.csv $O/r02_pmc_write_size_l2hit_fwd_lora_on.csv $O/r02_pmc_traffic.json
cd $R; cut -c1-220 $O/r02_bench_line.json; cut -c1-160 $O/r02_bench_line_sd1_512.json; cut -c1-160 $O/r02_bench_line_image_512.json; cat $O/r02_fwd_kernel_gaps.txt; head -4 $O/r02_pmc_mfma_busy_fwd_lora_on.csv | cut -c1-200
