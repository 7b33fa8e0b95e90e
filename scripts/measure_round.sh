#!/bin/bash
# Per-round measurement on the GPU box (gpurun -- 'RTAG=r05 TREE_HEAD=<git head> bash scripts/measure_round.sh'): the contract bench line
# (with extra_configs), the rocprofv3 --kernel-trace --stats summary of the same bench command, kernel stats + idle gaps of
# LoRA-on passes alone, the PMC passes (each in its own run, --kernel-trace only), iteration pieces.  Copy
# gpurun_out/${RTAG}_* into profiles/ afterwards.
RTAG=${RTAG:-r05}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --no-extra > $O/${RTAG}_prof_bench.log 2>&1
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/${RTAG}_bench_sdxl1024_kernel_stats.csv \;
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fwd -o fwd -- python $R/scripts/bench_forward.py --lora --warm 1 --iters 3 > $O/${RTAG}_prof_fwd.log 2>&1
find /tmp/prof_fwd -name "*kernel_stats.csv" -exec cp {} $O/${RTAG}_fwd_lora_on_kernel_stats.csv \;
python $R/scripts/trace_gaps.py /tmp/prof_fwd > $O/${RTAG}_fwd_kernel_gaps.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc5 -o p -- python $R/scripts/bench_forward.py --lora --warm 0 --iters 1 > $O/${RTAG}_pmc5.log 2>&1
python $R/scripts/pmc_summary.py /tmp/pmc5 > $O/${RTAG}_pmc_fetch_size_fwd_lora_on.csv 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc6 -o p -- python $R/scripts/bench_forward.py --lora --warm 0 --iters 1 > $O/${RTAG}_pmc6.log 2>&1
python $R/scripts/pmc_summary.py /tmp/pmc6 > $O/${RTAG}_pmc_write_size_l2hit_fwd_lora_on.csv 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc7 -o p -- python $R/scripts/bench_forward.py --lora --warm 0 --iters 1 > $O/${RTAG}_pmc7.log 2>&1
python $R/scripts/pmc_summary.py /tmp/pmc7 > $O/${RTAG}_pmc_mfma_busy_fwd_lora_on.csv 2>&1
python $R/scripts/make_pmc_traffic.py $O/${RTAG}_pmc_fetch_size_fwd_lora_on.csv $O/${RTAG}_pmc_write_size_l2hit_fwd_lora_on.csv $O/${RTAG}_pmc_traffic.json ${TREE_HEAD:-unrecorded}
cd $R
# the bench line reads the counter file of THIS tree (copied next to the sources on the box)
cp $O/${RTAG}_pmc_traffic.json $R/profiles/${RTAG}_pmc_traffic.json
cp $O/${RTAG}_pmc_mfma_busy_fwd_lora_on.csv $R/profiles/${RTAG}_pmc_mfma_busy_fwd_lora_on.csv      # (mfma_util_pmc is read from the newest committed file)
timeout 600 python bench.py > $O/${RTAG}_bench_line.json 2> $O/${RTAG}_bench_line.err
timeout 300 python scripts/time_train_iter.py --breakdown > $O/${RTAG}_iteration_pieces.txt 2>&1
timeout 200 python scripts/probe_gn.py > $O/${RTAG}_probe_gn.txt 2>&1
timeout 200 python scripts/probe_attn.py > $O/${RTAG}_probe_attn.txt 2>&1
cut -c1-220 $O/${RTAG}_bench_line.json; cat $O/${RTAG}_fwd_kernel_gaps.txt | tail -3; head -4 $O/${RTAG}_pmc_mfma_busy_fwd_lora_on.csv | cut -c1-200
timeout 300 python scripts/insitu_gemms.py > $O/${RTAG}_insitu_gemm_shapes.txt 2>&1
tail -45 $O/${RTAG}_insitu_gemm_shapes.txt | cut -c1-160
