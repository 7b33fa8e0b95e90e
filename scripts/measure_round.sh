#!/bin/bash
# Per-round measurement on the GPU box (gpurun -- 'RTAG=r05 TREE_HEAD=<git head> bash scripts/measure_round.sh'): the contract bench line
# (with extra_configs), the rocprofv3 --kernel-trace --stats summary of the same bench command, kernel stats + idle gaps of
# LoRA-on passes alone, the PMC passes (each in its own run, --kernel-trace only), iteration pieces.  Copy
# gpurun_out/${RTAG}_* into profiles/ afterwards.
RTAG=${RTAG:-r06}
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
# counter passes: each in its own run (--kernel-trace only), three LoRA-on passes profiled (first call, warm-up, replay), the first call's
# dispatches dropped per kernel (cold weights); one set per single-GPU configuration of the bench line: "" = SDXL 1024^2, SD-1.x 512^2, SDXL 512^2
pmc_set() {   # $1 = file tag, $2.. = bench_forward arguments
  local tag=$1; shift
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc5$tag -o p -- python $R/scripts/bench_forward.py --lora --warm 1 --iters 1 "$@" > $O/${RTAG}_pmc5$tag.log 2>&1
  python $R/scripts/pmc_summary.py /tmp/pmc5$tag --drop-first-third > $O/${RTAG}_pmc_fetch_size_fwd_lora_on$tag.csv 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc6$tag -o p -- python $R/scripts/bench_forward.py --lora --warm 1 --iters 1 "$@" > $O/${RTAG}_pmc6$tag.log 2>&1
  python $R/scripts/pmc_summary.py /tmp/pmc6$tag --drop-first-third > $O/${RTAG}_pmc_write_size_l2hit_fwd_lora_on$tag.csv 2>&1
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc7$tag -o p -- python $R/scripts/bench_forward.py --lora --warm 1 --iters 1 "$@" > $O/${RTAG}_pmc7$tag.log 2>&1
  python $R/scripts/pmc_summary.py /tmp/pmc7$tag --drop-first-third > $O/${RTAG}_pmc_mfma_busy_fwd_lora_on$tag.csv 2>&1
  python $R/scripts/make_pmc_traffic.py $O/${RTAG}_pmc_fetch_size_fwd_lora_on$tag.csv $O/${RTAG}_pmc_write_size_l2hit_fwd_lora_on$tag.csv $O/${RTAG}_pmc_traffic$tag.json ${TREE_HEAD:-unrecorded}
  # the bench line reads the counter files of THIS tree (copied next to the sources on the box)
  cp $O/${RTAG}_pmc_traffic$tag.json $R/profiles/${RTAG}_pmc_traffic$tag.json
  cp $O/${RTAG}_pmc_mfma_busy_fwd_lora_on$tag.csv $R/profiles/${RTAG}_pmc_mfma_busy_fwd_lora_on$tag.csv
}
pmc_set ""
pmc_set _sd1_64 --model sd1 --hw 64
pmc_set _sdxl_64 --model sdxl --hw 64
cd $R
timeout 600 python bench.py > $O/${RTAG}_bench_line.json 2> $O/${RTAG}_bench_line.err
timeout 300 python scripts/time_train_iter.py --breakdown > $O/${RTAG}_iteration_pieces.txt 2>&1
timeout 200 python scripts/probe_gn.py > $O/${RTAG}_probe_gn.txt 2>&1
timeout 200 python scripts/probe_attn.py > $O/${RTAG}_probe_attn.txt 2>&1
cut -c1-220 $O/${RTAG}_bench_line.json; cat $O/${RTAG}_fwd_kernel_gaps.txt | tail -3; head -4 $O/${RTAG}_pmc_mfma_busy_fwd_lora_on.csv | cut -c1-200
timeout 300 python scripts/insitu_gemms.py --attn > $O/${RTAG}_insitu_gemm_shapes.txt 2>&1
timeout 300 python scripts/insitu_gemms.py --attn --model sd1 --hw 64 > $O/${RTAG}_insitu_gemm_shapes_sd1_64.txt 2>&1
timeout 300 python scripts/insitu_gemms.py --attn --model sdxl --hw 64 > $O/${RTAG}_insitu_gemm_shapes_sdxl_64.txt 2>&1
tail -45 $O/${RTAG}_insitu_gemm_shapes.txt | cut -c1-160
