#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "gemm" > gpurun_out/t5_kernels.log 2>&1
timeout 600 python scripts/bench_forward.py --model sdxl --hw 128 > gpurun_out/t5_fwd_off.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_on -o on -- python $GRAFT_REPO_ROOT/scripts/bench_forward.py --model sdxl --hw 128 --lora --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/t5_fwd_on.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_on -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/t5_kernels.log; grep "ms /" gpurun_out/t5_fwd_off.log gpurun_out/t5_fwd_on.log
