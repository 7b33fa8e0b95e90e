#!/bin/bash
# second GPU pass: backward parity + LoRA e2e + rocprof kernel stats of the SDXL forward
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -p no:cacheprovider -k "cfg_ddim or loss" > gpurun_out/t2_kernels.log 2>&1
timeout 900 python -m pytest tests/test_backward_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/t2_backward.log 2>&1
timeout 600 python -m pytest tests/test_unet_gpu.py -m gpu -q -s -p no:cacheprovider -k with_lora > gpurun_out/t2_unet.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fwd -o fwd -- python $GRAFT_REPO_ROOT/scripts/bench_forward.py --model sdxl --hw 128 --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_fwd.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_fwd -name "*kernel_trace.csv" -size +20M -delete
ls -la gpurun_out/prof_fwd/* | head
tail -4 gpurun_out/t2_kernels.log gpurun_out/t2_backward.log gpurun_out/t2_unet.log gpurun_out/prof_fwd.log
