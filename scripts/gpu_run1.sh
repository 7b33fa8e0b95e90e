#!/bin/bash
# first GPU pass: kernel parity + tiny end-to-end + full-size forward timing
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/t_kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/t_kernels.log
timeout 600 python -m pytest tests/test_unet_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/t_unet.log 2>&1
echo "unet rc=$?" >> gpurun_out/t_unet.log
timeout 600 python scripts/bench_forward.py --model sdxl --hw 64 > gpurun_out/b_sdxl64.log 2>&1
timeout 600 python scripts/bench_forward.py --model sdxl --hw 128 > gpurun_out/b_sdxl128.log 2>&1
tail -5 gpurun_out/t_kernels.log gpurun_out/t_unet.log gpurun_out/b_sdxl64.log gpurun_out/b_sdxl128.log
