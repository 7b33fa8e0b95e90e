#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(free -g; nproc; cat /sys/fs/cgroup/memory.max 2>/dev/null; lscpu | grep -E "Model name|Flags" | cut -c1-300) > gpurun_out/t9_host.log 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_backward_gpu.py -m gpu -q -s -p no:cacheprovider -k "attention or lora_gradients" > gpurun_out/t9_attn.log 2>&1
timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_trainer_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/t9_unet.log 2>&1
grep -E "passed|failed" gpurun_out/t9_attn.log gpurun_out/t9_unet.log; grep -E "full-size|dedup|iteration" gpurun_out/t9_unet.log | cut -c1-250; cat gpurun_out/t9_host.log | head -12
