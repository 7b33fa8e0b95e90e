#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c13
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_trainer_gpu.py tests/test_seam_gpu.py tests/test_vae_gpu.py tests/test_backward_gpu.py tests/test_bench_config_gpu.py tests/test_unet_gpu.py -m gpu -q -s > $O/parity_lines.log 2>&1; grep -E "^\[parity\]|\[seam\]|passed|failed" $O/parity_lines.log | cut -c1-260 > $O/parity_summary.txt; tail -3 $O/parity_summary.txt
bash scripts/measure_round3.sh
