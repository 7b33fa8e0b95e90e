#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c15
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_cli_gpu.py tests/test_seam_gpu.py tests/test_trainer_gpu.py tests/test_vae_gpu.py -m gpu -q -x -s > $O/cli.log 2>&1; grep -E "passed|failed|^E |Error" $O/cli.log | head -20; grep -E "^it |Done|flush" $O/cli.log | head -20
