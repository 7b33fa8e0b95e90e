#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c14
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 200 python scripts/probe_attn.py > $O/attn_xcd.log 2>&1; grep -v amdgpu.ids $O/attn_xcd.log
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/suite_serial.log 2>&1; tail -2 $O/suite_serial.log
timeout 300 python scripts/time_train_iter.py > $O/pieces.log 2>&1; grep -v amdgpu.ids $O/pieces.log | head -5
