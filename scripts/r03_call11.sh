#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c11
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 200 python scripts/probe_attn.py > $O/attn_grouped.log 2>&1; grep -v amdgpu.ids $O/attn_grouped.log
SLIDERS_HIP_LIB=$R/sliders_amd/libsliders_hip_b.so timeout 200 python scripts/probe_attn.py > $O/attn_ungrouped.log 2>&1; grep -v amdgpu.ids $O/attn_ungrouped.log
timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attention" > $O/attn_tests.log 2>&1; tail -2 $O/attn_tests.log
bash scripts/r03_call10.sh
