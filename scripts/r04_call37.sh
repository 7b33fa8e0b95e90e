#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04_c37; mkdir -p $o
S="2048x1280x1280,2048x1280x5120,2048x3840x1280,8192x640x640,8192x640x2560"
{
echo "== default"; timeout 300 python scripts/probe_gemm8p.py --shapes $S --convs "" --tiles 4412,4012,12 2>&1 | grep -v amdgpu.ids
echo "== all-panel L2 prefetch + wait before the K loop"; SLIDERS_HIP_LIB=$PWD/sliders_amd/libsliders_hip_pf.so timeout 300 python scripts/probe_gemm8p.py --shapes $S --convs "" --tiles 4412,4012,12 2>&1 | grep -v amdgpu.ids
} > $o/log.txt 2>&1
cat $o/log.txt
