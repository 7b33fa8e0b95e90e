#!/bin/bash
mkdir -p gpurun_out/r04_c15
cd /root/repo
O=gpurun_out/r04_c15
for v in dma base; do
 for g in 0 1 2 4 8 16; do
  if [ $v == base ]; then unset SLIDERS_HIP_LIB; else export SLIDERS_HIP_LIB=/root/repo/sliders_amd/libsliders_hip_$v.so; fi
  if [ $g == 0 ]; then unset SLIDERS_GEMM_GROUPS; else export SLIDERS_GEMM_GROUPS=$g; fi
  echo "== $v G=$g" >> $O/groups.log
  timeout 300 python scripts/probe_gemm8p.py --shapes "4096x4096x8192,2048x10240x1280,8192x5120x640" --convs "2x128x128x320x320" --tiles 8042,8015,4012 --check 0 2>&1 | grep -v amdgpu.ids >> $O/groups.log
 done
done
cat $O/groups.log
