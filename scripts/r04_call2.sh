#!/bin/bash
# K-loop ablation of the 256x256 ping-pong kernel (variant builds: SLH8P_ABL bits)
mkdir -p gpurun_out/r04_c2
cd /root/repo
SH="4096x4096x4096,8192x5120x640,2048x10240x1280"
CV="2x128x128x320x320"
for v in base abl1 abl2 abl4 abl6 abl7 abl8 abl16; do
  if [ $v == base ]; then unset SLIDERS_HIP_LIB; else export SLIDERS_HIP_LIB=/root/repo/sliders_amd/libsliders_hip_$v.so; fi
  echo "== $v" >> gpurun_out/r04_c2/abl.log
  timeout 300 python scripts/probe_gemm8p.py --shapes $SH --convs $CV --tiles 8042 --check 0 >> gpurun_out/r04_c2/abl.log 2>&1
done
cat gpurun_out/r04_c2/abl.log
