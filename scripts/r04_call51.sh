#!/bin/bash
# L2 blocking of the single-round M = 2048 products: forced number of m-tile groups (SLIDERS_GEMM_GROUPS), cold weights
export TMPDIR=/tmp
o=gpurun_out/r04_c51; mkdir -p $o
for g in 0 1 2 4 8 16; do
  if [ $g == 0 ]; then unset SLIDERS_GEMM_GROUPS; else export SLIDERS_GEMM_GROUPS=$g; fi
  echo "== G=$g" >> $o/groups.log
  timeout 300 python scripts/probe_gemm8p.py --shapes "2048x1280x5120,2048x1280x1280,2048x3840x1280,2048x10240x1280" --convs "" --tiles 4412,4012,8015,8014 --check 0 2>&1 | grep -v amdgpu.ids >> $o/groups.log
done
cat $o/groups.log
