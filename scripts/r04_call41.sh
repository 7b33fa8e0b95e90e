#!/bin/bash
# stream-K diagnosis: the SK kernel run as plain data-parallel (per = K tiles of a tile), as 2 aligned halves, and default
export TMPDIR=/tmp
o=gpurun_out/r04_c41; mkdir -p $o
S="2048x1280x5120,3072x1280x5120"
for per in 80 40 20 0; do
  if [ $per == 0 ]; then unset SLIDERS_SK_PER; else export SLIDERS_SK_PER=$per; fi
  echo "== SK_PER=$per" >> $o/probe.log
  timeout 300 python scripts/probe_gemm8p.py --shapes $S --convs "" --tiles 4412,104412 --check 0 2>&1 | grep -v amdgpu.ids >> $o/probe.log
done
cat $o/probe.log
