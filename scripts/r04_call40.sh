#!/bin/bash
# stream-K on the 128 x 128 8-wave tile: correctness + timing against the plain tiles (weights HBM-cold, activations warm)
export TMPDIR=/tmp
o=gpurun_out/r04_c40; mkdir -p $o
S="2048x1280x5120,2048x1280x1280,2048x1280x2560,2048x3840x1280,3072x1280x5120,4096x640x5120"
timeout 300 python scripts/probe_gemm8p.py --shapes $S --convs "" --tiles 4412,104412,4012,104012,204012 2>&1 | grep -v amdgpu.ids > $o/probe.log
cat $o/probe.log
