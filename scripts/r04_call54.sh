#!/bin/bash
# ping-pong kernels: chunk statistics of a folded LayerNorm requested ahead of the prologue's LDS-DMA (hidden loads) vs the serial form
export TMPDIR=/tmp
o=gpurun_out/r04_c54; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm_folded or geglu_16" > $o/pytest_k.log 2>&1
tail -2 $o/pytest_k.log
for v in new serial new serial; do
  unset SLIDERS_HIP_LIB
  [ $v == serial ] && export SLIDERS_HIP_LIB=$PWD/sliders_amd/libsliders_hip_lnserial.so
  echo "== $v" >> $o/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $o/ab.log
done
cat $o/ab.log
