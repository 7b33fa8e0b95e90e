#!/bin/bash
# q|k|v projection (fused adapter) on the ping-pong tiles: needs the V^T store out of the epilogue (SLIDERS_NO_FUSED_VT=1): whole-pass A/B
export TMPDIR=/tmp
o=gpurun_out/r04_c48; mkdir -p $o
for v in base novt 8014 8013 base 8014; do
  unset SLIDERS_NO_FUSED_VT SLIDERS_TUNING_OVERRIDE
  if [ $v != base ]; then export SLIDERS_NO_FUSED_VT=1; fi
  if [ $v == 8014 ] || [ $v == 8013 ]; then export SLIDERS_TUNING_OVERRIDE=$PWD/scripts/tuning_ab/ovr_qkv_$v.json; fi
  echo "== $v" >> $o/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $o/ab.log
done
cat $o/ab.log
