#!/bin/bash
# GEGLU.proj on the ping-pong tiles (16 | 16 weight blocks, SLIDERS_GEGLU16=1): whole-pass A/B against the default 32 | 32 form
export TMPDIR=/tmp
o=gpurun_out/r04_c44; mkdir -p $o
for v in base g3_8015 g3_8014 g3_4012 base g3_8015; do
  unset SLIDERS_GEGLU16 SLIDERS_TUNING_OVERRIDE
  if [ $v != base ]; then export SLIDERS_GEGLU16=1 SLIDERS_TUNING_OVERRIDE=$PWD/scripts/tuning_ab/ovr_ff1_$v.json; fi
  echo "== $v" >> $o/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $o/ab.log
done
cat $o/ab.log
