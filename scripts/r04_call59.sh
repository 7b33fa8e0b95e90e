#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04_c59; mkdir -p $o
for v in touch none touch none; do
  unset SLIDERS_NO_WEIGHT_TOUCH
  [ $v == none ] && export SLIDERS_NO_WEIGHT_TOUCH=1
  echo "== $v" >> $o/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $o/ab.log
done
unset SLIDERS_NO_WEIGHT_TOUCH
cat $o/ab.log
timeout 300 python scripts/insitu_gemms.py 2>&1 | grep -v amdgpu.ids | head -8
