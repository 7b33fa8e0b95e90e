#!/bin/bash
# weight touch: farthest vs nearest carrier inside the window, and a short window
export TMPDIR=/tmp
o=gpurun_out/r04_c60; mkdir -p $o
for v in far near none far near; do
  unset SLIDERS_NO_WEIGHT_TOUCH SLIDERS_TOUCH_NEAREST
  [ $v == none ] && export SLIDERS_NO_WEIGHT_TOUCH=1
  [ $v == near ] && export SLIDERS_TOUCH_NEAREST=1
  echo "== $v" >> $o/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $o/ab.log
done
cat $o/ab.log
export SLIDERS_TOUCH_NEAREST=1
timeout 300 python scripts/insitu_gemms.py 2>&1 | grep -v amdgpu.ids | head -8
unset SLIDERS_TOUCH_NEAREST
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "weight_touch" 2>&1 | tail -2
