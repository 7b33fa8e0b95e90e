#!/bin/bash
# stream-K test + query projection with the cross-attention in its epilogue: kernel parity, pass A/B on one box, bench-config parity
export TMPDIR=/tmp
o=gpurun_out/r04_c42; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "streamk or fused_cross_attention" -s > $o/pytest_k.log 2>&1
grep -E "parity|passed|failed|Error|assert" $o/pytest_k.log | tail -30
for v in fused nofuse all fused nofuse; do
  unset SLIDERS_NO_FUSED_XATTN SLIDERS_XATTN_ALL
  [ $v == nofuse ] && export SLIDERS_NO_FUSED_XATTN=1
  [ $v == all ] && export SLIDERS_XATTN_ALL=1
  echo "== $v" >> $o/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $o/ab.log
done
cat $o/ab.log
unset SLIDERS_NO_FUSED_XATTN SLIDERS_XATTN_ALL
timeout 900 python -m pytest tests/test_bench_config_gpu.py tests/test_unet_gpu.py -x -q -m gpu -s -k "forward or parity or reproducible" > $o/pytest_cfg.log 2>&1
grep -E "parity|passed|failed|Error|assert" $o/pytest_cfg.log | tail -20
