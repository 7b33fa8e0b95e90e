#!/bin/bash
# q|k|v projection on the 128 x 256 ping-pong tile with the V^T store in its epilogue: kernel tests + whole-pass A/B vs 0x4012
export TMPDIR=/tmp
o=gpurun_out/r04_c49; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_seam_gpu.py -x -q -m gpu -k "head_transposed or integration_md" -s > $o/pytest_k.log 2>&1
grep -E "passed|failed|Error|assert" $o/pytest_k.log | tail -8
for v in new 4012 new 4012; do
  unset SLIDERS_TUNING_OVERRIDE
  if [ $v == 4012 ]; then export SLIDERS_TUNING_OVERRIDE=$PWD/scripts/tuning_ab/ovr_qkv_4012.json; fi
  echo "== $v" >> $o/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $o/ab.log
done
cat $o/ab.log
timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_bench_config_gpu.py -x -q -m gpu -k "reproducible or forward_parity" 2>&1 | tail -3
