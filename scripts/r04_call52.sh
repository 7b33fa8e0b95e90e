#!/bin/bash
# norm1 folded into the adapter-carrying q|k|v projection: kernel test, whole-pass A/B, bench-config parity
export TMPDIR=/tmp
o=gpurun_out/r04_c52; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "folded_with_fused_adapter or layernorm_folded" -s > $o/pytest_k.log 2>&1
grep -E "parity|passed|failed|Error|assert" $o/pytest_k.log | tail -14
for v in fold nofold fold nofold; do
  unset SLIDERS_NO_LORA_LN_FOLD
  [ $v == nofold ] && export SLIDERS_NO_LORA_LN_FOLD=1
  echo "== $v" >> $o/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $o/ab.log
done
cat $o/ab.log
unset SLIDERS_NO_LORA_LN_FOLD
timeout 900 python -m pytest tests/test_bench_config_gpu.py tests/test_unet_gpu.py -x -q -m gpu -s -k "forward_parity or reproducible or sdxl" > $o/pytest_cfg.log 2>&1
grep -E "sdxl.*adapters|passed|failed|Error|assert" $o/pytest_cfg.log | tail -8
