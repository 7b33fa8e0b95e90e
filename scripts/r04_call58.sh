#!/bin/bash
# weight touch from idle workgroup slots (slh_gemm_desc.pf_*): kernel test, whole-pass A/B, parity subset
export TMPDIR=/tmp
o=gpurun_out/r04_c58; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "weight_touch or streamk or fused_cross" 2>&1 | tail -3
for v in touch none touch none; do
  unset SLIDERS_NO_WEIGHT_TOUCH
  [ $v == none ] && export SLIDERS_NO_WEIGHT_TOUCH=1
  echo "== $v" >> $o/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $o/ab.log
done
unset SLIDERS_NO_WEIGHT_TOUCH
cat $o/ab.log
timeout 300 python scripts/insitu_gemms.py 2>&1 | grep -v amdgpu.ids | head -12
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_graph_gpu.py tests/test_trainer_gpu.py -x -q -m gpu -n 4 2>&1 | tail -3
