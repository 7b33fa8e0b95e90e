#!/bin/bash
# attention, key-split form for the 32 x 32 level (attn_fwd_ks_kernel): kernel tests, probe, whole-pass A/B (SLH_ATTN_KS=0 = old form)
export TMPDIR=/tmp
o=gpurun_out/r04_c61; mkdir -p $o
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -3
for v in 1 0 1 0; do
  export SLH_ATTN_KS=$v
  echo "== KS=$v" >> $o/ab.log
  timeout 200 python scripts/probe_attn.py 2>&1 | grep -E "32\^2|16\^2|sum" >> $o/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $o/ab.log
done
unset SLH_ATTN_KS
cat $o/ab.log
