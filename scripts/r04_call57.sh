#!/bin/bash
# weight prefetch on a side stream (SLH_OP_PREFETCH): whole-pass A/B over the lookahead, plain launches and graph replay; parity subset
export TMPDIR=/tmp
o=gpurun_out/r04_c57; mkdir -p $o
for v in 4 off 2 8 1 4 off; do
  unset SLIDERS_NO_PREFETCH SLIDERS_PREFETCH_AHEAD
  if [ $v == off ]; then export SLIDERS_NO_PREFETCH=1; else export SLIDERS_PREFETCH_AHEAD=$v; fi
  echo "== ahead $v" >> $o/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $o/ab.log
done
unset SLIDERS_NO_PREFETCH SLIDERS_PREFETCH_AHEAD
echo "== ahead 4, plain launches" >> $o/ab.log
timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 --no-graph 2>&1 | tail -1 >> $o/ab.log
export SLIDERS_NO_PREFETCH=1
echo "== off, plain launches" >> $o/ab.log
timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 --no-graph 2>&1 | tail -1 >> $o/ab.log
unset SLIDERS_NO_PREFETCH
cat $o/ab.log
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_graph_gpu.py tests/test_trainer_gpu.py -x -q -m gpu -n 4 2>&1 | tail -4
