#!/bin/bash
mkdir -p gpurun_out/r04_c16
cd /root/repo
O=gpurun_out/r04_c16
for v in none 38014 28014 34322 none 38014; do
  if [ $v == none ]; then unset SLIDERS_TUNING_OVERRIDE; else export SLIDERS_TUNING_OVERRIDE=/root/repo/scripts/tuning_ab/ovr_ff2_$v.json; fi
  echo "== ff2 $v" >> $O/ab.log
  timeout 300 python scripts/bench_forward.py --lora --warm 2 --iters 10 2>&1 | tail -1 >> $O/ab.log
done
cat $O/ab.log
