#!/bin/bash
out=gpurun_out/r03_bwd; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $out/suite.txt; cat $out/suite.txt
python scripts/time_train_iter.py > $out/iter_pieces.txt 2>&1; tail -12 $out/iter_pieces.txt
python bench.py --workload image --res 512 --steps 6 --warmup 2 --no-extra --no-cpu-baseline > $out/bench_image.txt 2>&1; tail -1 $out/bench_image.txt | cut -c1-300
