#!/bin/bash
out=gpurun_out/r03_splitk; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "splitk or geglu or layernorm_folded or head_transposed or attention" 2>&1 | tail -15 > $out/kernels.txt
cat $out/kernels.txt
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $out/suite.txt
cat $out/suite.txt
python scripts/bench_forward.py --lora --warm 3 --iters 10 > $out/fwd.txt 2>&1; tail -3 $out/fwd.txt
python bench.py --steps 8 --warmup 2 --no-extra > $out/bench.txt 2>&1; tail -1 $out/bench.txt | cut -c1-400
