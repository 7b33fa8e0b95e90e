#!/bin/bash
# Round 3, GPU call 2: the deterministic engine (fixed-order GroupNorm / split-K reductions).  Logs -> gpurun_out/r03_c2/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c2
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q -n 6 -p no:cacheprovider -rf > $O/1_suite.log 2>&1; tail -25 $O/1_suite.log | grep -E "passed|failed|FAILED|Error" | head -30
for m in tiny_sdxl sd1; do timeout 120 python scripts/determinism_probe.py --model $m --hw 32 > $O/2_determinism_$m.log 2>&1; grep -v amdgpu.ids $O/2_determinism_$m.log | head -4; done
timeout 200 python scripts/determinism_probe.py --model sdxl --hw 128 > $O/2_determinism_sdxl128.log 2>&1; grep -v amdgpu.ids $O/2_determinism_sdxl128.log | head -4
for m in tiny_sdxl tiny_sd1; do timeout 200 python scripts/noise_floor.py --model $m > $O/3_noise_$m.log 2>&1; tail -1 $O/3_noise_$m.log; done
timeout 300 python scripts/noise_floor.py --model sdxl --hw 32 --n 3 > $O/3_noise_sdxl32.log 2>&1; tail -1 $O/3_noise_sdxl32.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fwd -o fwd -- python $R/scripts/bench_forward.py --lora --warm 1 --iters 3 > $O/4_prof_fwd.log 2>&1
find /tmp/prof_fwd -name "*kernel_stats.csv" -exec cp {} $O/4_fwd_lora_on_kernel_stats.csv \;
head -30 $O/4_fwd_lora_on_kernel_stats.csv | cut -c1-150
cd $R
timeout 300 python bench.py --no-cpu-baseline > $O/5_bench.json 2> $O/5_bench.err; cut -c1-200 $O/5_bench.json
