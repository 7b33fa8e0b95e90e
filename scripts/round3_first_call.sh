#!/bin/bash
# First GPU call of round 3 (gpurun --timeout 900 -- 'bash scripts/round3_first_call.sh'): everything that was written or
# changed after round 2's GPU budget ran out, in the order of how much depends on it.  Logs -> gpurun_out/r03_first/.
#  1. the default tree: full GPU suite (serial, like the driver) + smoke
#  2. the code that has never run on hardware: tensor-op schedulers and Prodigy inside SliderTrainer / SliderSampler
#  3. the run-to-run floor of an iteration after the LDS-DMA wait fix -> tighten the 8 % / 0.97 bounds
#  4. batched text K/V in the TRAINING forward (root cause fixed): suite x2 + the reproducer; then flip the default in
#     planner.py (SLIDERS_TRAIN_KV_BATCHED) if all green
#  5. bench lines, default and with (4)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_first
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q > $O/1_suite_serial.log 2>&1; tail -2 $O/1_suite_serial.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/1_smoke.log 2>&1; tail -1 $O/1_smoke.log
SLIDERS_RUN_UNVALIDATED=1 timeout 300 python -m pytest tests/test_schedulers_gpu.py -q -s > $O/2_schedulers.log 2>&1; grep -E "parity|passed|failed" $O/2_schedulers.log
timeout 300 python scripts/smoke_unvalidated.py > $O/2_smoke_unvalidated.log 2>&1; tail -9 $O/2_smoke_unvalidated.log
for m in tiny_sdxl sd1; do timeout 120 python scripts/determinism_probe.py --model $m --hw 32 > $O/3_determinism_$m.log 2>&1; head -4 $O/3_determinism_$m.log; done
for m in tiny_sdxl tiny_sd1; do timeout 200 python scripts/noise_floor.py --model $m > $O/3_noise_$m.log 2>&1; tail -1 $O/3_noise_$m.log; done
timeout 300 python scripts/noise_floor.py --model sdxl --hw 32 --n 3 > $O/3_noise_sdxl32.log 2>&1; tail -1 $O/3_noise_sdxl32.log
for i in 1 2; do SLIDERS_TRAIN_KV_BATCHED=1 timeout 400 python -m pytest tests -m gpu -q -n 6 -p no:cacheprovider > $O/4_suite_kvb_$i.log 2>&1; tail -1 $O/4_suite_kvb_$i.log; done
SLIDERS_TRAIN_KV_BATCHED=1 NAN_TRIALS=16 timeout 200 python scripts/debug_nan_forward.py train > $O/4_repro.log 2>&1; tail -1 $O/4_repro.log
timeout 300 python bench.py > $O/5_bench.json 2> $O/5_bench.err; cut -c1-200 $O/5_bench.json
SLIDERS_TRAIN_KV_BATCHED=1 timeout 300 python bench.py --no-cpu-baseline > $O/5_bench_kvb.json 2> $O/5_bench_kvb.err; cut -c1-200 $O/5_bench_kvb.json
