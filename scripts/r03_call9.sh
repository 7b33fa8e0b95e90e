#!/bin/bash
# Round 3, GPU call 9: backward with batched launches + U fused into the backward-data GEMMs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c9
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "batched or backward_data_form or lora_wgrad" > $O/0_new.log 2>&1; tail -3 $O/0_new.log
timeout 600 python -m pytest tests -m gpu -q -n 6 -p no:cacheprovider -rf > $O/1_suite.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/1_suite.log | head -30
timeout 300 python scripts/time_train_iter.py --breakdown > $O/2_pieces_new.log 2>&1; grep -v amdgpu.ids $O/2_pieces_new.log | head -24
SLIDERS_BWD_UNBATCHED=1 SLIDERS_BWD_UNFUSED_U=1 timeout 300 python scripts/time_train_iter.py > $O/2_pieces_old.log 2>&1; grep -v amdgpu.ids $O/2_pieces_old.log | head -6
timeout 300 python bench.py --workload image --no-cpu-baseline --no-roofline --steps 10 > $O/3_image_new.json 2> $O/3_image_new.err; cut -c1-160 $O/3_image_new.json
SLIDERS_BWD_UNBATCHED=1 SLIDERS_BWD_UNFUSED_U=1 timeout 300 python bench.py --workload image --no-cpu-baseline --no-roofline --steps 10 > $O/3_image_old.json 2> $O/3_image_old.err; cut -c1-160 $O/3_image_old.json
