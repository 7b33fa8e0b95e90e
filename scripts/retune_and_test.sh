#!/bin/bash
# Re-run the tile tuner, then the whole GPU test suite and the smoke test with the new table.
mkdir -p gpurun_out
cp sliders_amd/tuning/gfx950_sdxl_128.json gpurun_out/gfx950_sdxl_128.json
timeout 900 python scripts/tune_gemm.py --out gpurun_out/gfx950_sdxl_128.json > gpurun_out/t17_tune.log 2>&1; tail -2 gpurun_out/t17_tune.log
cp gpurun_out/gfx950_sdxl_128.json sliders_amd/tuning/gfx950_sdxl_128.json
python scripts/bench_forward.py --lora --iters 10 > gpurun_out/t17_fwd_on.log 2>&1; tail -1 gpurun_out/t17_fwd_on.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t17_pytest_gpu.log 2>&1; tail -3 gpurun_out/t17_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/t17_smoke.log 2>&1; tail -2 gpurun_out/t17_smoke.log
