cd /root/repo
mkdir -p gpurun_out/c53
timeout 900 python -m pytest tests/test_trainer_gpu.py tests/test_graph_gpu.py tests/test_seam_gpu.py tests/test_dp_sim_gpu.py tests/test_rccl_gpu.py -x -q > gpurun_out/c53/tests.txt 2>&1
for i in 1 2 3; do timeout 300 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-130 >> gpurun_out/c53/bench.txt; done
