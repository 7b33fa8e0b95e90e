cd /root/repo
mkdir -p gpurun_out/c44
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "64x160" > gpurun_out/c44/t_g5.txt 2>&1
for v in base q_g5 q_g5s5 base q_g5; do
  if [ $v == base ]; then unset SLIDERS_TUNING_OVERRIDE; else export SLIDERS_TUNING_OVERRIDE=/root/repo/scripts/tuning_ab/ovr_$v.json; fi
  timeout 300 python scripts/insitu_gemms.py --attn 2>&1 | grep -E "^ +2048 +1280 +1280|all ops|Tk77|GEMM in situ" >> gpurun_out/c44/insitu_$v.txt
  timeout 300 python bench.py --steps 6 --warmup 1 2>&1 | tail -1 | cut -c1-140 >> gpurun_out/c44/bench_$v.txt
done
