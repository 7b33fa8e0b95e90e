#!/bin/bash
# one GPU call: occupancy sweep of the attention forward + A/B of the loop variants (scripts/build_variant.sh)
mkdir -p gpurun_out/r03_attn
out=gpurun_out/r03_attn
python scripts/probe_attn_occ.py > $out/occ_default.txt 2>&1
for v in nopk summfma nopk_summfma prio; do
  SLIDERS_HIP_LIB=$PWD/sliders_amd/libsliders_hip_$v.so python scripts/probe_attn_occ.py > $out/occ_$v.txt 2>&1
done
python scripts/probe_attn.py > $out/probe_default.txt 2>&1
for v in nopk summfma nopk_summfma prio; do
  SLIDERS_HIP_LIB=$PWD/sliders_amd/libsliders_hip_$v.so python scripts/probe_attn.py > $out/probe_$v.txt 2>&1
  SLIDERS_HIP_LIB=$PWD/sliders_amd/libsliders_hip_$v.so timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" -x > $out/test_$v.txt 2>&1
done
tail -n 3 $out/test_*.txt
cat $out/occ_default.txt
for v in nopk summfma nopk_summfma prio; do echo; paste <(cut -c1-60 $out/occ_default.txt) <(awk '{print $(NF-11), $(NF-10)}' $out/occ_$v.txt) | head -3; done
grep -h "^sum" $out/probe_*.txt
