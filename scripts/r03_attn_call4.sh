#!/bin/bash
out=gpurun_out/r03_attn; mkdir -p $out
L=$PWD/sliders_amd
for v in old fast fast_nopk g1 g1_nopk g2 g2_nopk; do
  lib=$L/libsliders_hip_$v.so; [ $v == fast ] && lib=$L/libsliders_hip.so
  SLIDERS_HIP_LIB=$lib timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" -x 2>&1 | tail -1 > $out/t_$v.txt
  SLIDERS_HIP_LIB=$lib python scripts/probe_attn.py > $out/p_$v.txt 2>&1
  SLIDERS_HIP_LIB=$lib python scripts/probe_attn_occ.py > $out/o_$v.txt 2>&1
done
SLIDERS_HIP_LIB=$L/libsliders_hip_trace.so python scripts/probe_attn_place.py > $out/phases_fast.txt 2>&1
SLIDERS_HIP_LIB=$L/libsliders_hip_g2_trace.so python scripts/probe_attn_place.py > $out/phases_g2.txt 2>&1
for v in old fast fast_nopk g1 g1_nopk g2 g2_nopk; do echo "== $v $(cat $out/t_$v.txt)"; grep -v amdgpu $out/p_$v.txt | awk '{printf "%s ", $(NF-3)} END {print ""}'; grep "^T" $out/o_$v.txt | awk '{printf "%s ", $7} END {print ""}'; done
grep -h "cycles per tile" $out/phases_fast.txt $out/phases_g2.txt | sed 's/CUs used.*wave time/wave time/'
