#!/bin/bash
out=gpurun_out/r03_attn; mkdir -p $out
timeout 120 scripts/ubench_mfma_valu > $out/ubench_mfma_valu.txt 2>&1
SLIDERS_HIP_LIB=$PWD/sliders_amd/libsliders_hip_trace.so python scripts/probe_attn_place.py > $out/phases.txt 2>&1
cat $out/ubench_mfma_valu.txt; grep -v amdgpu.ids $out/phases.txt | sed 's/CUs used.*wave time/wave time/'
