#!/bin/bash
out=gpurun_out/r03_attn; mkdir -p $out
T=$PWD/sliders_amd/libsliders_hip_trace.so
{
SLIDERS_HIP_LIB=$T python scripts/probe_attn_place.py
SLIDERS_HIP_LIB=$T SLH_ATTN_SPREAD=1 python scripts/probe_attn_place.py
SLIDERS_HIP_LIB=$T SLH_ATTN_NW2=1 python scripts/probe_attn_place.py
SLIDERS_HIP_LIB=$T SLH_ATTN_NW2=1 SLH_ATTN_SPREAD=1 python scripts/probe_attn_place.py
python scripts/probe_attn_place.py
SLH_ATTN_SPREAD=1 python scripts/probe_attn_place.py
SLH_ATTN_NW2=1 python scripts/probe_attn_place.py
SLH_ATTN_NW2=1 SLH_ATTN_SPREAD=1 python scripts/probe_attn_place.py
SLH_ATTN_SPREAD=1 python scripts/probe_attn.py
SLH_ATTN_NW2=1 SLH_ATTN_SPREAD=1 python scripts/probe_attn.py
} > $out/place.txt 2>&1
grep -v amdgpu.ids $out/place.txt
