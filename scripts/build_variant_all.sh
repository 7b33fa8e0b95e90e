#!/bin/bash
# Development aid: builds libsliders_hip_<name>.so with extra compiler flags for EVERY translation unit (A/B of a switch that lives in
# a shared header; SLIDERS_HIP_LIB=<path> selects the library).  usage: build_variant_all.sh <name> <flags...>
set -e
name=$1; shift 1
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/sliders_amd/csrc
mkdir -p $src/build/var_$name
objs=""
for u in gemm gemm8p gemm5 lora norm attention attention_bwd ops vae; do
  extra=""; [ $u == attention ] && extra="-fno-slp-vectorize"; [ $u == gemm5 ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $extra "$@" -c $src/$u.hip -o $src/build/var_$name/$u.o &
  objs="$objs $src/build/var_$name/$u.o"
done
for u in program error; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -x hip "$@" -c $src/$u.cpp -o $src/build/var_$name/$u.o &
  objs="$objs $src/build/var_$name/$u.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/sliders_amd/libsliders_hip_$name.so $objs
echo built sliders_amd/libsliders_hip_$name.so
