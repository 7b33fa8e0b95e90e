#!/bin/bash
# Development aid: builds libsliders_hip_<name>.so with extra compiler flags for ONE translation unit (A/B of kernel variants
# in one GPU call; SLIDERS_HIP_LIB=<path> selects it).  usage: build_variant.sh <name> <file.hip> <flags...>
set -e
name=$1; unit=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/sliders_amd/csrc
make -C $src -s
mkdir -p $src/build/var_$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $src/$unit -o $src/build/var_$name/${unit%.hip}.o
objs=""
for o in gemm gemm8p gemm5 gemm7 lora norm attention attention_bwd ops vae program error; do
  if [ "$o.hip" == "$unit" ]; then objs="$objs $src/build/var_$name/$o.o"; else objs="$objs $src/build/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/sliders_amd/libsliders_hip_$name.so $objs
echo built sliders_amd/libsliders_hip_$name.so
