#!/bin/bash
# usage: r04_variants.sh <outdir> <shapes> <convs> <tiles> name:check ...   (name "base" = the default library)
out=gpurun_out/$1; shapes=$2; convs=$3; tiles=$4; shift 4
mkdir -p $out
cd /root/repo
for nc in "$@"; do
  v=${nc%%:*}; chk=${nc##*:}
  if [ $v == base ]; then unset SLIDERS_HIP_LIB; else export SLIDERS_HIP_LIB=/root/repo/sliders_amd/libsliders_hip_$v.so; fi
  echo "== $v" >> $out/variants.log
  timeout 300 python scripts/probe_gemm8p.py --shapes "$shapes" --convs "$convs" --tiles $tiles --check $chk 2>&1 | grep -v amdgpu.ids >> $out/variants.log
done
cat $out/variants.log
