#!/bin/bash
# A/B build of libpatchioner_hip.so with extra -D flags:  tools/microbench/build_variant.sh NAME -DPIO_X=1 ...
# -> tools/microbench/bin/libpio_NAME.so   (run with PIO_LIB_PATH=$PWD/tools/microbench/bin/libpio_NAME.so)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
out=tools/microbench/bin/var_$name; mkdir -p $out
for s in api.cpp vit_gemm.hip vit_gemm256.hip vit_gemm_roll.hip vit_attention.hip vit_fp32.hip vit_misc.hip region.hip project.hip decoder.hip viecap.hip preprocess.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -w -mllvm -amdgpu-kernarg-preload-count=16 -I include -I patchioner_amd/csrc "$@" -c patchioner_amd/csrc/$s -o $out/${s%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/microbench/bin/libpio_$name.so $out/*.o
echo tools/microbench/bin/libpio_$name.so
