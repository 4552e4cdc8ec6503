#!/bin/bash
# A/B builds of one kernel file: tools/build_variant.sh <name> <file.hip> <extra hipcc flags...>
# -> gnn-lm_amd/build/exp/lib<name>.so (use with GNNLM_LIB=...)
set -e
cd "$(dirname "$0")/../gnn-lm_amd/csrc"
name=$1; file=$2; shift 2
mkdir -p ../build/exp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $file -o ../build/exp/$name.o
objs=""
for f in api gemm_f32 gemm_f32_dma gemm_f32_sched gemm_f32_skinny gemm_split gather_decode attn star_tab star_dense rowops groups knn_bucket topk ivfpq ivfpq_mfma; do
  if [ "$f.hip" == "$file" ]; then objs="$objs ../build/exp/$name.o"; else objs="$objs ../build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../build/exp/lib$name.so
echo ../build/exp/lib$name.so
