#!/bin/bash
# One libawt variant that differs from the shipped build in ONE translation unit:  tools/build_variant.sh <name> <unit> "<flags>" [source root]
#   -> mlx8-ws-audio-transformer_amd/variants/libawt_<name>.so   (<unit> e.g. gemm, attention_f8; source root defaults to this tree, a `git archive` copy gives an older unit)
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; flags=$3; src=${4:-$PWD}
P=mlx8-ws-audio-transformer_amd; mkdir -p $P/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc -Wno-unused-function $flags -c $src/$P/csrc/$unit.hip -o $P/variants/${unit}_$name.o
objs=$(ls $P/csrc/_obj/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/variants/libawt_$name.so $objs $P/variants/${unit}_$name.o -ldl
echo "built $P/variants/libawt_$name.so  ($unit: $flags)"
