#!/bin/bash
# Builds a diagnostic variant of libtvmi_kernels.so: tools/build_variant.sh <name> <file.hip> <extra hipcc flags...>
# -> _variants/libtvmi_kernels_<name>.so (the other objects come from build/obj: run `make -C vision_amd/csrc` first).
# A GPU visit swaps it in on the box's scratch copy: cp _variants/libtvmi_kernels_<name>.so vision_amd/_lib/libtvmi_kernels.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p _variants
stem=$(basename "$src" .hip)
extra=""
[ "$stem" = deform_conv2d ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $extra \
  -Iinclude "$@" -c vision_amd/csrc/$src -o _variants/${stem}_$name.o
objs=$(ls build/obj/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/libtvmi_kernels_$name.so $objs _variants/${stem}_$name.o \
  -Wl,-soname,libtvmi_kernels.so
echo _variants/libtvmi_kernels_$name.so
