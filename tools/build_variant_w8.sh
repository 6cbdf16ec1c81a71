#!/bin/bash
# tools/build_variant_w8.sh <name> <extra hipcc flags...>  -> variants/libhsrle_<name>.so
# Quick A/B build for changes that only touch the 8 bit translation unit: compiles csrc/inst_w8.hip with the flags and links it with the
# objects of the default build (hypersonic-rle-kit_amd/build/*.o, `make` first).  Developer tool, nothing shipped depends on it.
set -eu
cd "$(dirname "$0")/../hypersonic-rle-kit_amd"
name=$1; shift
mkdir -p ../variants/build_"$name"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch="${HSRLE_ARCH:-gfx950}" "$@" -c csrc/inst_w8.hip -o ../variants/build_"$name"/inst_w8.o
objs=$(ls build/*.o | grep -v inst_w8.o)
/opt/rocm/bin/hipcc --offload-arch="${HSRLE_ARCH:-gfx950}" -shared -fPIC -o ../variants/libhsrle_"$name".so ../variants/build_"$name"/inst_w8.o $objs -ldl
rm -rf ../variants/build_"$name"
