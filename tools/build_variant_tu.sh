#!/bin/bash
# tools/build_variant_tu.sh <translation unit, e.g. inst_pp8> <name> <extra hipcc flags...>  -> variants/libhsrle_<name>.so
# Quick A/B build for changes that only touch ONE translation unit: compiles csrc/<tu>.hip with the flags and links it with the
# objects of the default build (hypersonic-rle-kit_amd/build/*.o, `make` first).  Developer tool, nothing shipped depends on it.
set -eu
cd "$(dirname "$0")/../hypersonic-rle-kit_amd"
tu=$1; name=$2; shift 2
mkdir -p ../variants/build_"$name"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch="${HSRLE_ARCH:-gfx950}" "$@" -c csrc/"$tu".hip -o ../variants/build_"$name"/"$tu".o
objs=$(ls build/*.o | grep -v "/$tu.o")
/opt/rocm/bin/hipcc --offload-arch="${HSRLE_ARCH:-gfx950}" -shared -fPIC -o ../variants/libhsrle_"$name".so ../variants/build_"$name"/"$tu".o $objs -ldl
rm -rf ../variants/build_"$name"
