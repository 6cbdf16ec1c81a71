#!/bin/bash
# tools/build_variant_all.sh <name> <extra hipcc flags...>  -> variants/libhsrle_<name>.so
# A/B build for changes in a header that every decoder / encoder instantiation unit includes: compiles all csrc/inst_w*.hip with the flags (in parallel) and
# links them with the remaining objects of the default build (`make` first).  Developer tool, nothing shipped depends on it.
set -eu
cd "$(dirname "$0")/../hypersonic-rle-kit_amd"
name=$1; shift
B=../variants/build_"$name"; mkdir -p "$B"
for tu in inst_w8 inst_w16 inst_w24 inst_w32 inst_w48 inst_w64 inst_w128; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch="${HSRLE_ARCH:-gfx950}" "$@" -c csrc/$tu.hip -o "$B"/$tu.o &
done
wait
objs=$(ls build/*.o | grep -v "/inst_w")
/opt/rocm/bin/hipcc --offload-arch="${HSRLE_ARCH:-gfx950}" -shared -fPIC -o ../variants/libhsrle_"$name".so "$B"/*.o $objs -ldl
rm -rf "$B"
