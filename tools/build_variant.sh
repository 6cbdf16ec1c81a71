#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  -> variants/libhsrle_<name>.so  (developer A/B builds)
set -e
cd "$(dirname "$0")/../hypersonic-rle-kit_amd"
name=$1; shift
id=$(cat $(ls csrc/*.h csrc/*.hip csrc/*.inc | LC_ALL=C sort) ../include/hsrle.h | (sha256sum; echo "$@") | sha256sum | cut -c1-16)
mkdir -p ../variants/build_$name
for f in hsrle_capi hsrle_rccl inst_w8 inst_w16 inst_w24 inst_w32 inst_w48 inst_w64 inst_w128; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=${HSRLE_ARCH:-gfx950} "$@" -DHSRLE_BUILD_ID=\"$id\" -c csrc/$f.hip -o ../variants/build_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=${HSRLE_ARCH:-gfx950} -shared -fPIC -o ../variants/libhsrle_$name.so ../variants/build_$name/*.o -ldl
rm -rf ../variants/build_$name
