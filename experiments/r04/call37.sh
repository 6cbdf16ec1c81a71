#!/bin/bash
# decoder stream ring 64 / 128 on run-distributed data for the codecs whose config-5 rows sit at 40 % (experiment build: HSRLE_DEC_RING forces)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
K=rle16_3symlut_byte,rle24_3symlut_byte,rle24_byte_packed,rle32_3symlut_byte,rle32_byte_packed,rle48_3symlut_byte,rle48_byte_packed,rle64_3symlut_byte,rle64_byte_packed,rle24_7symlut_byte,rle32_7symlut_sym
export HSRLE_LIB=variants/libhsrle_exp.so
for r in 128 64; do echo "== HSRLE_DEC_RING=$r"; env HSRLE_DEC_RING=$r timeout 900 python tools/ab_codecs.py 4096 $K 0 2>&1 | grep -v amdgpu.ids | awk '{print $2,$3,$5}'; done
