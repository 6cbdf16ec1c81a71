#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
K=rle32_3symlut_sym,rle32_sym_packed,rle24_3symlut_sym,rle32_7symlut_sym
HSRLE_LIB=$PWD/variants/libhsrle_old.so timeout 600 python tools/ab_codecs.py 8192 $K 1 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/ab_codecs.py 8192 $K 1 2>&1 | grep -v amdgpu.ids
