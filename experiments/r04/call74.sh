#!/bin/bash
# phase stamps of the headline encoder with 64 and 128 byte windows
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for v in st64 st128; do echo "== $v"; HSRLE_LIB=$PWD/variants/libhsrle_$v.so timeout 300 python tools/probe_enc_stamps.py rle8_packed_multi 0 2>&1 | grep -v amdgpu.ids; done
