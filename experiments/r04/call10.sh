#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4j
{ HSRLE_LIB=$PWD/variants/libhsrle_encst.so timeout 300 python tools/probe_enc_stamps.py rle8_packed_multi 0 2>&1 | grep -v amdgpu.ids
  HSRLE_LIB=$PWD/variants/libhsrle_encst.so timeout 300 python tools/probe_enc_stamps.py rle8_packed_multi 1 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r4j/log.txt 2>&1
cat gpurun_out/r4j/log.txt
