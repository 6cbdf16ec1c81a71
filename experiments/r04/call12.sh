#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4k
{ for v in fastonly default old; do if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi; timeout 200 python tools/enc_time.py rle8_packed_multi 2>&1 | grep -v amdgpu.ids | tail -1; done
  HSRLE_LIB=$PWD/variants/libhsrle_fastonlyst.so timeout 300 python tools/probe_enc_stamps.py rle8_packed_multi 0 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r4k/log.txt 2>&1
cat gpurun_out/r4k/log.txt
