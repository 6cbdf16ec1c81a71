#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4c
{
echo "== stamps default"; HSRLE_LIB=$PWD/variants/libhsrle_st.so timeout 200 python tools/probe_kernel_stamps.py 8192 2>&1 | grep -v amdgpu.ids | tail -6
echo "== stamps mw3"; HSRLE_LIB=$PWD/variants/libhsrle_mw3st.so timeout 200 python tools/probe_kernel_stamps.py 8192 2>&1 | grep -v amdgpu.ids | tail -6
} > gpurun_out/r4c/log.txt 2>&1
cat gpurun_out/r4c/log.txt
