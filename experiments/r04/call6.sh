#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4f
{
timeout 600 python tools/mini_sweep.py 8192 rle8_packed_multi,rle8_3symlut,rle16_sym,rle16_byte_packed,rle16_3symlut_byte,rle16_7symlut_sym,rle24_3symlut_byte,rle32_byte_packed,rle32_3symlut_sym,rle48_7symlut_byte,rle64_3symlut_byte,rle64_sym,rle16_sym_short,rle32_3symlut_byte_short,rle64_7symlut_sym_short 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r4f/sweep.txt 2>&1
cat gpurun_out/r4f/sweep.txt
( time timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r4f/tests.txt
