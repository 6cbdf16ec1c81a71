#!/bin/bash
# round 6, call 37: first run of the windowed general LUT encoder (7 symbol LUT, 3 symbol LUT of 1 / 2 byte symbols, Short with 3 / 7 symbol lists) against the oracle
mkdir -p gpurun_out/r06_c37
timeout 1700 python tools/probe_ppws.py "rle8_3,rle8_7,rle16_3symlut,rle16_7,rle24_7,rle32_7,rle48_7,rle64_7,rle16_3symlut_sym_short,rle24_3symlut_byte_short,rle32_3symlut_sym_short,rle32_7symlut_byte_short" 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c37/probe.log | grep -v " done, " | head -60
tail -3 gpurun_out/r06_c37/probe.log
