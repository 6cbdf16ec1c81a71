#!/bin/bash
# round 6, call 16: encoders after "symbols from LDS in pass 1" (8 bit multi) and "the stream's offset asked for at the start" (all position-parallel kernels)
mkdir -p gpurun_out/r06_c16
SKIP_DEFAULT= REPS=2 bash tools/ab.sh 2>&1 | tee gpurun_out/r06_c16/ab.log
for k in rle8_multi rle8_packed_multi rle8_single rle16_sym_packed rle32_byte rle64_3symlut_byte rle128_sym_packed rle8_7symlut rle32_7symlut_sym; do for kind in 0 1; do python tools/enc_time.py $k $kind 8; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c16/enc_time.log
