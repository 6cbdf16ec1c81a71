#!/bin/bash
# round 6, call 8: rocprofv3 kernel stats of the new position-parallel encoders (Single, 128 bit, LUT) on the 8 GiB buffers
for spec in "rle8_single 0" "rle8_packed_single 1" "rle8_single_short 0" "rle128_sym_packed 0" "rle128_byte 1" "rle8_7symlut 0" "rle8_7symlut 1" "rle16_3symlut_byte 0" "rle64_7symlut_byte 1" "rle32_7symlut_sym_short 1" "rle8_packed_multi 0"; do
  set -- $spec
  echo "== $1 kind $2"
  ROWS=6 bash tools/prof_script.sh r06_c08_$1_$2 tools/enc_time.py $1 $2 8 2>&1 | grep -v amdgpu.ids | tail -8
done
