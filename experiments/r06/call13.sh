#!/bin/bash
# round 6, call 13: where the 8 bit Short-with-list encodes spend their time (kernel stats)
for spec in "rle8_7symlut_short 0" "rle8_3symlut_short 1" "rle8_7symlut 0"; do
  set -- $spec
  echo "== $1 kind $2"
  ROWS=3 bash tools/prof_script.sh r06_c13_$1_$2 tools/enc_time.py $1 $2 8 2>&1 | grep -v amdgpu.ids | tail -4
done
