#!/bin/bash
# round 5 call 53: how many lanes starved of literal bytes make a capped round go on without a flush?  1 / 16 / 32 (default) / 48, against uncapped
cd /root/repo
K=rle8_single_short,rle8_multi_short,rle8_3symlut_short,rle8_7symlut_short,rle16_sym,rle24_byte_short,rle32_sym_packed,rle48_7symlut_byte,rle64_byte
for v in uncapped sm1 sm16 default sm48; do
  if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=/root/repo/variants/libhsrle_$v.so; fi
  python tools/mini_sweep.py 4096 $K 2>&1 | grep -v "amdgpu" | sed "s/^/$v /"
done
