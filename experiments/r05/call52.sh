#!/bin/bash
# round 5 call 52: three ways on one box: rounds uncapped (round 4's kernel) / capped without / with the extra trip for lanes starved of literal bytes
cd /root/repo
K=rle8_single_short,rle8_multi_short,rle8_1symlut_short,rle8_3symlut_short,rle8_7symlut_short,rle16_sym,rle24_byte_short,rle32_sym_packed,rle48_7symlut_byte,rle64_byte,rle64_7symlut_byte_short_greedy
for v in uncapped nosp default; do
  if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=/root/repo/variants/libhsrle_$v.so; fi
  python tools/mini_sweep.py 4096 $K 2>&1 | grep -v "amdgpu" | sed "s/^/$v /"
done
