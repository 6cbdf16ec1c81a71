#!/bin/bash
# round 5 call 57: rounds that end when fewer than M lanes still have work (adaptive cap; ring rows, cap 64 = none): M = 16, 32 against uncapped / shipped caps
cd /root/repo
K=rle8_packed_multi,rle8_multi,rle8_3symlut,rle8_7symlut_short,rle8_multi_short,rle16_sym,rle24_byte_short,rle32_sym_packed,rle48_7symlut_byte,rle64_byte,rle64_7symlut_byte_short_greedy,rle128_sym
for v in uncapped default c64 c64m16 c64m32; do
  if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=/root/repo/variants/libhsrle_$v.so; fi
  python tools/mini_sweep.py 4096 $K 2>&1 | grep -v "amdgpu" | sed "s/^/$v /"
done
