#!/bin/bash
# round 5 call 48: capped decoder rounds, N = 3, 4, 5, 6, 8 against the shipped kernels (wide symbols + 8 bit Short): 4 GiB
cd /root/repo
K=rle8_7symlut_short,rle8_3symlut,rle16_sym,rle16_7symlut_sym_short,rle16_3symlut_byte_short_greedy,rle24_byte_short,rle24_7symlut_byte,rle32_sym_packed,rle32_7symlut_sym_short,rle48_7symlut_byte,rle48_7symlut_byte_short_greedy,rle64_byte,rle64_3symlut_byte,rle64_7symlut_byte_short_greedy,rle128_sym_packed
for v in default cap3 cap4 cap5 cap6 cap8; do
  if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=/root/repo/variants/libhsrle_$v.so; fi
  python tools/mini_sweep.py 4096 $K 2>&1 | grep -v "random\|amdgpu" | sed "s/^/$v /"
done
