#!/bin/bash
# round 5 call 47: capped decoder rounds for every symbol width (N = 6, 10) against the shipped kernels: 4 GiB, run-distributed and video-shaped
cd /root/repo
K=rle8_packed_multi,rle8_7symlut_short,rle16_sym,rle16_7symlut_sym_short,rle16_3symlut_byte_short_greedy,rle24_byte_short,rle24_7symlut_byte,rle32_sym_packed,rle32_3symlut_byte,rle32_7symlut_sym_short,rle48_7symlut_byte,rle48_7symlut_byte_short_greedy,rle48_3symlut_byte_short,rle64_byte,rle64_3symlut_byte,rle64_7symlut_byte_short_greedy,rle128_sym
for v in default cap6 cap10; do
  if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=/root/repo/variants/libhsrle_$v.so; fi
  python tools/mini_sweep.py 4096 $K 2>&1 | grep -v "random\|amdgpu" | sed "s/^/$v /"
done
