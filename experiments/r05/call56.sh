#!/bin/bash
# round 5 call 56: with capped rounds a lane uses less of its ring per round -- does the 64-byte ring (12 instead of 9 waves per CU) now pay at higher ratios?  forced 64 / 128 against the library's choice
cd /root/repo
export HSRLE_LIB=/root/repo/variants/libhsrle_exp.so
K=rle16_sym,rle16_7symlut_sym_short,rle24_byte_short,rle24_7symlut_byte,rle32_sym_packed,rle32_3symlut_byte,rle48_7symlut_byte,rle48_byte_packed,rle64_byte,rle64_3symlut_byte,rle64_7symlut_byte_short
for v in 0 64 128; do
  HSRLE_DEC_RING=$v python tools/mini_sweep.py 4096 $K 2>&1 | grep -v "amdgpu\|random" | sed "s/^/ring$v /"
done
