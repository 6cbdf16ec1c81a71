#!/bin/bash
# round 5 call 59: capped rounds that flush only when a row is about to run out of room (unflushed >= T bytes) or is through, or >= 32 rows have a complete half
cd /root/repo
K=rle8_7symlut_short,rle16_sym,rle16_7symlut_sym_short,rle24_byte_short,rle32_sym_packed,rle48_7symlut_byte,rle64_byte,rle64_7symlut_byte_short_greedy
for v in default lazy80 lazy96 lazy112; do
  if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=/root/repo/variants/libhsrle_$v.so; fi
  python tools/mini_sweep.py 4096 $K 2>&1 | grep -v "amdgpu" | sed "s/^/$v /"
done
