#!/bin/bash
# round 5 call 44: decoder rounds capped at N packets per lane (rows as rings of 128 bytes, 64-byte halves flushed): 8 bit codecs, 4 GiB, default against N = 2, 3, 4, 6
cd /root/repo
K=rle8_multi,rle8_packed_multi,rle8_3symlut,rle8_7symlut,rle8_multi_short,rle8_7symlut_short
echo default; python tools/mini_sweep.py 4096 $K 2>&1 | grep -v "random\|amdgpu"
for n in 2 3 4 6; do echo trips$n; HSRLE_LIB=/root/repo/variants/libhsrle_trips$n.so python tools/mini_sweep.py 4096 $K 2>&1 | grep -v "random\|amdgpu"; done
