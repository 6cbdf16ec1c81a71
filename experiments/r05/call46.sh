#!/bin/bash
# round 5 call 46: what does the Greedy block kernel spend its time on?  the scan without its output stores (timing only) against the shipped kernel, 2 GiB
cd /root/repo
for k in rle16_3symlut_byte_short_greedy rle16_7symlut_byte_short_greedy; do for kind in 0 1; do
python tools/enc_time.py $k $kind 2 2>&1 | grep -v amdgpu; HSRLE_LIB=/root/repo/variants/libhsrle_gdry16.so python tools/enc_time.py $k $kind 2 2>&1 | grep -v amdgpu; done; done
for k in rle64_3symlut_byte_short_greedy rle64_7symlut_byte_short_greedy; do for kind in 0 1; do
python tools/enc_time.py $k $kind 2 2>&1 | grep -v amdgpu; HSRLE_LIB=/root/repo/variants/libhsrle_gdry64.so python tools/enc_time.py $k $kind 2 2>&1 | grep -v amdgpu; done; done
