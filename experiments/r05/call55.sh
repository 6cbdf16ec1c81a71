#!/bin/bash
# round 5 call 55: Greedy block kernel with 8 (6) bytes per in-run trip: parity (every test that has a Greedy codec in its name), then encode time at 2 GiB
cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu -k "greedy" 2>&1 | tail -3
for k in rle16_3symlut_byte_short_greedy rle24_7symlut_byte_short_greedy rle32_1symlut_byte_short_greedy rle48_3symlut_byte_short_greedy; do for kind in 0 1; do python tools/enc_time.py $k $kind 2 2>&1 | grep -v amdgpu; done; done
