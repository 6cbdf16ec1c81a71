#!/bin/bash
# round 6, call 25: windowed encoder -- mono flow without the read-back in the middle (parity + time), and where it overtakes split / ring on small containers
mkdir -p gpurun_out/r06_c25
{
timeout 600 python tools/probe_ppw.py mono | tail -3
timeout 300 python tools/mono_enc_bench.py rle8_packed_multi,rle8_multi 1
for g in 0.25 0.0824 0.015625; do timeout 300 python tools/mono_enc_bench.py rle8_packed_multi $g; done
for v in ppwall noppw; do HSRLE_LIB=$PWD/variants/libhsrle_$v.so timeout 600 python tools/ppw_threshold.py; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c25/log.txt
