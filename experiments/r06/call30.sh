#!/bin/bash
# round 6, call 30: windowed encoders of the 1 .. 8 byte symbol codecs -- larger cases, the differential stress (block sizes above 4 KiB in the mix), and their speed on 8 GiB
mkdir -p gpurun_out/r06_c30
{
timeout 900 python tools/probe_ppws.py "rle16,rle24_3,rle32_byte,rle48_sym_packed,rle64,rle8_multi_short,rle8_1" 4 2>&1 | tail -4
STRESS_KEYS=rle16,rle24,rle32,rle48,rle64,rle8_multi_short,rle8_1symlut_short timeout 500 python tools/gpu_stress.py 300 59 2>&1 | tail -6
for k in rle16_sym_packed rle32_byte rle64_3symlut_byte rle24_sym rle48_byte_packed rle32_sym_short; do for B in 8192 65536; do timeout 300 python tools/enc_time.py $k 0 8 $B; done; done
timeout 300 python tools/enc_time.py rle64_3symlut_byte 1 8 65536
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c30/log.txt
