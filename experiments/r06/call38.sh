#!/bin/bash
# round 6, call 38: windowed general LUT encoder -- larger cases, the differential stress of every codec (block sizes above 4 KiB in the mix), speed on 8 GiB in 8 / 64 KiB blocks
mkdir -p gpurun_out/r06_c38
{
timeout 900 python tools/probe_ppws.py "rle8_3,rle8_7,rle16_3symlut,rle16_7,rle24_7symlut_byte,rle32_7symlut_sym,rle64_7,rle16_3symlut_sym_short,rle16_7symlut_byte_short,rle24_3symlut_byte_short,rle32_3symlut_sym_short,rle48_7symlut_sym_short" 4 2>&1 | tail -3
timeout 500 python tools/gpu_stress.py 360 71 2>&1 | tail -6
for k in rle8_3symlut rle8_7symlut rle16_3symlut_byte rle32_7symlut_byte rle64_7symlut_sym rle24_7symlut_byte_short; do for B in 8192 65536; do timeout 300 python tools/enc_time.py $k 0 8 $B; done; done
timeout 300 python tools/enc_time.py rle8_7symlut 1 8 65536; timeout 300 python tools/enc_time.py rle48_7symlut_byte 1 8 65536
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c38/log.txt
