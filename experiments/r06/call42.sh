#!/bin/bash
# round 6, call 42: windowed emission with the second image piece 16-byte aligned in LDS; codec-sized workspaces without split regions for the windowed codecs
mkdir -p gpurun_out/r06_c42
{
timeout 600 python tools/probe_ppw.py all | tail -2
timeout 900 python tools/probe_ppws.py "rle16_sym,rle24_3symlut_byte,rle32_byte_packed,rle64_3symlut_byte_short,rle8_multi_short,rle8_7symlut,rle16_3symlut_sym,rle32_7symlut_byte,rle24_7symlut_sym_short" 1 | tail -2
for rep in 1 2; do for k in rle8_packed_multi rle16_sym_packed rle32_byte rle64_3symlut_byte rle32_7symlut_byte; do timeout 300 python tools/enc_time.py $k 0 8 65536; done; done
timeout 300 python tools/enc_time.py rle8_packed_multi 1 8 65536
timeout 300 python tools/mono_enc_bench.py rle8_packed_multi 1
timeout 300 python tools/ppw_threshold.py rle8_packed_multi | awk '{print $3,$4,$5,$6,$9,$10,$11,$12,$13,$14,$15}' | head -8
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c42/log.txt
