#!/bin/bash
# round 6, call 34: windowed emission with its state, records and input requested together (block mode: block and window follow from the window number) -- against call 30 / 33
mkdir -p gpurun_out/r06_c34
{
for rep in 1 2; do
for k in rle8_packed_multi rle16_sym_packed rle32_byte rle64_3symlut_byte; do for B in 8192 65536; do timeout 300 python tools/enc_time.py $k 0 8 $B; done; done
done
timeout 300 python tools/enc_time.py rle8_packed_multi 1 8 65536
timeout 600 python tools/probe_ppw.py blocks | tail -2
timeout 600 python tools/probe_ppws.py "rle16_sym,rle24_3symlut_byte,rle32_byte_packed,rle64_3symlut_byte_short,rle8_multi_short" 1 | tail -2
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c34/log.txt
