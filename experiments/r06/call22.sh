#!/bin/bash
# round 6, call 22: the windowed position-parallel encoder against the ring / split paths it replaces (variant noppw): containers of 8 KiB .. 1 MiB blocks, monolithic streams
mkdir -p gpurun_out/r06_c22
{
for v in default noppw; do
  if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi
  for B in 8192 16384 65536 1048576; do for kind in 0 1; do timeout 300 python tools/enc_time.py rle8_packed_multi $kind 8 $B; done; done
  for B in 8192 65536; do timeout 300 python tools/enc_time.py rle8_multi 0 8 $B; timeout 300 python tools/enc_time.py rle8_packed_multi 0 1 $B; done
  echo "== mono $v"; timeout 600 python tools/mono_enc_bench.py rle8_packed_multi,rle8_multi 1
  timeout 600 python tools/mono_enc_bench.py rle8_packed_multi 0.25
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c22/log.txt
