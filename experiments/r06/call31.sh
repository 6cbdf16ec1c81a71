#!/bin/bash
# round 6, call 31: windowed encoders of the wide codecs on small containers, against the split / ring paths (variant noppw)
mkdir -p gpurun_out/r06_c31
for k in rle64_3symlut_byte rle16_sym_packed; do for v in default noppw; do
  if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi
  timeout 600 python tools/ppw_threshold.py $k
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c31/log.txt
