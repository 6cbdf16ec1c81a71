#!/bin/bash
# round 6, call 17: differential stress of the final build (all codecs; then the Short codecs with lists and the 8 bit multi codecs, whose encoders changed last)
mkdir -p gpurun_out/r06_c17
timeout 500 python tools/gpu_stress.py 300 71 > gpurun_out/r06_c17/stress_all.log 2>&1; echo "stress all rc=$?"; tail -2 gpurun_out/r06_c17/stress_all.log
STRESS_KEYS=rle8_multi,rle8_packed,rle8_single,rle16_3symlut,rle16_7symlut,rle24_3symlut,rle24_7symlut,rle32_3symlut,rle32_7symlut,rle48_7symlut,rle64_7symlut timeout 500 python tools/gpu_stress.py 300 72 > gpurun_out/r06_c17/stress_new.log 2>&1; echo "stress new rc=$?"; tail -2 gpurun_out/r06_c17/stress_new.log
