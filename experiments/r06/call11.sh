#!/bin/bash
# round 6, call 11: differential stress of the round's encoders against the oracle (every block stream, several block sizes, the drop-in functions too)
mkdir -p gpurun_out/r06_c11
STRESS_KEYS=rle8_single,rle8_packed_single,rle128,rle8_3symlut,rle8_7symlut,rle16_3symlut,rle16_7symlut,rle24_7symlut,rle32_7symlut,rle48_7symlut,rle64_7symlut timeout 700 python tools/gpu_stress.py 420 61 > gpurun_out/r06_c11/stress_new.log 2>&1; echo "stress new rc=$?"; tail -3 gpurun_out/r06_c11/stress_new.log
timeout 500 python tools/gpu_stress.py 300 62 > gpurun_out/r06_c11/stress_all.log 2>&1; echo "stress all rc=$?"; tail -3 gpurun_out/r06_c11/stress_all.log
