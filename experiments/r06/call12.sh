#!/bin/bash
# round 6, call 12: the Short codecs with a 3 / 7 symbol list on the position-parallel LUT kernel -- parity, then time on the 8 GiB buffers
mkdir -p gpurun_out/r06_c12
python -m pytest tests/test_gpu_pp.py -q -x -k "short_list or lut_general" > gpurun_out/r06_c12/pp_shortl.log 2>&1; echo "pp rc=$?"; tail -6 gpurun_out/r06_c12/pp_shortl.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_big.py -q -x -k "symlut" > gpurun_out/r06_c12/parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r06_c12/parity.log
for k in rle8_3symlut rle8_7symlut rle16_7symlut_byte rle16_7symlut_byte_short rle32_3symlut_byte_short rle32_7symlut_sym_short rle64_7symlut_byte_short; do for kind in 0 1; do python tools/enc_time.py $k $kind 8; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c12/enc_time.log
