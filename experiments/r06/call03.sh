#!/bin/bash
# round 6, call 3: the position-parallel LUT encoder (7 symbol LUT, narrow 3 symbol LUT) -- parity, then its time on the 8 GiB buffers
mkdir -p gpurun_out/r06_c03
python -m pytest tests/test_gpu_pp.py -q -x -k "lut_general" > gpurun_out/r06_c03/pp_lut.log 2>&1; echo "pp_lut rc=$?"
tail -15 gpurun_out/r06_c03/pp_lut.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_big.py -q -x -k "symlut and not short" > gpurun_out/r06_c03/parity_lut.log 2>&1; echo "parity rc=$?"
tail -5 gpurun_out/r06_c03/parity_lut.log
for k in rle8_3symlut rle8_7symlut rle16_3symlut_byte rle16_7symlut_byte rle32_7symlut_sym rle64_7symlut_byte; do for kind in 0 1; do python tools/enc_time.py $k $kind 8; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c03/enc_time.log
