#!/bin/bash
# round 6, call 2: the position-parallel 128 bit encoder -- parity, then its time on the 8 GiB buffers
mkdir -p gpurun_out/r06_c02
python -m pytest tests/test_gpu_pp.py -q -x -k "128" > gpurun_out/r06_c02/pp_128.log 2>&1; echo "pp_128 rc=$?"
tail -15 gpurun_out/r06_c02/pp_128.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_big.py -q -x -k "rle128" > gpurun_out/r06_c02/parity_128.log 2>&1; echo "parity rc=$?"
tail -5 gpurun_out/r06_c02/parity_128.log
for k in rle128_sym rle128_sym_packed rle128_byte rle128_byte_packed; do for kind in 0 1; do python tools/enc_time.py $k $kind 8; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c02/enc_time.log
