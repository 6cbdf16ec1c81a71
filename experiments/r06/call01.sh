#!/bin/bash
# round 6, call 1: the position-parallel Single encoder -- parity (new cases + every existing test that names a Single codec), then its time on the 8 GiB buffers
mkdir -p gpurun_out/r06_c01
python -m pytest tests/test_gpu_pp.py -q -x -k "single" > gpurun_out/r06_c01/pp_single.log 2>&1; echo "pp_single rc=$?"
tail -5 gpurun_out/r06_c01/pp_single.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_big.py -q -x -k "single and not short" > gpurun_out/r06_c01/parity_single.log 2>&1; echo "parity rc=$?"
tail -5 gpurun_out/r06_c01/parity_single.log
for k in rle8_single rle8_packed_single; do for kind in 0 1; do python tools/enc_time.py $k $kind 8; done; done 2>&1 | tee gpurun_out/r06_c01/enc_time.log
