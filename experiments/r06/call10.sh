#!/bin/bash
# round 6, call 10: rle8_single_short on the position-parallel Single kernel -- parity, time
mkdir -p gpurun_out/r06_c10
python -m pytest tests/test_gpu_pp.py -q -x -k "single" > gpurun_out/r06_c10/pp_single.log 2>&1; echo "pp_single rc=$?"; tail -4 gpurun_out/r06_c10/pp_single.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_big.py tests/test_gpu_split.py -q -x -k "single" > gpurun_out/r06_c10/parity_single.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r06_c10/parity_single.log
for kind in 0 1; do python tools/enc_time.py rle8_single_short $kind 8; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c10/enc_time.log
