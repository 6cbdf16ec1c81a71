#!/bin/bash
# round 6, call 6: low-entropy encoders with cuts inside long runs (parity incl. the helpers' tests), the all-zero row of the bench
mkdir -p gpurun_out/r06_c06
python -m pytest tests/test_gpu_parity.py tests/test_gpu_low_entropy_helpers.py -q -x -k "low_entropy or rle8m" > gpurun_out/r06_c06/le.log 2>&1; echo "le rc=$?"
tail -4 gpurun_out/r06_c06/le.log
python tools/low_entropy_bench.py 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c06/le_bench.log
