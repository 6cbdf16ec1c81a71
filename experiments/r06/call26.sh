#!/bin/bash
# round 6, call 26: windowed encoder -- its tests, and the differential stress of the two codecs (block sizes above 4 KiB in the mix, drop-in monolithic streams)
mkdir -p gpurun_out/r06_c26
{
timeout 1500 python -m pytest tests/test_gpu_pp.py -q -m gpu -k "windowed" -x 2>&1 | tail -8
STRESS_KEYS=rle8_multi,rle8_packed_multi timeout 400 python tools/gpu_stress.py 240 31 2>&1 | tail -8
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c26/log.txt
