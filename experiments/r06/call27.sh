#!/bin/bash
# round 6, call 27: the enqueue-only monolithic encode (plain, refused codecs, HIP graph replay), the windowed tests again, the mono tests, a short stress
mkdir -p gpurun_out/r06_c27
{
timeout 900 python -m pytest tests/test_gpu_mono_async.py -q -m gpu -k "encode" -x 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_pp.py -q -m gpu -k "windowed" -x 2>&1 | tail -4
timeout 1500 python -m pytest tests -q -m gpu -k "mono or dropin" -x 2>&1 | tail -4
STRESS_KEYS=rle8_multi,rle8_packed_multi timeout 300 python tools/gpu_stress.py 120 47 2>&1 | tail -4
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c27/log.txt
