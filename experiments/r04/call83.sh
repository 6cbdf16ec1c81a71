#!/bin/bash
# 128-byte symbol-free skips (cut finder, Single chunk encoders): parity, stress, times
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_mono.py -x -q -k "single or rle8_" 2>&1 | grep -v "^Extension" | tail -3
STRESS_KEYS=rle8_single,rle8_packed_single timeout 300 python tools/gpu_stress.py 120 1301 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python tools/mono_enc_bench.py rle8_single,rle8_packed_single,rle8_single_short 1 2>&1 | grep -v amdgpu.ids
for k in rle8_single rle8_single_short; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
