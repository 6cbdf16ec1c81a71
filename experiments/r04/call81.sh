#!/bin/bash
# rle8_single_short: split encode of small containers, literal stretches as copy jobs, four-window skips: parity, stress of the Single family, times
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_mono.py -x -q -k "single" 2>&1 | grep -v "^Extension" | tail -3
STRESS_KEYS=rle8_single,rle8_packed_single timeout 300 python tools/gpu_stress.py 150 1201 2>&1 | grep -v amdgpu.ids | tail -4
timeout 120 python tools/frame_enc_time.py rle8_single_short 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/mono_enc_bench.py rle8_single_short,rle8_single 1 2>&1 | grep -v amdgpu.ids
