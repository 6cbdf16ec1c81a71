#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mono.py -x -q -k "single" 2>&1 | tail -3
for k in rle8_single rle8_packed_single; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
