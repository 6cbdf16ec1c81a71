#!/bin/bash
# split encode of small containers for the Greedy encoders: parity, then the 88 MB frame / 64 MiB run data timings
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py -x -q -k "greedy" 2>&1 | tail -5
for k in rle16_3symlut_byte_short_greedy rle32_7symlut_byte_short_greedy rle64_1symlut_byte_short_greedy rle64_7symlut_byte_short_greedy; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
for k in rle32_7symlut_byte_short_greedy rle64_3symlut_byte_short_greedy; do HSRLE_SPLIT_PIECES=8 timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
