#!/bin/bash
# Greedy with a one-symbol list: split encode with the cut symbol as the first guess; 4 or 8 pieces per block for the per-lane chunk encoders
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py -x -q -k "greedy" 2>&1 | tail -3
for p in 4 8; do echo "pieces $p"; for k in rle16_1symlut_byte_short_greedy rle32_1symlut_byte_short_greedy rle64_1symlut_byte_short_greedy rle128_sym rle8_single; do HSRLE_PPB_TEST=$p timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done; done
