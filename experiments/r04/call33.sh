#!/bin/bash
# small-container encode after: one-launch size scan + finish, one walk per wave in the 8-byte run list encoder (candidate list 832)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for k in rle64_3symlut_byte rle8_packed_multi rle32_sym_packed rle8_single; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
( timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 )
( timeout 600 python -m pytest tests/test_gpu_split.py -x -q -k "config3 or range" 2>&1 | tail -2 )
