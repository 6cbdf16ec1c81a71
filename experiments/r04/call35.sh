#!/bin/bash
# mixed rounds in k_encodeS_runlist (phase C takes the rest of one block and the start of the next in one round)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5 )
for k in rle64_3symlut_byte rle32_sym_packed rle16_sym rle48_7symlut_byte_short; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
( timeout 900 python -m pytest tests/test_gpu_big.py -x -q -k "config3 or frame" 2>&1 | tail -3 )
