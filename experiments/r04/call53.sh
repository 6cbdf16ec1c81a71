#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "single or 128" 2>&1 | tail -5
for k in rle8_single rle8_packed_single rle128_sym rle128_byte_packed rle128_sym_packed; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
