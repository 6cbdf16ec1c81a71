#!/bin/bash
# split encode for the small containers of the 8 bit Single and the 128 bit codecs
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -12 )
for k in rle8_single rle8_packed_single rle128_sym rle128_byte_packed rle128_sym_packed; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
