#!/bin/bash
# k_mono_cutsS with 32 positions per trip: parity of everything that goes through it (monolithic encode of all S >= 2 codecs, split encode), 128-bit small-container time
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1500 python -m pytest tests/test_gpu_mono.py tests/test_gpu_parity.py -x -q -k "not single" 2>&1 | tail -3
for k in rle128_sym rle128_byte_packed rle64_sym; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
