#!/bin/bash
# quarter-wave compaction of the split encode's small chunks: parity (every codec x block sizes that take the split encode), then times
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_mono.py -x -q 2>&1 | grep -v "^Extension" | tail -3
for k in rle8_single rle128_sym rle64_1symlut_byte_short_greedy; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
