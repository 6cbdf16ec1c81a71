#!/bin/bash
# cut finder for multi-byte symbols with the straight-line window path: parity of everything that goes through it, then times
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1500 python -m pytest tests/test_gpu_mono.py tests/test_gpu_parity.py tests/test_gpu_split.py -x -q 2>&1 | tail -3
for k in rle128_sym rle64_1symlut_byte_short_greedy rle16_1symlut_byte_short_greedy; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
timeout 600 python tools/mono_enc_bench.py rle16_sym,rle24_byte_packed,rle32_3symlut_sym,rle64_3symlut_byte,rle128_sym 1 2>&1 | grep -v amdgpu.ids
