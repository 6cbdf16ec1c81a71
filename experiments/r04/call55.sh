#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mono.py -x -q 2>&1 | tail -4
for k in rle8_single rle8_packed_single rle128_sym rle128_byte_packed rle64_3symlut_byte; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
timeout 300 python tools/mono_bench.py --reps 4 --cases packed8_runs_1g,lut64_video_88m 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -2
