#!/bin/bash
# differential stress of the round's new encode paths (split encode of Single / 128 bit / Greedy one-symbol) and of everything, fresh seeds
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
STRESS_KEYS=rle8_single,rle8_packed_single,rle128,rle16_1symlut_byte_short_g,rle24_1symlut_byte_short_g,rle32_1symlut_byte_short_g,rle48_1symlut_byte_short_g,rle64_1symlut_byte_short_g timeout 400 python tools/gpu_stress.py 240 404 2>&1 | grep -v amdgpu.ids | tail -4
timeout 400 python tools/gpu_stress.py 240 405 2>&1 | grep -v amdgpu.ids | tail -4
