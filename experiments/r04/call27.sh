#!/bin/bash
# branch-free packet parse + expand with four chunks in flight + per-input run list / ring choice for 3 / 4 byte symbols
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4c27
( timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_mono.py -x -q -k "not wave" 2>&1 | tail -4 )
( timeout 1500 python -m pytest tests/test_gpu_big.py -x -q 2>&1 | tail -4 )
timeout 300 python tools/split_bench.py --subs 4096,1024,1 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/split_bench.py --codec rle8_packed_multi --synth runs --size 67108864 --subs 4096,1 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/mono_bench.py --reps 4 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -8
timeout 900 python tools/ab_codecs.py 4096 rle24_sym_packed,rle24_byte,rle32_byte_packed,rle32_3symlut_byte,rle32_7symlut_sym,rle24_sym_short,rle32_3symlut_byte_short,rle16_sym,rle64_byte 2>&1 | grep -v amdgpu.ids
