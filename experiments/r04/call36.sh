#!/bin/bash
# look-ahead reads in the walks through global memory (GlobalReader::load24): monolithic decode, packet list decode of bigger containers
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( timeout 1500 python -m pytest tests/test_gpu_mono.py tests/test_gpu_split.py -x -q -k "not wave" 2>&1 | tail -3 )
timeout 300 python tools/mono_bench.py --reps 4 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tail -8
echo "-- region 4096 lookback 2048"
timeout 300 python tools/mono_bench.py --reps 4 --cases packed8_runs_1g --region 4096 --lookback 2048 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tail -1
timeout 300 python tools/split_bench.py --codec rle8_packed_multi --synth runs --size 67108864 --subs 1 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/split_bench.py --size 268435456 --subs 1 2>&1 | grep -v amdgpu.ids
for k in rle64_3symlut_byte rle48_byte; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
