#!/bin/bash
# round 6, call 28: monolithic decode of the 1 GiB rle8_packed stream, regions / look-back of the index walk (more, shorter latency chains?)
mkdir -p gpurun_out/r06_c28
for rm in "0 0" "4096 2048" "2048 2048" "2048 1024" "4096 1024" "1024 1024" "1024 512" "8192 2048"; do set -- $rm; echo "== region $1 lookback $2"; timeout 300 python tools/mono_bench.py --cases packed8_runs_1g --reps 4 --region $1 --lookback $2 2>&1 | grep -v amdgpu.ids | tail -2; done | tee gpurun_out/r06_c28/log.txt
