#!/bin/bash
# packet list walk from global memory (64 lanes per wave, no LDS window) against the LDS-staged walk, by container size
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4pl
for sz in 16777216 88473600 268435456 536870912; do
  for g in 999999999999 0; do
    echo "== HSRLE_PL_GLOBAL=$g (payload bytes above which the walk reads global memory)"
    env HSRLE_PL_GLOBAL=$g timeout 300 python tools/split_bench.py --size $sz --subs 1 2>&1 | grep -v amdgpu.ids
    env HSRLE_PL_GLOBAL=$g timeout 300 python tools/split_bench.py --codec rle8_packed_multi --synth runs --size $sz --subs 1 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/r4pl/global.txt
