#!/bin/bash
# where the packet list decode stops paying: container sizes 16 .. 512 MiB, plain / records / list
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4pl
for sz in 16777216 33554432 134217728 268435456 536870912; do
  timeout 300 python tools/split_bench.py --size $sz --subs 4096,1024,1 2>&1 | grep -v amdgpu.ids
  timeout 300 python tools/split_bench.py --codec rle8_packed_multi --synth runs --size $sz --subs 4096,1024,1 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4pl/sizes.txt
