#!/bin/bash
# does the 4096-byte row stride of the output matter? (block sizes that are not powers of two)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4d
fmt='
import sys, json
for l in sys.stdin:
    j = json.loads(l); print("B %5d dec %.3f ms frac %.4f | enc %.3f ms frac %.4f | ok %s" % (j["config"]["block_size"], j["ms_per_step"], j["roofline"]["frac"], j["encode"]["ms"], j["encode"]["roofline"]["frac"], j["bit_exact"]))'
{
for b in 4096 3968 4224 4352 2048 2176 8192 8320; do timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 --block $b 2>&1 | grep '^{' | python -c "$fmt"; done
} > gpurun_out/r4d/log.txt 2>&1
cat gpurun_out/r4d/log.txt
