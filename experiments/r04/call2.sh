#!/bin/bash
# round 4, GPU call 2: three-wave decoder (mw3) against the default build
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4b
fmt='
import sys, json
for l in sys.stdin:
    j = json.loads(l); print("dec %.3f ms frac %.4f w/cu %s | enc %.3f ms | ok %s" % (j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("waves_per_cu"), j["encode"]["ms"], j["bit_exact"]))'
{
echo "== mw3 probe"; HSRLE_LIB=$PWD/variants/libhsrle_mw3.so timeout 180 python tools/probe_correctness.py rle8_packed_multi,rle8_multi,rle8_single,rle8_packed_single 2>&1 | grep -v amdgpu.ids | tail -5
echo "== mw3 bench"; for i in 1 2; do HSRLE_LIB=$PWD/variants/libhsrle_mw3.so timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "$fmt"; done
echo "== default"; for i in 1; do timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "$fmt"; done
echo "== mw3 video"; HSRLE_LIB=$PWD/variants/libhsrle_mw3.so timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 --synth video 2>&1 | grep '^{' | python -c "$fmt"
echo "== default video"; timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 --synth video 2>&1 | grep '^{' | python -c "$fmt"
} > gpurun_out/r4b/log.txt 2>&1
cat gpurun_out/r4b/log.txt
