#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4r
fmt='
import sys, json
for l in sys.stdin:
    j = json.loads(l); print("dec %.4f ms frac %.4f kernel_ms %.4f" % (j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["kernel_ms"]))'
for v in default sector old default sector old; do echo -n "$v: "; if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi; timeout 300 python bench.py --no-cpu --no-extras --steps 30 --warmup 5 2>&1 | grep '^{' | python -c "$fmt"; done 2>&1 | tee gpurun_out/r4r/log.txt
