#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4i
fmt='
import sys, json
for l in sys.stdin:
    j = json.loads(l); print("dec %.3f ms frac %.4f | enc %.3f ms frac %.4f | ok %s" % (j["ms_per_step"], j["roofline"]["frac"], j["encode"]["ms"], j["encode"]["roofline"]["frac"], j["bit_exact"]))'
{
for v in default old default old; do echo "== $v"; if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi; timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "$fmt"; done
K=rle8_multi,rle8_packed_multi,rle8_3symlut,rle8_7symlut,rle8_multi_short,rle8_3symlut_short
HSRLE_LIB=$PWD/variants/libhsrle_old.so timeout 600 python tools/ab_codecs.py 8192 $K 2>&1 | grep -v amdgpu.ids
unset HSRLE_LIB; timeout 600 python tools/ab_codecs.py 8192 $K 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r4i/log.txt 2>&1
cat gpurun_out/r4i/log.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
