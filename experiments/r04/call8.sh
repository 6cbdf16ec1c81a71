#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4h
fmt='
import sys, json
for l in sys.stdin:
    j = json.loads(l); print("dec %.3f ms frac %.4f | enc %.3f ms | ok %s" % (j["ms_per_step"], j["roofline"]["frac"], j["encode"]["ms"], j["bit_exact"]))'
{
for v in default prio1 prio3 default prio1 prio3; do echo "== $v"; if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi; timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "$fmt"; done
export HSRLE_LIB=$PWD/variants/libhsrle_exp.so
for r in 64 128; do echo "== ring $r"; HSRLE_DEC_RING=$r timeout 300 python tools/ab_codecs.py 8192 rle16_sym,rle16_3symlut_byte,rle32_byte_packed,rle32_3symlut_sym,rle24_byte_packed 1 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r4h/log.txt 2>&1
cat gpurun_out/r4h/log.txt
