#!/bin/bash
# round 4, GPU call 1: two-wave decoder (mw) against the default build; encoder store ablations (timing only)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4a
{
echo "== default"; for i in 1 2; do timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('dec %.3f ms frac %.4f w/cu %s | enc %.3f ms | ok %s' % (j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('waves_per_cu'), j['encode']['ms'], j['bit_exact']))"; done
echo "== mw probe"; HSRLE_LIB=$PWD/variants/libhsrle_mw.so timeout 180 python tools/probe_correctness.py rle8_packed_multi,rle8_multi 2>&1 | grep -v amdgpu.ids | tail -5
echo "== mw bench"; for i in 1 2; do HSRLE_LIB=$PWD/variants/libhsrle_mw.so timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('dec %.3f ms frac %.4f w/cu %s | enc %.3f ms | ok %s' % (j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('waves_per_cu'), j['encode']['ms'], j['bit_exact']))"; done
for v in es1 es2 es3; do echo "== $v"; HSRLE_LIB=$PWD/variants/libhsrle_$v.so timeout 200 python tools/enc_time.py rle8_packed_multi 2>&1 | grep -v amdgpu.ids | tail -1; done
echo "== default enc_time"; timeout 200 python tools/enc_time.py rle8_packed_multi 2>&1 | grep -v amdgpu.ids | tail -1
} > gpurun_out/r4a/log.txt 2>&1
cat gpurun_out/r4a/log.txt
