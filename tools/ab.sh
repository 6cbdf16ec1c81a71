#!/bin/bash
# tools/ab.sh <variant names...>: correctness probe + 8 GiB bench line for the default build and each variants/libhsrle_<name>.so
cd "$(dirname "$0")/.."
run() {
  echo "== $1"
  timeout 200 python tools/probe_correctness.py rle8_packed_multi,rle8_multi,rle8_3symlut,rle16_sym,rle64_3symlut_byte 2>&1 | grep -v amdgpu.ids | tail -3
  for rep in $(seq 1 ${REPS:-3}); do timeout 300 python bench.py --no-cpu --steps 10 --warmup 3 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('dec %.2f ms %.0f GiB/s frac %.4f w/cu %s | enc %.0f GiB/s | ok %s' % (j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline'].get('waves_per_cu'), j['encode']['value'], j['bit_exact']))
    else: print(l.rstrip()[-300:])"; done
}
[ -n "$SKIP_DEFAULT" ] || { HSRLE_LIB= ; unset HSRLE_LIB; run default; }
for v in "$@"; do export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; run $v; done
