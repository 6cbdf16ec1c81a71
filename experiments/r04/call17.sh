#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4q
fmt='
import sys, json
for l in sys.stdin:
    j = json.loads(l); print("dec %.3f ms frac %.4f | enc %.3f ms | ok %s" % (j["ms_per_step"], j["roofline"]["frac"], j["encode"]["ms"], j["bit_exact"]))'
{
for v in default old default old; do echo "== $v"; if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$PWD/variants/libhsrle_$v.so; fi; timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "$fmt"; done
K=rle8_packed_multi,rle8_3symlut,rle16_sym,rle16_3symlut_byte,rle32_byte_packed,rle64_3symlut_byte,rle24_sym,rle8_single
HSRLE_LIB=$PWD/variants/libhsrle_old.so timeout 600 python tools/ab_codecs.py 8192 $K 2>&1 | grep -v amdgpu.ids
unset HSRLE_LIB; timeout 600 python tools/ab_codecs.py 8192 $K 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r4q/log.txt 2>&1
python - <<'PY'
import collections
d=collections.defaultdict(dict)
for l in open('gpurun_out/r4q/log.txt'):
    p=l.split()
    if len(p)>=8 and p[3]=='dec': d[(p[1],p[2])].setdefault(p[0],[]).append(float(p[4]))
    elif l.startswith('==') or l.startswith('dec'): print(l.rstrip())
for k,v in d.items():
    o=max(v.get('libhsrle_old.so',[0])); n=max(v.get('default',[0]))
    print('%-26s %-5s old %6.0f new %6.0f  %+.1f%%'%(k[0],k[1],o,n,(n/o-1)*100 if o else 0))
PY
( time timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | tail -8
