#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4g
K=rle16_sym,rle16_byte_packed,rle16_3symlut_byte,rle16_7symlut_sym,rle24_3symlut_byte,rle32_byte_packed,rle32_3symlut_sym,rle48_7symlut_byte,rle64_3symlut_byte,rle64_sym,rle16_sym_short,rle32_3symlut_byte_short
{
for rep in 1 2; do
HSRLE_LIB=$PWD/variants/libhsrle_old.so timeout 600 python tools/ab_codecs.py 8192 $K 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/ab_codecs.py 8192 $K 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r4g/ab.txt 2>&1
python - <<'PY'
import collections
d=collections.defaultdict(dict)
for l in open('gpurun_out/r4g/ab.txt'):
    p=l.split()
    if len(p)<8 or p[3]!='dec': continue
    d[(p[1],p[2])].setdefault(p[0],[]).append(float(p[4]))
for k,v in d.items():
    o=max(v.get('libhsrle_old.so',[0])); n=max(v.get('default',[0]))
    print('%-26s %-5s old %6.0f new %6.0f  %+.1f%%'%(k[0],k[1],o,n,(n/o-1)*100 if o else 0))
PY
