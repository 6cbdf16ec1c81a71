#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mono.py tests/test_gpu_split.py -x -q -k "not wave" 2>&1 | tail -3 )
K=rle16_sym_short,rle16_1symlut_sym_short,rle16_7symlut_byte_short,rle24_byte_short,rle24_3symlut_byte_short,rle32_sym_short,rle32_1symlut_byte_short,rle32_7symlut_byte_short,rle48_byte_short,rle48_3symlut_sym_short,rle64_sym_short,rle64_7symlut_byte_short,rle16_sym,rle32_3symlut_byte
env HSRLE_LIB=variants/libhsrle_base.so timeout 1200 python tools/ab_codecs.py 4096 $K 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$5,$9}' > gpurun_out/ab_parse3.txt
timeout 1200 python tools/ab_codecs.py 4096 $K 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$5,$9}' >> gpurun_out/ab_parse3.txt
python - <<'PY'
rows=[l.split() for l in open('gpurun_out/ab_parse3.txt') if len(l.split())>=4]
base=[r for r in rows if r[0].startswith('libhsrle_base')]; new=[r for r in rows if r[0]=='default']
for a,b in zip(base,new):
    assert a[1]==b[1] and a[2]==b[2]
    print('%-28s %-5s base %5s new %5s  %+5.1f%% %s'%(a[1],a[2],a[3],b[3],(float(b[3])/float(a[3])-1)*100,b[-1]))
PY
