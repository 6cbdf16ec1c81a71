#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mono.py -x -q -k "greedy or short or small or fuzz or stream" 2>&1 | tail -4 )
( timeout 1200 python -m pytest tests/test_gpu_big.py -x -q 2>&1 | tail -2 )
K=rle16_1symlut_byte_short_greedy,rle16_7symlut_byte_short_greedy,rle24_3symlut_byte_short_greedy,rle32_1symlut_byte_short_greedy,rle32_7symlut_byte_short_greedy,rle48_3symlut_byte_short_greedy,rle64_1symlut_byte_short_greedy,rle64_7symlut_byte_short_greedy
env HSRLE_LIB=variants/libhsrle_base.so timeout 1200 python tools/ab_codecs.py 4096 $K 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$7,$9}' > gpurun_out/ab_greedy.txt
timeout 1200 python tools/ab_codecs.py 4096 $K 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$7,$9}' >> gpurun_out/ab_greedy.txt
python - <<'PY'
rows=[l.split() for l in open('gpurun_out/ab_greedy.txt') if len(l.split())>=4]
base=[r for r in rows if r[0].startswith('libhsrle_base')]; new=[r for r in rows if r[0]=='default']
for a,b in zip(base,new):
    assert a[1]==b[1] and a[2]==b[2]
    print('%-34s %-5s enc base %5s new %5s  %+5.1f%% %s'%(a[1],a[2],a[3],b[3],(float(b[3])/float(a[3])-1)*100,b[-1]))
PY
