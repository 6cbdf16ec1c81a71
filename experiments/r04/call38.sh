#!/bin/bash
# unified header parse (hsrle_parse.hip.h) in the per-lane loop of the block decoder for 2 .. 8 byte symbols: parity, then same-box A/B against the build before
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mono.py tests/test_gpu_split.py -x -q -k "not wave" 2>&1 | tail -4 )
K=rle16_sym,rle16_sym_packed,rle16_3symlut_byte,rle16_7symlut_byte,rle24_sym,rle24_byte_packed,rle24_3symlut_byte,rle32_byte,rle32_sym_packed,rle32_7symlut_byte,rle48_byte_packed,rle48_7symlut_byte,rle64_byte,rle64_3symlut_byte,rle16_sym_short,rle32_3symlut_byte_short,rle64_7symlut_byte_short
for lib in variants/libhsrle_base.so ""; do env HSRLE_LIB=$lib timeout 1200 python tools/ab_codecs.py 4096 $K 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$5,$9}'; done > gpurun_out/ab_parse.txt
python - <<'PY'
rows=[l.split() for l in open('gpurun_out/ab_parse.txt') if len(l.split())>=5]
half=len(rows)//2
for a,b in zip(rows[:half],rows[half:]):
    assert a[1]==b[1] and a[2]==b[2]
    print('%-28s %-5s base %5s new %5s  %+5.1f%% %s %s'%(a[1],a[2],a[3],b[3],(float(b[3])/float(a[3])-1)*100,a[4],b[4]))
PY
