#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "low_entropy or rle8m" 2>&1 | tail -12
python3 - <<'PY'
import sys, time, ctypes
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import hsrle, numpy as np
from hsrle_testlib import Oracle
ora=Oracle()
for kind,size in ((1,64<<20),(0,64<<20),(1,1<<30)):
    d=ora.synth(kind,1,2,size).tobytes()
    cap=len(d)+297
    for name in ('rle8_low_entropy_compress','rle8_low_entropy_short_compress_only_max_frequency'):
        hsrle.call_dropin(name,d[:1<<20],(1<<20)+297)
        t0=time.time(); sz,st=hsrle.call_dropin(name,d,cap); t1=time.time()
        dn='rle8_low_entropy_short_decompress' if 'short' in name else 'rle8_low_entropy_decompress'
        t2=time.time(); n,back=hsrle.call_dropin(dn,st,len(d)); t3=time.time()
        ok = back==d
        ref = ora.low_entropy_compress((1 if 'short' in name else 0)|(2 if 'only' in name else 0), d) if size <= (64<<20) else None
        print(name,'kind',kind,size>>20,'MiB ratio %.3f'%(sz/len(d)),'enc %.0f MiB/s dec %.0f MiB/s (host pointers, PCIe included)'%((size>>20)/(t1-t0),(size>>20)/(t3-t2)),'roundtrip',ok,'== oracle',(ref==st) if ref is not None else 'n/a', flush=True)
PY
