#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R=$PWD; mkdir -p gpurun_out/r4o; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4o/split -o s -- python3 $R/tools/split_bench.py > $R/gpurun_out/r4o/split.log 2>&1
cd $R; cat gpurun_out/r4o/split.log | grep -v amdgpu | tail -8
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r4o/split/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print('%-100s calls %5s avg %10.1f us' % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3))
PY
# low-entropy drop-in speed (PCIe included), 64 MiB
python3 - <<'PY'
import sys, time, ctypes
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import hsrle, numpy as np
from hsrle_testlib import Oracle
ora=Oracle()
d=ora.synth(1,1,2,64<<20).tobytes()
L=hsrle.lib()
cap=len(d)+297
for name in ('rle8_low_entropy_compress','rle8_low_entropy_short_compress'):
    t0=time.time(); size,st=hsrle.call_dropin(name,d,cap); t1=time.time()
    t2=time.time(); n,back=hsrle.call_dropin(name.replace('compress','decompress'),st,len(d)); t3=time.time()
    print(name,'ratio %.3f'%(size/len(d)),'enc %.1f MiB/s dec %.1f MiB/s'%(64/(t1-t0),64/(t3-t2)),'ok',back==d)
PY
