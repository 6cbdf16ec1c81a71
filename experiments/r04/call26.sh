#!/bin/bash
# run list encoder against ring encoder at 4 GiB, all 88 codecs that have both, both data kinds (experiment build: HSRLE_RUNLIST=1 forces the run list)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4rl
KEYS=$(python - <<'PY'
import sys; sys.path.insert(0,'tests')
from hsrle_testlib import CODECS
ids=[i for i in range(len(CODECS)) if i<=3 or 6<=i<=45 or 50<=i<94]
print(",".join(CODECS[i].key for i in ids))
PY
)
export HSRLE_LIB=variants/libhsrle_exp.so
env HSRLE_RUNLIST=0 timeout 1500 python tools/ab_codecs.py 4096 $KEYS 2>&1 | grep -v amdgpu.ids > gpurun_out/r4rl/ring.txt
env HSRLE_RUNLIST=1 timeout 1500 python tools/ab_codecs.py 4096 $KEYS 2>&1 | grep -v amdgpu.ids > gpurun_out/r4rl/runlist.txt
python - <<'PY'
a=[l.split() for l in open('gpurun_out/r4rl/ring.txt') if ' enc ' in l]
b=[l.split() for l in open('gpurun_out/r4rl/runlist.txt') if ' enc ' in l]
for x,y in zip(a,b):
    assert x[1]==y[1] and x[2]==y[2]
    r=float(x[6]); l=float(y[6])
    print('%-28s %-5s ring %6.0f runlist %6.0f  %+5.1f%% %s %s' % (x[1],x[2],r,l,(l/r-1)*100,x[-1],y[-1]))
PY
unset HSRLE_LIB
echo "== mono decode: region / look-back"
for rl in "0 0" "4096 2048" "4096 3072" "2048 2048"; do set -- $rl; echo "-- region $1 lookback $2"; timeout 300 python tools/mono_bench.py --cases packed8_runs_1g,lut8_runs_256m,lut64_video_88m --reps 4 --region $1 --lookback $2 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tail -3; done
