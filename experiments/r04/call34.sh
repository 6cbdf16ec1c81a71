#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4enc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4enc/frame" -o f -- python3 "$R/tools/frame_prof.py" rle64_3symlut_byte > "$R/gpurun_out/r4enc/frame.log" 2>&1
cd "$R"
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r4enc/frame/**/f_kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hsrle' in r['Name'] and 'synth' not in r['Name']: print(' ', r['Name'][:110], r['Calls'], r['AverageNs'])
rows=[]
for f in glob.glob('gpurun_out/r4enc/frame/**/f_kernel_trace.csv',recursive=True):
    rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=[r for r in rows if 'synth' not in r['Kernel_Name']][-8:]
t0=int(last[0]['Start_Timestamp'])
for r in last: print(' ', r['Kernel_Name'][:60], (int(r['Start_Timestamp'])-t0)/1000, (int(r['End_Timestamp'])-t0)/1000)
PY
for k in rle64_3symlut_byte rle8_packed_multi; do timeout 120 python tools/frame_enc_time.py $k 2>&1 | grep -v amdgpu.ids; done
( timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 )
