#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4ss
cd /tmp && export TMPDIR=/tmp
for k in rle8_single rle128_sym; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4ss/$k" -o f -- python3 "$R/tools/frame_enc_time.py" $k > "$R/gpurun_out/r4ss/$k.log" 2>&1
done
cd "$R"
python - <<'PY'
import csv,glob
for k in ('rle8_single','rle128_sym'):
    for f in glob.glob('gpurun_out/r4ss/%s/**/f_kernel_trace.csv'%k,recursive=True):
        rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
        rows=[r for r in rows if 'synth' not in r['Kernel_Name']]
        # split by dataset: find the last call sequence of each dataset: print the last 12 kernels before each big gap
        seq=[(r['Kernel_Name'][:48], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000) for r in rows]
        n=len(seq)
        print(k, 'first dataset last call:'); 
        half=n//2
        for s in seq[half-10:half]: print('   ',s)
        print(k, 'second dataset last call:')
        for s in seq[-10:]: print('   ',s)
PY
