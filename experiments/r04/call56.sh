#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4ss2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4ss2/s" -o f -- python3 "$R/tools/frame_enc_time.py" rle8_single > "$R/gpurun_out/r4ss2/s.log" 2>&1
cd "$R"
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r4ss2/s/**/f_kernel_trace.csv',recursive=True):
    rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
    rows=[r for r in rows if 'synth' not in r['Kernel_Name']]
    seq=[(r['Kernel_Name'][:56], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000, int(r['Start_Timestamp'])) for r in rows]
    idx=[i for i,s in enumerate(seq) if 'k_single_pick' in s[0]]
    for last in (idx[len(idx)//2-1], idx[-1]):
        t0=seq[last][2]; print('--')
        for s in seq[last:last+12]: print('  %-58s %8.1f us  @%8.1f'%(s[0],s[1],(s[2]-t0)/1000))
PY
