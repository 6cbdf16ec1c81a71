#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4sx
cd /tmp && export TMPDIR=/tmp
for k in rle8_single; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4sx/$k" -o f -- python3 "$R/tools/frame_enc_time.py" $k > "$R/gpurun_out/r4sx/$k.log" 2>&1
done
cd "$R"
python3 - <<'PY'
import csv,glob
for k in ('rle8_single',):
    for f in glob.glob('gpurun_out/r4sx/%s/**/f_kernel_trace.csv'%k,recursive=True):
        rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
        rows=[r for r in rows if 'synth' not in r['Kernel_Name']]
        seq=[(r['Kernel_Name'][:60], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000, int(r['Start_Timestamp'])) for r in rows]
        idx=[i for i,s in enumerate(seq) if 'k_single_pick' in s[0]]
        for last in (idx[len(idx)//2-1], idx[-1]):
            i0=max(0,last-2); t0=seq[i0][2]; print('--',k)
            for s in seq[i0:i0+16]: print('  %-62s %8.1f us  @%8.1f'%(s[0],s[1],(s[2]-t0)/1000))
PY
