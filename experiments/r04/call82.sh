#!/bin/bash
# where the monolithic rle8_single encode of run-distributed data (symbol rare: all literals) spends its 8 ms per GiB
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4ms
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4ms" -o f -- python3 "$R/tools/mono_enc_bench.py" rle8_single 1 > "$R/gpurun_out/r4ms/log.txt" 2>&1
tail -2 "$R/gpurun_out/r4ms/log.txt"
python3 - "$R/gpurun_out/r4ms" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/f_kernel_trace.csv',recursive=True):
    rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
    rows=[r for r in rows if 'synth' not in r['Kernel_Name']]
    seq=[(r['Kernel_Name'][:64], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000) for r in rows]
    idx=[i for i,s in enumerate(seq) if 'k_single_pick_mono' in s[0]]
    # the kind-0 (run data) calls come first; print the 3rd call's sequence, then the last call's (kind 1)
    for st in (idx[3], idx[-1]):
        print('--')
        for s in seq[st:st+14]: print('  %-66s %9.1f us'%s)
PY
